"""Kernel metadata of the gfx950 code objects inside librpn_hip.so (no external tools): the clang offload bundles of the
.hip_fatbin section -> ELF notes (NT_AMDGPU_METADATA, msgpack) -> {kernel name: metadata dict}."""
import re
import struct

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        off = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, off + 4)
        sh_offset, sh_size = struct.unpack_from("<QQ", elf, off + 0x18)
        if sh_type != 7:                                    # SHT_NOTE
            continue
        pos, end = sh_offset, sh_offset + sh_size
        while pos + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, pos)
            pos += 12
            name = elf[pos:pos + namesz].rstrip(b"\0")
            pos += (namesz + 3) & ~3
            desc = elf[pos:pos + descsz]
            pos += (descsz + 3) & ~3
            yield name, ntype, desc


def code_objects(so_path):
    """The gfx950 ELF images inside the library's offload bundles."""
    blob = open(so_path, "rb").read()
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        n, = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            offset, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "amdgcn" not in triple or size == 0:
                continue
            yield blob[base + offset:base + offset + size]


def kernels(so_path):
    out = {}
    for elf in code_objects(so_path):
        for name, ntype, desc in _elf_notes(elf):
            if name == b"AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out


def _vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def store_data_hazards(asm_text):
    """Candidates for the store-data hazard hipcc does not guard on gfx950 (NOTES round 6, scripts/micro/store_hazard.hip): a buffer store
    of MORE than 64 bits whose scalar offset is a REGISTER, followed -- with no other instruction in between -- by an instruction that
    writes one of the store's data registers.  (One wait state is enough; with a constant scalar offset the compiler inserts two itself;
    dword stores are not affected.)  Accepts `hipcc -S` output and `llvm-objdump -d` output.  Returns [(kernel, store, next)]."""
    found, kernel, prev = [], "?", None
    for raw in asm_text.splitlines():
        line = raw.split("//")[0].split(";")[0].strip()
        m = re.match(r"(?:[0-9a-f]+ <)?(_Z\w+)>?:$", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        if prev is not None:
            op = line.split()[0]
            if op.startswith(("v_", "ds_read", "buffer_load", "global_load")) and not op.startswith("v_cmp") and len(line.split()) > 1:
                if _vregs(line.split()[1].rstrip(",")) & prev[0]:
                    found.append((kernel, prev[1], line))
        prev = None
        m = re.match(r"buffer_store_dwordx([34])\s+(v\[\d+:\d+\]),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)", line)
        if m and re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi", m.group(5)):
            prev = (_vregs(m.group(2)), line)
    return found


def short_name(mangled):
    """`_ZN3rpn18conv_igemm_f32_dmaILi2ELi2ELi2ELi2ELb0EEEvNS_8ConvArgsEiii` -> `conv_igemm_f32_dma<2,2,2,2,false>` (the library's
    kernels live in namespace rpn and take integer / bool template arguments only); anything else is returned unchanged."""
    m = re.match(r"_ZN3rpn(\d+)", mangled)
    if not m:
        return mangled
    start = m.end()
    n = int(m.group(1))
    name, rest = mangled[start:start + n], mangled[start + n:]
    if not rest.startswith("I"):
        return name
    args, pos = [], 1
    while pos < len(rest) and rest[pos] != "E":
        a = re.match(r"L([a-z])(n?\d+)E", rest[pos:])
        if not a:
            return mangled
        v = a.group(2).replace("n", "-")
        args.append({"0": "false", "1": "true"}[v] if a.group(1) == "b" else v)
        pos += a.end()
    return "%s<%s>" % (name, ",".join(args))


def table(so_path):
    """{short kernel name: (VGPRs incl. AGPRs, SGPR spills, VGPR spills, scratch bytes, LDS bytes, max workgroup size)}"""
    out = {}
    for name, k in kernels(so_path).items():
        # (.vgpr_count is the unified allocation: architectural registers + AGPRs)
        out[short_name(name)] = (k[".vgpr_count"], k.get(".sgpr_spill_count", 0), k.get(".vgpr_spill_count", 0), k[".private_segment_fixed_size"],
                                 k[".group_segment_fixed_size"], k[".max_flat_workgroup_size"])
    return out


if __name__ == "__main__":
    import sys
    for nm, row in sorted(table(sys.argv[1]).items()):
        print("%-64s vgpr %3d  sgpr-spill %3d  vgpr-spill %3d  scratch %4d  lds %6d  wg %4d" % ((nm,) + row))
