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


def kernels(so_path):
    blob = open(so_path, "rb").read()
    out = {}
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
            elf = blob[base + offset:base + offset + size]
            for name, ntype, desc in _elf_notes(elf):
                if name == b"AMDGPU" and ntype == 32:
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in meta.get("amdhsa.kernels", []):
                        out[k[".name"]] = k
    return out
