"""Scan hipcc's gfx950 assembly for the store-data hazard hipcc does not guard (round 6, NOTES; tests/codeobj.py::store_data_hazards has the
rule; tests/test_host.py runs the same scan over the shipped library's disassembly).
Usage: python3 scripts/isa_store_hazard.py [/tmp/isa/*.s | librpn_hip.so]      exit code 1 if a candidate is found."""
import glob, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import codeobj

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
bad = 0
for path in sys.argv[1:] or sorted(glob.glob("/tmp/isa/*.s")):
    if path.endswith(".so"):
        texts = []
        for elf in codeobj.code_objects(path):
            with tempfile.NamedTemporaryFile(suffix=".elf") as f:
                f.write(elf)
                f.flush()
                texts.append(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout)
    else:
        texts = [open(path).read()]
    for text in texts:
        for kernel, store, nxt in codeobj.store_data_hazards(text):
            bad += 1
            print("%s  %s\n      %s\n      %s" % (path, kernel[:80], store, nxt))
print("candidates:", bad)
sys.exit(1 if bad else 0)
