"""Epilogue phase stamps of the persistent kernel for a workgroup's first and second tile (-DRPN_STAMP build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_rpn_amd import _lib as L
lib = L.lib(); raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (B, H, Cin, Cout) in [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(8, 250, 64, 128), (8, 125, 256, 256)]:
    x = torch.rand((B, H, H, Cin), device="cuda"); w = torch.randn((3, 3, Cin, Cout), device="cuda") * 0.05
    b = torch.zeros((Cout,), device="cuda"); out = torch.empty((B, H, H, Cout), device="cuda")
    for _ in range(3):
        L.check(lib.rpn_conv2d(L.ptr(x), B, H, H, Cin, L.ptr(w), L.ptr(b), 3, 3, Cout, 1, 1, 1, H, H, 1, 2, L.ptr(out), None), "conv")
    torch.cuda.synchronize()
    st = np.zeros(8192 * 32, dtype=np.uint64); raw.rpn_debug_read_stamps(st.ctypes.data, st.size)
    s = st.reshape(8192, 32)[:256].astype(np.int64)
    print("layer B%d %dx%d %d->%d" % (B, H, H, Cin, Cout))
    for t in (0, 1):
        o = 20 + 6 * t
        d = np.diff(s[:, o:o + 6], axis=1)
        print("  tile %d: vmcnt(0) %d | stage row0 %d | store row0 %d | stage row1 %d | store row1 %d   (median cycles; total %d)" % (
            t, *np.median(d, axis=0), np.median(s[:, o + 5] - s[:, o])))
