"""In-kernel phase timing of the 16x16x32 split conv kernel (needs the -DRPN_STAMP build:
RPN_HIP_LIB=tf_rpn_amd/csrc/librpn_hip_stamp.so python scripts/stamp_probe.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_rpn_amd import _lib as L

lib = L.lib()
raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run(B, H, Cin, Cout, reps=3):
    x = torch.rand((B, H, H, Cin), device="cuda"); w = torch.randn((3, 3, Cin, Cout), device="cuda") * 0.05
    b = torch.zeros((Cout,), device="cuda"); out = torch.empty((B, H, H, Cout), device="cuda")
    for _ in range(reps):
        L.check(lib.rpn_conv2d(L.ptr(x), B, H, H, Cin, L.ptr(w), L.ptr(b), 3, 3, Cout, 1, 1, 1, H, H, 1, 2, L.ptr(out), None), "conv")
    torch.cuda.synchronize()
    n = 8192 * 32
    st = np.zeros(n, dtype=np.uint64)
    assert raw.rpn_debug_read_stamps(st.ctypes.data, n) == 0
    st = st.reshape(8192, 32)
    chunks = Cin // 32
    nblk = 8 * ((max(1, (Cout + 127) // 128) + 0)) * 0  # unknown mapping; use nonzero rows
    live = st[:, 3] > 0
    s = st[live].astype(np.int64)
    t0 = s[:, 0].min()
    start, pro, loop_end, end = s[:, 0] - t0, s[:, 2] - s[:, 0], s[:, 4 + chunks - 1], s[:, 3]
    ch = np.diff(np.concatenate([s[:, 2:3], s[:, 4:4 + chunks]], axis=1), axis=1)
    epi = end - loop_end
    total = end - s[:, 0]
    print("layer B%d %dx%d %d->%d: %d workgroups, kernel span %d cycles" % (B, H, H, Cin, Cout, live.sum(), (end - t0).max()))
    print("  per workgroup (cycles, median [p10 p90]): total %d  prologue %d [%d %d]  chunk %d [%d %d] (first %d, last %d)  epilogue %d [%d %d]" % (
        np.median(total), np.median(pro), *np.percentile(pro, [10, 90]), np.median(ch), *np.percentile(ch, [10, 90]),
        np.median(ch[:, 0]), np.median(ch[:, -1]), np.median(epi), *np.percentile(epi, [10, 90])))
    for per in (48, 24):
        print("  ideal MFMA cycles per chunk (9 taps x 2 waves/SIMD x %d MFMA x 16): %d -> loop efficiency %.1f%%" % (per, 9 * 2 * per * 16, 100 * 9 * 2 * per * 16 / np.median(ch)))
    print("  shares of workgroup time: prologue %.1f%%  loop %.1f%%  epilogue %.1f%%" % (100 * pro.sum() / total.sum(), 100 * ch.sum() / total.sum(), 100 * epi.sum() / total.sum()))
    hw = s[:, 1]
    xcc, cu, se, sh = (hw >> 32) & 0xf, (hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1
    key = xcc * 1000 + se * 100 + sh * 10 + cu     # not unique per physical CU for sure, good enough to chain
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        o = idx[np.argsort(s[idx, 0])]
        for a_, b_ in zip(o[:-1], o[1:]):
            gaps.append(s[b_, 0] - s[a_, 3])
    gaps = np.array(gaps)
    print("  distinct (xcc,se,sh,cu) keys %d; gap between a workgroup's end and the next start on the same key: median %d [p10 %d p90 %d]" % (
        len(np.unique(key)), np.median(gaps), *np.percentile(gaps, [10, 90])))
    print("  start-time waves (cycles since first start): p25 %d p50 %d p75 %d max %d" % tuple(np.percentile(start, [25, 50, 75, 100])))
import sys as _s
cfgs = [tuple(int(v) for v in a.split(',')) for a in _s.argv[1:]] or [(8, 125, 256, 256), (8, 250, 128, 128)]
for cfg in cfgs:
    run(*cfg)
