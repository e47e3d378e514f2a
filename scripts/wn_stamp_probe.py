"""Per-wave slice timing of the wide F(4x4,3x3) Winograd kernel (needs the -DRPN_STAMP build: scripts/build_wn_stamp.sh, then
RPN_HIP_LIB=tf_rpn_amd/csrc/librpn_hip_wnstamp.so python scripts/wn_stamp_probe.py [B,H,Cin,Cout ...]).
For every wave of the first 64 workgroups: work = arrival at slice s's barrier - release from slice s-1's; wait = release - arrival."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_rpn_amd import _lib as L
lib = L.lib()
raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_wn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run(B, H, Cin, Cout):
    x = torch.rand((B, H, H, Cin), device="cuda"); w = torch.randn((3, 3, Cin, Cout), device="cuda") * 0.05
    b = torch.zeros((Cout,), device="cuda"); out = torch.empty((B, H, H, Cout), device="cuda")
    for _ in range(3):
        L.check(lib.rpn_conv2d(L.ptr(x), B, H, H, Cin, L.ptr(w), L.ptr(b), 3, 3, Cout, 1, 1, 1, H, H, 1, L.PRECISIONS["f32w"], L.ptr(out), None), "conv")
    torch.cuda.synchronize()
    n = 64 * 16 * 64
    st = np.zeros(n, dtype=np.uint64)
    assert raw.rpn_debug_read_wn_stamps(st.ctypes.data, n) == 0
    s = st.reshape(64, 16, 64).astype(np.int64)
    ns = min(28, Cin // 4)
    npair = ns // 2
    # slots (conv_wino_kernels.hip): MFMA waves stamp 2 s / 2 s + 1 around the barrier behind every ODD slice s; staging waves stamp
    # 4 P + 2 / 4 P + 3 around the barrier of pair P -- the same slots
    arr, rel = s[:, :, 2:4 * npair:4], s[:, :, 3:4 * npair:4]
    work = arr[:, :, 1:] - rel[:, :, :-1]          # a pair's work (pair >= 1)
    wait = rel - arr
    period = rel[:, :, 1:] - rel[:, :, :-1]
    sl = slice(2, npair - 1)
    print("layer B%d %dx%d %d->%d  pairs of slices stamped %d" % (B, H, H, Cin, Cout, npair))
    print("  pair period (release to release), median over workgroups/waves: %d cycles = %d per slice" % (np.median(period[:, :, sl]), np.median(period[:, :, sl]) / 2))
    print("  MFMA waves 0..11: work per pair median %d [p10 %d p90 %d]  wait at the barrier median %d [p10 %d p90 %d]" % (
        np.median(work[:, :12, sl]), *np.percentile(work[:, :12, sl], [10, 90]), np.median(wait[:, :12, 2:npair - 1]), *np.percentile(wait[:, :12, 2:npair - 1], [10, 90])))
    print("  staging waves 12..15: work per pair median %d [p10 %d p90 %d]  wait median %d [p10 %d p90 %d]" % (
        np.median(work[:, 12:, sl]), *np.percentile(work[:, 12:, sl], [10, 90]), np.median(wait[:, 12:, 2:npair - 1]), *np.percentile(wait[:, 12:, 2:npair - 1], [10, 90])))
    last = arr[:, :, 2:npair - 1].argmax(axis=1)
    print("  last arriver is a staging wave in %.0f %% of the barriers" % (100.0 * (last >= 12).mean()))
    m = s[:, :12, :]
    tot, pro, loop, epi = m[:, :, 63] - m[:, :, 60], m[:, :, 61] - m[:, :, 60], m[:, :, 62] - m[:, :, 61], m[:, :, 63] - m[:, :, 62]
    rt = (m[:, :, 59] - m[:, :, 58]).astype(np.float64)          # s_memrealtime ticks of 10 ns
    print("  per workgroup (MFMA waves, median): total %d = prologue %d + loop %d (%d per slice) + epilogue %d cycles;  in-kernel clock %.2f GHz" % (
        np.median(tot), np.median(pro), np.median(loop), np.median(loop) / (Cin // 4), np.median(epi), np.median(tot / np.maximum(rt, 1.0)) * 0.1))
    print("  (persistent workgroups: the stamps are those of a workgroup's LAST tile -- its prologue had its first pairs requested in front of the previous epilogue)")
cfgs = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(8, 125, 256, 256), (8, 250, 128, 128)]
for cfg in cfgs:
    run(*cfg)
