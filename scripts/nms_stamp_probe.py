"""Phase stamps of the NMS kernel (a -DRPN_NMS_STAMP build: RPN_HIP_LIB=ab/nmsstamp.so).  Prints, for workgroup 0 and the
median over workgroups, the cycles between consecutive stamps with their phase codes:
1 start, 2 band selected, 3 band compacted, 4 band sorted, 5 chunk set up, 6 step A, 7 step B, 8 walk (wave 0), 9 group done,
10 greedy done, 11 outputs written; 33 band select entered, 34 histogram zeroed (30: barrier behind it), 35 score pass done (31: barrier), 32 threshold found;
40 order: cursors zeroed, 43 keys scattered (41: barrier), 42 ranked."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
from oracle import bbox_oracle as bo
from tf_rpn_amd import _lib as L
KIND = sys.argv[1] if len(sys.argv) > 1 else "perm"
THR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
if KIND in ("model", "model_c5", "model_c1"):   # the head outputs of the bench's model (random weights, U[0,1) images): what bench.py's step feeds it
    from tf_rpn_amd.models._rpn_model import synthetic_weights
    from tf_rpn_amd.predictor import Proposer
    from tf_rpn_amd.utils import train_utils
    if KIND == "model":
        bb, B, size = "vgg16", 8, 500
        hp = dict(train_utils.get_hyper_params("vgg16"))
    elif KIND == "model_c1":         # configs[0]: MobileNetV2, 500 x 500, one image
        bb, B, size = "mobilenet_v2", 1, 500
        hp = dict(train_utils.get_hyper_params("mobilenet_v2"))
    else:                           # configs[4]: MobileNetV2, 1024 x 1024, 15 anchors per cell, one image
        bb, B, size = "mobilenet_v2", 1, 1024
        hp = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                               anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
    prop = Proposer(bb, hyper_params=hp, weights=synthetic_weights(bb, hp, seed=1), precision="f16x3", max_batch=B,
                    iou_threshold=0.7, overlap_nms=True)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    prop.propose_async(torch.rand((B, size, size, 3), generator=gen, device="cuda", dtype=torch.float32))
    torch.cuda.synchronize()
    a = prop.anchors; A = prop.total_anchors
    d = prop._bufs[0]["reg"][:B].view(B, -1, 4).clone(); s = prop._bufs[0]["cls"][:B].view(B, -1).clone()
else:
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16")); A = len(anchors)
    B = 64
    rng = np.random.RandomState(2)
    d = torch.from_numpy(rng.standard_normal((B, A, 4)).astype(np.float32)).cuda()
    s = torch.from_numpy(cases.permutation_scores(np.random.RandomState(3), B, A)).cuda()
    a = torch.from_numpy(anchors).cuda()
ob = torch.zeros((B, 300, 4), device="cuda"); osc = torch.zeros((B, 300), device="cuda")
oi = torch.zeros((B, 300), dtype=torch.int32, device="cuda"); ov = torch.zeros((B,), dtype=torch.int32, device="cuda")
keep, vptr = L.host_floats(np.float32([0.1, 0.1, 0.2, 0.2]))
lib = L.lib()
WS = torch.empty((max(16, int(lib.rpn_nms_workspace_bytes(B, A, 1, 300, 300))),), dtype=torch.uint8, device="cuda")
for _ in range(3):
    L.check(lib.rpn_decode_nms(L.ptr(a), L.ptr(d), vptr, L.ptr(s), B, A, 300, THR, float("-inf"), 1, L.ptr(ob), L.ptr(osc),
                               L.ptr(oi), L.ptr(ov), L.ptr(WS), WS.numel(), L.stream_ptr()), "nms")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 512))()
raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_nms_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.rpn_debug_read_nms_stamps(buf, 64 * 512) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 512)
code = (st >> np.uint64(56)).astype(int); cyc = (st & np.uint64((1 << 56) - 1)).astype(np.int64)
n = int((code[0] > 0).sum())
print("workgroup 0: %d stamps, total %d cycles" % (n, cyc[0, n - 1] - cyc[0, 0]))
row = ["%d:%d" % (code[0, i], cyc[0, i] - cyc[0, i - 1]) for i in range(1, n)]
print(" ".join(row))
# per-code totals, median over workgroups
tot = {}
for w in range(B):
    nw = int((code[w] > 0).sum())
    t = {}
    for i in range(1, nw):
        t[code[w, i]] = t.get(code[w, i], 0) + int(cyc[w, i] - cyc[w, i - 1])
    for k, v in t.items():
        tot.setdefault(k, []).append(v)
print("median cycles per phase code over workgroups:", {k: int(np.median(v)) for k, v in sorted(tot.items())})
print("valid:", ov[:4].tolist())
