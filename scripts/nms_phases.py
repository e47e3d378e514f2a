"""Time the NMS kernel with RPN_NMS_STOP set by the caller (outputs are garbage when it is set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
from oracle import bbox_oracle as bo
from tf_rpn_amd import _lib as L
VAR = np.float32([0.1, 0.1, 0.2, 0.2])
anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
A = len(anchors)
_model = {}
def model_outputs(B):     # the head outputs of bench.py's model (random weights, U[0,1) images)
    if not _model:
        from tf_rpn_amd.models._rpn_model import synthetic_weights
        from tf_rpn_amd.predictor import Proposer
        from tf_rpn_amd.utils import train_utils
        hp = dict(train_utils.get_hyper_params("vgg16"))
        prop = Proposer("vgg16", hyper_params=hp, weights=synthetic_weights("vgg16", hp, seed=1), precision="f16x3",
                        max_batch=B, iou_threshold=0.7, overlap_nms=True)
        gen = torch.Generator(device="cuda"); gen.manual_seed(0)
        prop.propose_async(torch.rand((B, 500, 500, 3), generator=gen, device="cuda", dtype=torch.float32))
        torch.cuda.synchronize()
        _model["d"] = prop._bufs[0]["reg"][:B].view(B, -1, 4).cpu().numpy().copy()
        _model["s"] = prop._bufs[0]["cls"][:B].view(B, -1).cpu().numpy().copy()
    return _model["d"], _model["s"]
THR = float(os.environ.get("NMS_THR", "0.7"))
for B, kind in ((8, "perm"), (8, "model-like"), (8, "model"), (64, "perm")):
    rng = np.random.RandomState(2)
    deltas = rng.standard_normal((B, A, 4)).astype(np.float32)
    if kind == "model":
        deltas, scores = model_outputs(B)
    elif kind == "perm":
        scores = cases.permutation_scores(np.random.RandomState(3), B, A)
    else:   # smooth score field: neighbours have similar scores -> deep walks (like a conv head's output)
        f = rng.standard_normal((B, 31, 31, 9)).astype(np.float32)
        for _ in range(3):
            f = (f + np.roll(f, 1, 1) + np.roll(f, 1, 2) + np.roll(f, -1, 1) + np.roll(f, -1, 2)) / 5
        scores = (1 / (1 + np.exp(-8 * f))).reshape(B, A).astype(np.float32)
    d, s, a = torch.from_numpy(deltas).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(anchors).cuda()
    ob = torch.zeros((B, 300, 4), device="cuda"); osc = torch.zeros((B, 300), device="cuda")
    oi = torch.zeros((B, 300), dtype=torch.int32, device="cuda"); ov = torch.zeros((B,), dtype=torch.int32, device="cuda")
    keep, vptr = L.host_floats(VAR)
    wsb = int(L.lib().rpn_nms_workspace_bytes(B, A, 1, 300, 300))
    ws = torch.empty((max(16, wsb),), dtype=torch.uint8, device="cuda")
    def run():
        L.check(L.lib().rpn_decode_nms(L.ptr(a), L.ptr(d), vptr, L.ptr(s), B, A, 300, THR, float("-inf"), 1, L.ptr(ob),
                                       L.ptr(osc), L.ptr(oi), L.ptr(ov), L.ptr(ws), wsb, L.stream_ptr()), "nms")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("thr=%s B=%d %s: %.1f us" % (THR, B, kind, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
