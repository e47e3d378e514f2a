"""BASELINE config C3: batch=64 box path only (anchors + decode + IoU map + NMS), HBM-bandwidth view.
Prints achieved GB/s per kernel against its algorithmic bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
from tf_rpn_amd import _lib as L
from tf_rpn_amd.utils import bbox_utils, train_utils

def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

out = {}
for backbone in ("vgg16", "mobilenet_v2"):
    hp = dict(train_utils.get_hyper_params(backbone))
    B, G = 64, 42
    anchors = bbox_utils.generate_anchors(hp)
    A = anchors.shape[0]
    rng = np.random.RandomState(2)
    deltas = torch.from_numpy(rng.standard_normal((B, A, 4)).astype(np.float32)).cuda()
    scores = torch.from_numpy(cases.permutation_scores(np.random.RandomState(3), B, A)).cuda()
    gt = torch.from_numpy(cases.gt_boxes(np.random.RandomState(4), B, G=G, n_valid=10)).cuda()
    var = [0.1, 0.1, 0.2, 0.2]
    boxes = torch.empty((B, A, 4), device="cuda"); iou = torch.empty((B, A, G), device="cuda")
    keep, vptr = L.host_floats(var)
    lib = L.lib()
    r = {}
    t = timeit(lambda: bbox_utils.generate_anchors(hp))
    r["anchors"] = {"us": t * 1e6, "GB/s": 16 * A / t / 1e9}
    t = timeit(lambda: lib.rpn_decode(L.ptr(anchors), 0, L.ptr(deltas), vptr, B, A, L.ptr(boxes), L.stream_ptr()))
    r["decode"] = {"us": t * 1e6, "GB/s": 32.0 * B * A / t / 1e9}
    t = timeit(lambda: lib.rpn_iou_map(L.ptr(anchors), 0, A, L.ptr(gt), B, G, L.ptr(iou), L.stream_ptr()))
    r["iou_map"] = {"us": t * 1e6, "GB/s": (4.0 * B * A * G + 16 * (A + B * G)) / t / 1e9}
    for thr in (0.5, 0.7):
        ob = torch.zeros((B, 300, 4), device="cuda"); osc = torch.zeros((B, 300), device="cuda")
        oi = torch.zeros((B, 300), dtype=torch.int32, device="cuda"); ov = torch.zeros((B,), dtype=torch.int32, device="cuda")
        t = timeit(lambda: lib.rpn_decode_nms(L.ptr(anchors), L.ptr(deltas), vptr, L.ptr(scores), B, A, 300, thr, float("-inf"), 1,
                                              L.ptr(ob), L.ptr(osc), L.ptr(oi), L.ptr(ov), L.vp(0), 0, L.stream_ptr()), n=20)
        r["decode+nms iou=%.1f" % thr] = {"us": t * 1e6, "boxes/s": B * A / t, "GB/s": B * (20.0 * A + 7204) / t / 1e9}
        bx = boxes.view(B, A, 1, 4); sc3 = scores.view(B, A, 1)
        oc = torch.zeros((B, 300), device="cuda")
        t = timeit(lambda: lib.rpn_combined_nms(L.ptr(bx), L.ptr(sc3), B, A, 1, 1, 300, 300, thr, float("-inf"), 1, L.ptr(ob), L.ptr(osc),
                                                L.ptr(oc), L.ptr(oi), L.ptr(ov), L.vp(0), 0, L.stream_ptr()), n=20)
        r["nms iou=%.1f" % thr] = {"us": t * 1e6, "boxes/s": B * A / t}
    out["%s A=%d B=%d G=%d" % (backbone, A, B, G)] = r
print(json.dumps(out, indent=1))
