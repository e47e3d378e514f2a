"""IoU-map kernel timing at config C3 (B=64, A=8649, G=42)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_rpn_amd import _lib as L
from tf_rpn_amd.utils import bbox_utils, train_utils
hp = dict(train_utils.get_hyper_params("vgg16"))
B, G = 64, 42
anchors = bbox_utils.generate_anchors(hp); A = anchors.shape[0]
rng = np.random.RandomState(4)
gt = np.zeros((B, G, 4), np.float32); y1, x1 = rng.uniform(0, .7, (2, B, 10)); h, w = rng.uniform(.05, .3, (2, B, 10))
gt[:, :10] = np.stack([y1, x1, y1 + h, x1 + w], -1); gt = torch.from_numpy(gt).cuda()
iou = torch.empty((B, A, G), device="cuda")
lib = L.lib()
f = lambda: lib.rpn_iou_map(L.ptr(anchors), 0, A, L.ptr(gt), B, G, L.ptr(iou), L.stream_ptr())
for _ in range(5): f()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 50)
print("%s: iou_map %.2f us  %.0f GB/s" % (os.environ.get("RPN_IOU_EXP", "0") + "/nt" + os.environ.get("RPN_IOU_NT", "1") + "/chunked" + os.environ.get("RPN_IOU_CHUNKED", "1"), best * 1e3, 4.0 * B * A * G / (best * 1e-3) / 1e9))
