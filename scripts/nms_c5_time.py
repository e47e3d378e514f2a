"""decode+NMS time on the head outputs of configs[4]'s model (MobileNetV2, 1024 x 1024, 15 anchors per cell, one image:
61 440 candidates), of configs[0]'s (MobileNetV2, 500 x 500, one image) and of configs[1]'s (VGG16, 500 x 500, batch 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_rpn_amd import _lib as L
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils
for bb, B, size, kw in (("mobilenet_v2", 1, 1024, dict(img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1 / 3.])),
                        ("mobilenet_v2", 1, 500, dict(img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5])),
                        ("vgg16", 8, 500, dict(img_size=500, feature_map_shape=31, anchor_ratios=[1., 2., .5]))):   # (get_hyper_params keeps overrides: the reference's quirk)
    hp = dict(train_utils.get_hyper_params(bb, **kw))
    prop = Proposer(bb, hyper_params=hp, weights=synthetic_weights(bb, hp, seed=1), precision="f16x3", max_batch=B,
                    iou_threshold=0.7, overlap_nms=True)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    prop.propose_async(torch.rand((B, size, size, 3), generator=gen, device="cuda", dtype=torch.float32)); torch.cuda.synchronize()
    d = prop._bufs[0]["reg"][:B].view(B, -1, 4).clone(); s = prop._bufs[0]["cls"][:B].view(B, -1).clone()
    A = prop.total_anchors
    ob = torch.zeros((B, 300, 4), device="cuda"); osc = torch.zeros((B, 300), device="cuda")
    oi = torch.zeros((B, 300), dtype=torch.int32, device="cuda"); ov = torch.zeros((B,), dtype=torch.int32, device="cuda")
    WS = torch.empty((max(16, int(L.lib().rpn_nms_workspace_bytes(B, A, 1, 300, 300))),), dtype=torch.uint8, device="cuda")
    def run():
        L.check(L.lib().rpn_decode_nms(L.ptr(prop.anchors), L.ptr(d), prop._vptr, L.ptr(s), B, A, 300, 0.7, float("-inf"), 1,
                                       L.ptr(ob), L.ptr(osc), L.ptr(oi), L.ptr(ov), L.ptr(WS), WS.numel(), L.stream_ptr()), "nms")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("%s %dx%d B=%d A=%d: %.1f us" % (bb, size, size, B, A, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
    del prop
