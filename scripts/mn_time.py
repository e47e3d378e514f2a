"""MobileNetV2 step times of the three BASELINE shapes (c1 = configs[0], c5 = configs[4], b8 = configs[0] at batch 8) through
the pipelined Proposer loop bench.py's `other_configs` legs use; optional per-op event table.
usage: [RPN_HIP_LIB=...] python scripts/mn_time.py [--ops] [--only c1,c5,b8] [--steps N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils

only = None
steps = 300
for i, a in enumerate(sys.argv):
    if a == "--only": only = sys.argv[i + 1].split(",")
    if a == "--steps": steps = int(sys.argv[i + 1])
hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1. / 3.]))
hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5]))
out = []
for tag, hp, B in (("c1", hp_mn, 1), ("c5", hp_c5, 1), ("b8", hp_mn, 8)):
    if only and tag not in only: continue
    w = synthetic_weights("mobilenet_v2", hp, seed=1)
    prop = Proposer("mobilenet_v2", hyper_params=hp, weights=w, precision="f16x3", max_batch=B, iou_threshold=0.7, overlap_nms=True, check_range="--no-check" not in sys.argv)
    x = torch.rand((B, hp["img_size"], hp["img_size"], 3), device="cuda")
    best = None
    for rep in range(3):
        for _ in range(10): prop.propose_async(x)
        prop.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): prop.propose_async(x)
        prop.wait(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None else min(best, dt)
    # conv stack alone (events around whole forwards)
    st = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prop.forward(x); e1.record(); torch.cuda.synchronize()
        st.append(e0.elapsed_time(e1))
    st.sort()
    out.append("%s: %.4f ms/step (%.0f img/s)  conv stack alone %.4f ms  launches %d" % (tag, best * 1e3, B / best, st[len(st) // 2], sum(o["launches"] for o in prop.rpn_model.ops())))
    print(out[-1], flush=True)
    if "--ops" in sys.argv:
        m = prop.rpn_model
        m.set_profiling(5)
        for _ in range(5): prop.forward(x)
        torch.cuda.synchronize()
        ms, _ = m.profile_ms()
        m.set_profiling(0)
        print("   " + " ".join("%s=%.1f" % (o["name"].replace("_project", "").replace("block_", "b"), 1e3 * t) for o, t in zip(m.ops(), ms)))
    del prop
