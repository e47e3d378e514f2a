"""Throughput of D independent proposal pipelines (each its own model handle, conv stream and NMS side stream) fed alternately:
does a second / third single-image step in flight fill the chip?  usage: python scripts/depth_probe.py [c1|c5|b8|vgg1|vgg8] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils

hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1. / 3.]))
hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5]))
hp_vgg = dict(train_utils.get_hyper_params("vgg16"))
cfgs = {"c1": ("mobilenet_v2", hp_mn, 1), "c5": ("mobilenet_v2", hp_c5, 1), "b8": ("mobilenet_v2", hp_mn, 8), "vgg1": ("vgg16", hp_vgg, 1), "vgg8": ("vgg16", hp_vgg, 8)}
for tag in (sys.argv[1:] or ["c1", "c5"]):
    bb, hp, B = cfgs[tag]
    w = synthetic_weights(bb, hp, seed=1)
    for D in (1, 2, 3):
        props = [Proposer(bb, hyper_params=hp, weights=w, precision="f16x3", max_batch=B, iou_threshold=0.7, overlap_nms=True) for _ in range(D)]
        streams = [torch.cuda.Stream() for _ in range(D)]
        x = torch.rand((B, hp["img_size"], hp["img_size"], 3), device="cuda")
        torch.cuda.synchronize()
        def run(n):
            for k in range(n):
                i = k % D
                with torch.cuda.stream(streams[i]):
                    props[i].propose_async(x)
            for i in range(D):
                with torch.cuda.stream(streams[i]):
                    props[i].wait()
            torch.cuda.synchronize()
        run(30)
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); run(300); dt = (time.perf_counter() - t0) / 300
            best = dt if best is None else min(best, dt)
        print("%s depth %d: %.4f ms per step (%.0f img/s)" % (tag, D, best * 1e3, B / best), flush=True)
        del props
