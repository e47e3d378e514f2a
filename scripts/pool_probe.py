"""Host enqueue time against total time of a ProposerPool loop: is the two-in-flight rate bound by the host's launch rate?
usage: python scripts/pool_probe.py [c1|c5|b8] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer, ProposerPool
from tf_rpn_amd.utils import train_utils

hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1. / 3.]))
hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5]))
cfgs = {"c1": ("mobilenet_v2", hp_mn, 1), "c5": ("mobilenet_v2", hp_c5, 1), "b8": ("mobilenet_v2", hp_mn, 8)}
for tag in (sys.argv[1:] or ["c1", "c5", "b8"]):
    bb, hp, B = cfgs[tag]
    w = synthetic_weights(bb, hp, seed=1)
    x = torch.rand((B, hp["img_size"], hp["img_size"], 3), device="cuda")
    for D in (1, 2):
        pool = ProposerPool(D, bb, hyper_params=hp, weights=w, precision="f16x3", max_batch=B, iou_threshold=0.7)
        for ordered in (True, False):
            def run(n):
                for _ in range(n):
                    pool.propose_async(x, ordered=ordered)
                t1 = time.perf_counter()
                pool.wait()
                torch.cuda.synchronize()
                return t1
            run(20)
            t0 = time.perf_counter()
            t1 = run(400)
            t2 = time.perf_counter()
            print("%s D=%d ordered=%d: enqueue %.4f ms/step, total %.4f ms/step" % (tag, D, ordered, (t1 - t0) / 400 * 1e3, (t2 - t0) / 400 * 1e3), flush=True)
        del pool
