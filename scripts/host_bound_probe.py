"""Is a one-image step host-bound?  Times the Python / launch side of Proposer.propose_async (the loop WITHOUT a final sync) against
the whole loop with the sync: if the two are equal the GPU waits for the host.  Usage: python scripts/host_bound_probe.py [c1|c5|c2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
if cfg == "c1":
    bb, B, size = "mobilenet_v2", 1, 500; hp = dict(train_utils.get_hyper_params(bb))
elif cfg == "c5":
    bb, B, size = "mobilenet_v2", 1, 1024
    hp = dict(train_utils.get_hyper_params(bb, img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
else:
    bb, B, size = "vgg16", 8, 500; hp = dict(train_utils.get_hyper_params(bb))
prop = Proposer(bb, hyper_params=hp, weights=synthetic_weights(bb, hp, seed=1), precision="f16x3", max_batch=B, iou_threshold=0.7,
                overlap_nms=True)
imgs = torch.rand((B, size, size, 3), device="cuda")
for _ in range(20): prop.propose_async(imgs)
torch.cuda.synchronize()
N = 300
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N): prop.propose_async(imgs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host side %.1f us / step, with sync %.1f us / step" % (cfg, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N))
