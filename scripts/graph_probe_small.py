"""hipGraph replay of the conv stack alone vs eager launches at one image per step (MobileNetV2: configs[0], configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils
for label, kw in (("c1", {}), ("c5", dict(img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1 / 3.]))):
    hp = dict(train_utils.get_hyper_params("mobilenet_v2", **kw))
    prop = Proposer("mobilenet_v2", hyper_params=hp, precision="f16x3", max_batch=1, overlap_nms=False)
    imgs = torch.rand((1, hp["img_size"], hp["img_size"], 3), device="cuda")
    def run(fn, K=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
    print(label, "eager conv stack  %.4f ms" % run(lambda: prop.forward(imgs)), flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): prop.forward(imgs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            prop.forward(imgs)
        print(label, "graph conv stack  %.4f ms" % run(g.replay), flush=True)
    train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5])
