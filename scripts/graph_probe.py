"""What a captured step would give: one hipGraph per pipeline (conv stack + decode/NMS, joined), D graphs replayed round robin on
D streams.  Probe only (torch.cuda.CUDAGraph does the capture).  usage: python scripts/graph_probe.py [c1|c5|b8] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils

hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1. / 3.]))
hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., .5]))
cfgs = {"c1": ("mobilenet_v2", hp_mn, 1), "c5": ("mobilenet_v2", hp_c5, 1), "b8": ("mobilenet_v2", hp_mn, 8)}
for tag in (sys.argv[1:] or ["c1", "c5", "b8"]):
    bb, hp, B = cfgs[tag]
    w = synthetic_weights(bb, hp, seed=1)
    x = torch.rand((B, hp["img_size"], hp["img_size"], 3), device="cuda")
    ref = Proposer(bb, hyper_params=hp, weights=w, precision="f16x3", max_batch=B, iou_threshold=0.7)
    want = [t.clone() for t in ref.propose(x)]
    for D in (1, 2, 3):
        props = [Proposer(bb, hyper_params=hp, weights=w, precision="f16x3", max_batch=B, iou_threshold=0.7, overlap_nms=True, check_range=False)
                 for _ in range(D)]
        streams = [torch.cuda.Stream() for _ in range(D)]
        graphs, outs = [], []
        for p, s in zip(props, streams):
            with torch.cuda.stream(s):
                p.propose(x); p.propose(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                o = p.propose_async(x)
                p.wait()
            graphs.append(g); outs.append(o)
        torch.cuda.synchronize()
        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % D]):
                    graphs[k % D].replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            return t1
        run(20)
        t0 = time.perf_counter(); t1 = run(400); t2 = time.perf_counter()
        ok = all(torch.equal(a, b) for o in outs for a, b in zip(o, want))
        print("%s graphs D=%d: enqueue %.4f ms/step, total %.4f ms/step (%.0f img/s) equal=%s" % (tag, D, (t1 - t0) / 400 * 1e3, (t2 - t0) / 400 * 1e3, B * 400 / (t2 - t0), ok), flush=True)
        del graphs, props
