"""hipGraph replay of the conv stack vs eager launches; and whether events recorded inside a captured graph can be timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.predictor import Proposer
prop = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=False)
imgs = torch.rand((8, 500, 500, 3), device="cuda")
def eager(K=60):
    for _ in range(5): prop.propose(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): prop.propose(imgs)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
print("eager serial step      %.3f ms" % eager(), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): prop.propose(imgs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        prop.propose(imgs)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): g.replay()
    torch.cuda.synchronize()
    print("graph replay step      %.3f ms" % ((time.perf_counter() - t0) / 60 * 1e3), flush=True)
    # 4 steps per graph
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, stream=s):
        for _ in range(4): prop.propose(imgs)
    for _ in range(3): g4.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): g4.replay()
    torch.cuda.synchronize()
    print("graph replay (4 steps) %.3f ms/step" % ((time.perf_counter() - t0) / 60 * 1e3), flush=True)
print("eager serial step      %.3f ms" % eager(), flush=True)
