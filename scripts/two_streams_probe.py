"""Two independent proposal pipelines on two HIP streams (two model handles): do kernel tails / small kernels of one
overlap with the other's convs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.predictor import Proposer
imgs = torch.rand((8, 500, 500, 3), device="cuda")
def make(n):
    props, streams = [], []
    for i in range(n):
        s = torch.cuda.Stream() if n > 1 else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            props.append(Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=True))
        streams.append(s)
    return props, streams
def run(props, streams, K=60):
    n = len(props)
    for k in range(6):
        with torch.cuda.stream(streams[k % n]):
            props[k % n].propose(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k % n]):
            props[k % n].propose(imgs)
    for p, s in zip(props, streams):
        with torch.cuda.stream(s):
            p.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for n in (1, 2, 1, 2, 3):
    props, streams = make(n)
    ms = run(props, streams)
    print("%d pipeline(s): %.3f ms/step = %.0f images/s" % (n, ms, 8e3 / ms), flush=True)
    del props
    torch.cuda.synchronize()
