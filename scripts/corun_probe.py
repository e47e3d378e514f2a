"""How much of a memory-bound side kernel (stand-in for next batch's first layer: ~512 MB written) hides beside the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.predictor import Proposer
prop = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=True)
imgs = torch.rand((8, 500, 500, 3), device="cuda")
side = torch.cuda.Stream()
src = torch.rand((64 * 1024 * 1024,), device="cuda")          # 256 MB read
dst = torch.empty((128 * 1024 * 1024,), device="cuda")        # 512 MB written
def side_kernel():
    dst[:src.numel()].copy_(src, non_blocking=True)
    dst[src.numel():].copy_(src, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): side_kernel()
e1.record(); torch.cuda.synchronize()
print("side kernel alone: %.3f ms" % (e0.elapsed_time(e1) / 10))
def run(with_side, K=60):
    for _ in range(5): prop.propose(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        prop.propose(imgs)
        if with_side:
            with torch.cuda.stream(side):
                side_kernel()
    prop.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for ws in (False, True, False, True):
    print("side=%d: %.3f ms/step" % (ws, run(ws)), flush=True)
