"""Does the mere existence of an RCCL communicator slow the (non-collective) step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tf_rpn_amd.predictor import Proposer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
prop = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=True)
imgs = torch.rand((8, 500, 500, 3), device="cuda")
def run(K=40):
    for _ in range(3):
        prop.propose(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        prop.propose(imgs)
    prop.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
print("before init_process_group      %.3f ms/step" % run(), flush=True)
eager = os.environ.get("EAGER", "0") == "1"
if eager:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
print("after init (eager=%d)            %.3f ms/step" % (eager, run()), flush=True)
x = torch.ones(8, device="cuda"); y = torch.empty(8, device="cuda")
dist.all_gather_into_tensor(y, x); torch.cuda.synchronize()
print("after first collective         %.3f ms/step" % run(), flush=True)
print("again                          %.3f ms/step" % run(), flush=True)
dist.destroy_process_group()
print("after destroy_process_group    %.3f ms/step" % run(), flush=True)
