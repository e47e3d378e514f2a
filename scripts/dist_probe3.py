"""Order of communicator creation vs. model creation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tf_rpn_amd.predictor import Proposer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
mode = os.environ.get("MODE", "A")
def run(prop, imgs, K=40):
    for _ in range(3):
        prop.propose(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        prop.propose(imgs)
    prop.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
if mode == "A":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif mode == "C":
    dist.init_process_group("nccl", rank=0, world_size=1)
elif mode == "D":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.ones(8, device="cuda"); y = torch.empty(8, device="cuda")
    dist.all_gather_into_tensor(y, x); torch.cuda.synchronize()
prop = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=True)
imgs = torch.rand((8, 500, 500, 3), device="cuda")
print("mode %s: %.3f ms/step" % (mode, run(prop, imgs)), flush=True)
prop2 = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=False)
print("mode %s serial-nms: %.3f ms/step" % (mode, run(prop2, imgs)), flush=True)
