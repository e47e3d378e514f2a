"""Where does the N>1 step lose time?  Variants of the pipelined step on one GPU with a world-size-1 RCCL group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tf_rpn_amd.predictor import Proposer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
prop = Proposer("vgg16", precision="f16x3", max_batch=8, overlap_nms=True)
imgs = torch.rand((8, 500, 500, 3), device="cuda")
M = prop.topn
bufs = [torch.empty((8, M * 5 + 1), device="cuda") for _ in range(2)]
def run(mode, K=20):
    for _ in range(3):
        step(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        step(mode)
    prop.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
def step(mode):
    prop.propose(imgs)
    if mode == 0:
        return
    buf, B = prop._last, 8
    with torch.cuda.stream(prop._nms_stream):
        rec = prop.pack_records(buf["boxes"][:B], buf["scores"][:B], buf["valid"][:B])
        if mode == 2:
            bufs[0].copy_(rec)
        elif mode == 3:
            dist.all_gather_into_tensor(bufs[0], rec)
        elif mode == 4:
            w = dist.all_gather_into_tensor(bufs[0], rec, async_op=True)
for mode, name in ((0, "propose only"), (1, "+pack (side stream)"), (2, "+pack+copy"), (3, "+pack+all_gather"), (4, "+pack+all_gather async"), (0, "propose only")):
    print("%-26s %.3f ms/step" % (name, run(mode)), flush=True)
dist.destroy_process_group()
