"""Child process of tests/test_gpu_distributed.py: one rank of a world-size-2 run of the image-sharded proposal path on
ONE GPU.  Each rank runs the HIP path on its `shard_bounds` slice; the records are exchanged under the gloo backend,
staged through host memory HERE ONLY (the product calls torch.distributed.all_gather_into_tensor on device tensors, which
is RCCL on a real multi-GPU node; this machine has one GPU, and RCCL does not run two ranks on one device).

With a third argument "nccl" (tests/test_gpu_distributed.py::test_two_ranks_rccl, only where >= 2 devices are visible) each rank
takes the device LOCAL_RANK and the records travel by RCCL, device to device, exactly as in the product.

With a fourth argument "c5" the ranks run BASELINE.json configs[4]'s per-GPU shape (MobileNetV2, 1024 x 1024, 15 anchors per
cell, ONE image per rank when total = world) instead of the small VGG16 graph.

argv: total out_dir [gloo|nccl] [vgg160|c5]      env: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT
Writes out_dir/rank<r>.npz with the gathered records of the three code paths; exit code 0 when it ran to the end.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer, compact_gathered, shard_bounds
from tf_rpn_amd.utils import train_utils


def config(name):
    """(backbone, hyper_params, image size, weight seed) of the worker's two graphs; the test builds the same."""
    if name == "c5":
        hp = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                               anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
        train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32, anchor_ratios=[1., 2., 1. / 2.])   # (mutated global: put it back)
        return "mobilenet_v2", hp, 1024, 5
    hp = dict(train_utils.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
    train_utils.get_hyper_params("vgg16", img_size=500, feature_map_shape=31)
    return "vgg16", hp, 160, 5


def main():
    total, out_dir = int(sys.argv[1]), sys.argv[2]
    backbone, hp, size, seed = config(sys.argv[4] if len(sys.argv) > 4 else "vgg160")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = len(sys.argv) > 3 and sys.argv[3] == "nccl"
    device = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    if rccl:
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    real_all_gather = dist.all_gather_into_tensor

    def staged_all_gather(out, inp, *a, **k):       # device -> host -> gloo -> host -> device (the test's stand-in for RCCL)
        torch.cuda.current_stream().synchronize()
        host_out = torch.empty(out.shape, dtype=out.dtype)
        real_all_gather(host_out, inp.detach().cpu().contiguous(), *a, **k)
        out.copy_(host_out.to(out.device))

    if not rccl:
        dist.all_gather_into_tensor = staged_all_gather
    try:
        torch.cuda.set_device(device)
        weights = synthetic_weights(backbone, hp, seed=seed)
        imgs = torch.rand((total, size, size, 3), generator=torch.Generator().manual_seed(11))     # same on every rank
        lo, hi = shard_bounds(total, world, rank)
        local = imgs[lo:hi].cuda().contiguous()
        rows = -(-total // world)
        res = {}
        # (1) the serial distributed step with uneven shards: propose_distributed(total=...)
        prop = Proposer(backbone, hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=rows, iou_threshold=0.7)
        res["serial"] = prop.propose_distributed(local, total=total).cpu().numpy()
        # (2) the same with a caller-provided gather buffer
        gbuf = torch.full((world * rows, prop.topn * 5 + 1), 7.0, device="cuda")
        res["serial_buf"] = prop.propose_distributed(local, gather_out=gbuf, total=total).cpu().numpy()
        # (3) the pipelined step (NMS + packing + gather on the side stream), rows > B on the short rank: two steps + flush
        prop2 = Proposer(backbone, hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=rows, iou_threshold=0.7,
                         overlap_nms=True)
        gather_bufs = [torch.full((world * rows, prop2.topn * 5 + 1), 9.0, device="cuda") for _ in range(2)]
        first = prop2.propose_distributed_pipelined(local, gather_bufs)
        assert first is None
        second = prop2.propose_distributed_pipelined(local, gather_bufs)
        res["pipe_step1"] = compact_gathered(second, total, world).cpu().numpy()
        last = prop2.flush_distributed(gather_bufs)
        res["pipe_step2"] = compact_gathered(last, total, world).cpu().numpy()
        if rccl:        # the communicator's streams exist now: the NMS side stream must still run beside the conv stream
            res["side_stream_ok"] = np.array([1 if prop2.ensure_side_stream() else 0])
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
        dist.barrier()
    finally:
        dist.all_gather_into_tensor = real_all_gather
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
