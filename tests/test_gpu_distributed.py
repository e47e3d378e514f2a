"""GPU suite: the N > 1 path with world_size 2 on ONE MI355X.

Two fresh child processes (tests/dist_gpu_worker.py) each run the HIP path on their `shard_bounds` slice of `total`
images and exchange the proposal records (gloo, staged through host memory in the worker only: one GPU cannot host two
RCCL ranks).  The compacted gather must equal the single-process proposals of all `total` images BIT FOR BIT -- this runs
`Proposer.propose_distributed(total=...)`, its uneven-shard padding and the pipelined path's `rows > B` branch with
world > 1, which the CPU suite can only exercise with synthetic records."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total", [7, 8])
def test_two_ranks_on_one_gpu_match_single_process(total, tmp_path):
    sys.path.insert(0, ROOT)
    from bench import spawn_ranks
    rc, _out = spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(total), str(tmp_path)],
                           timeout=600)
    assert rc == 0
    # single process, all images in one batch (the default path is bit-identical across batch sizes)
    hp = dict(train_utils.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
    train_utils.get_hyper_params("vgg16", img_size=500, feature_map_shape=31)      # (the table is a mutated global: put it back)
    weights = synthetic_weights("vgg16", hp, seed=5)
    imgs = torch.rand((total, 160, 160, 3), generator=torch.Generator().manual_seed(11)).cuda()
    prop = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=total, iou_threshold=0.7)
    boxes, scores, valid, _ = prop.propose(imgs)
    want = prop.pack_records(boxes, scores, valid).cpu().numpy()
    assert want.shape == (total, prop.topn * 5 + 1) and (valid > 0).all()
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for key in ("serial", "serial_buf", "pipe_step1", "pipe_step2"):
            assert got[key].shape == want.shape, (rank, key, got[key].shape)
            assert np.array_equal(got[key].view(np.uint32), want.view(np.uint32)), (rank, key)


def test_two_ranks_on_one_gpu_c5_shape(tmp_path):
    """The same two-rank run at BASELINE.json configs[4]'s per-GPU shape -- MobileNetV2, 1024 x 1024, 15 anchors per cell
    (61 440 anchors), ONE image per rank: what `bench.py --gpus N`'s `multi_gpu_configs.c5` leg runs on every rank.  Each rank's
    gathered records must equal, bit for bit, what a one-image handle proposes for the two images in one process."""
    sys.path.insert(0, ROOT)
    from bench import spawn_ranks
    from tests.dist_gpu_worker import config
    rc, _out = spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "2", str(tmp_path), "gloo", "c5"],
                           timeout=900)
    assert rc == 0
    backbone, hp, size, seed = config("c5")
    assert hp["anchor_count"] == 15 and size == 1024
    weights = synthetic_weights(backbone, hp, seed=seed)
    imgs = torch.rand((2, size, size, 3), generator=torch.Generator().manual_seed(11)).cuda()
    prop = Proposer(backbone, hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=1, iou_threshold=0.7)
    assert prop.total_anchors == 61440
    rows = []
    for i in range(2):
        boxes, scores, valid, _ = prop.propose(imgs[i:i + 1])
        assert int(valid[0]) > 0
        rows.append(prop.pack_records(boxes, scores, valid).cpu().numpy())
    want = np.concatenate(rows)
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for key in ("serial", "serial_buf", "pipe_step1", "pipe_step2"):
            assert got[key].shape == want.shape, (rank, key, got[key].shape)
            assert np.array_equal(got[key].view(np.uint32), want.view(np.uint32)), (rank, key)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible devices (the 1-GPU pool skips it; the 8-GPU node runs it)")
def test_two_ranks_rccl(tmp_path):
    """The same two-rank run with each rank on a device of its own and the proposal records gathered by RCCL (backend "nccl"),
    device to device -- `propose_distributed` and two `propose_distributed_pipelined` steps + flush: the gathered records equal the
    single-process proposals bit for bit, and the NMS side stream still runs beside the conv stream once the communicator's own
    streams exist (hardware-queue aliasing).  The reference has no counterpart (utils/io_utils.py:52-59 is one
    process); the split is SURVEY.md section 8e's."""
    sys.path.insert(0, ROOT)
    from bench import spawn_ranks
    total = 7
    rc, _out = spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(total), str(tmp_path), "nccl"],
                           timeout=900)
    assert rc == 0
    hp = dict(train_utils.get_hyper_params("vgg16", img_size=160, feature_map_shape=10))
    train_utils.get_hyper_params("vgg16", img_size=500, feature_map_shape=31)
    weights = synthetic_weights("vgg16", hp, seed=5)
    imgs = torch.rand((total, 160, 160, 3), generator=torch.Generator().manual_seed(11)).cuda()
    prop = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=total, iou_threshold=0.7)
    boxes, scores, valid, _ = prop.propose(imgs)
    want = prop.pack_records(boxes, scores, valid).cpu().numpy()
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(got["side_stream_ok"][0]) == 1, rank
        for key in ("serial", "serial_buf", "pipe_step1", "pipe_step2"):
            assert np.array_equal(got[key].view(np.uint32), want.view(np.uint32)), (rank, key)
