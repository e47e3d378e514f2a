"""CPU suite, part 3: the N>1 path under gloo, world_size 2.

Images shard contiguously over ranks with no data-path collective; the only exchange is one
all-gather of fixed-size proposal records.  The GPU kernels cannot run here, so each rank fills
its records with a deterministic function of the global image index and the test checks that the
gathered tensor is the full batch in order.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tf_rpn_amd.predictor import Proposer, compact_gathered, pad_records, shard_bounds

M = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _records_for(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float32)
    boxes = idx.view(-1, 1, 1).expand(-1, M, 4) + torch.arange(4, dtype=torch.float32) / 8
    scores = idx.view(-1, 1).expand(-1, M) / 100
    valid = (torch.arange(lo, hi) % (M + 1)).to(torch.int32)
    return Proposer.pack_records(None, boxes.contiguous(), scores.contiguous(), valid)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(total, world, rank)
        rows = -(-total // world)                     # uneven shards: every rank contributes ceil(total / world) rows
        rec = pad_records(_records_for(lo, hi), rows)
        out = torch.empty((world * rows, rec.shape[1]))
        dist.all_gather_into_tensor(out, rec)
        out = compact_gathered(out, total, world)
        ok = torch.equal(out, _records_for(0, total))
        b, s, v = Proposer.unpack_records(out, M)
        ok = ok and b.shape == (total, M, 4) and v.dtype == torch.int32 and int(v[total - 1]) == (total - 1) % (M + 1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("total", [8, 7, 1])
def test_sharded_proposals_all_gather_world2(total):
    """Even (8 -> 4 + 4) and uneven (7 -> 4 + 3, 1 -> 1 + 0) contiguous shards: the all-gather needs equal sizes, so
    short shards are padded with valid = 0 records that are dropped after the gather."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
