#!/usr/bin/env python
"""bench.py -- proposal images/sec of the RPN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (predictor.py:50-56 + NMS(300)) over one batch of synthetic
500x500x3 images per GPU: conv backbone -> RPN head -> fused delta-decode + NMS -> (N > 1) one RCCL
all-gather of the proposal records.  Inputs are resident in HBM before the timed region.
At N = 1 the workload is BASELINE.json configs[1] (batch 8, VGG16 backbone + RPN head).
Images shard over ranks (weak scaling: the per-GPU batch is fixed), no other data-path collective.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (f16x3 / bf16x3: the persistent LDS-DMA
16x16x32-MFMA conv; f32: the 128x128 f32-MFMA implicit-GEMM conv), found by an untimed pre-pass and then timed
live with HIP events recorded on the launch stream around its launches in the K timed steps (one launch per step,
round robin, when K samples every launch at least twice; else all of them);
`cpu_baseline` is the CPU oracle (torch-CPU conv stack + plain-C decode/NMS restatement: a "port", the
TF2 reference cannot run here) timed on this box's host cores on a bounded sample, rank 0 at N = 1 only.
The N = 1 line also carries `c3` (BASELINE.json configs[2]: batch 64, box path only -- decode / IoU-map GB/s and the
metric's "NMS boxes/sec", every kernel timed with HIP events in this run) and `exact_f32` (the same workload on the
parity-clean exact-float32 MFMA path with its own roofline fraction).

`python bench.py --gpus N` with N > 1 and no launcher starts one child process per GPU itself (`spawn_ranks`) and
forwards rank 0's JSON line; under `torch.distributed.run` (WORLD_SIZE set) it is a plain rank.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0 / 3.0, "f16x3": 2500.0 / 3.0}   # MI355X_MICROARCH.md: f32 MFMA; dense 16-bit MFMA / 3 products


def spawn_ranks(n, cmd, env=None, timeout=None, poll=0.2):
    """Run `cmd` n times as child processes, one per rank (RANK / LOCAL_RANK = 0..n-1, WORLD_SIZE = n, MASTER_ADDR =
    127.0.0.1 and a free MASTER_PORT in their environment), the way `python -m torch.distributed.run --nproc-per-node n`
    would.  The parent never touches a GPU (a process that has initialised HIP must not exec or fork workers on this
    pool).  Rank 0's stdout is captured (through a temporary file, so a chatty child can never block on a full pipe)
    and returned; the other ranks' stdout goes to stderr.  When a child fails or the timeout expires the others are
    terminated by PID.  Returns (exit code, rank-0 stdout): 0 only if every rank exited 0; 124 on timeout."""
    import tempfile
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    with tempfile.TemporaryFile(mode="w+") as cap:
        for r in range(n):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen(cmd, env=e, stdout=cap if r == 0 else sys.stderr))
        deadline = None if timeout is None else time.time() + timeout
        rc = 0
        try:
            while rc == 0:
                codes = [p.poll() for p in procs]
                failed = [c for c in codes if c not in (None, 0)]
                if failed:
                    rc = failed[0]
                elif all(c == 0 for c in codes):
                    break
                elif deadline is not None and time.time() > deadline:
                    rc = 124
                else:
                    time.sleep(poll)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        cap.seek(0)
        return rc, cap.read()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start one process per GPU.  Nothing here initialises HIP:
    torch.cuda.device_count() only counts devices."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but this machine exposes %d HIP device(s)\n" % (args.gpus, have))
        return 2
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    rc, out = spawn_ranks(args.gpus, cmd, timeout=float(os.environ.get("BENCH_SPAWN_TIMEOUT", "1700")))
    sys.stdout.write(out)
    sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: a rank exited with code %d\n" % rc)
    return rc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="BASELINE.json config: c2 = configs[1] (default: VGG16, 500x500, batch 8 per GPU); c4 = configs[3] "
                         "(VGG16, 32 images per GPU = batch 256 on 8 GPUs); c5 = configs[4] (MobileNetV2, 1024x1024, 15 "
                         "anchors per cell, 1 image per GPU = batch 8 on 8 GPUs)")
    ap.add_argument("--backbone", default=None, choices=["vgg16", "mobilenet_v2"])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16x3", "f16x3"],
                    help="conv arithmetic: exact f32 MFMA, or f32 operands carried as hi+lo bf16/f16 halves with 3 MFMAs per "
                         "product and f32 accumulation (DESIGN.md 4.1: f16x3 measures as accurate as exact f32)")
    ap.add_argument("--iou-threshold", type=float, default=0.7)
    ap.add_argument("--serial-nms", action="store_true",
                    help="run decode+NMS on the conv stream (default: NMS of step k overlaps the convs of step k+1 on a "
                         "second HIP stream; every step's work still completes inside the timed region)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise a world-size-1 RCCL group and run the N > 1 code path (record packing + all-gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the configs[2] box-path leg (`c3`) and the exact-float32 leg (`exact_f32`) of the N = 1 line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    return ap.parse_args()


def cpu_baseline(backbone, hyper_params, weights, iou_threshold, target_seconds):
    """CPU oracle ("port") on this box's host cores: whole path for one image at a time."""
    import numpy as np
    import torch

    from oracle import bbox_oracle as bo
    from oracle import c_oracle as co
    from oracle import conv_oracle as cv
    rng = np.random.RandomState(0)
    anchors = bo.generate_anchors(hyper_params)
    var = np.float32(hyper_params["variances"])
    size = hyper_params["img_size"]

    def one_image():
        img = rng.uniform(0, 1, size=(1, size, size, 3)).astype(np.float32)
        reg, cls = cv.rpn_forward(backbone, img, weights)
        boxes = co.decode(anchors, reg.reshape(1, -1, 4), var)
        co.combined_nms(boxes[:, :, None, :], cls.reshape(1, -1, 1), 300, 300, iou_threshold=iou_threshold)

    # thread count: oneDNN collapses when oversubscribed (256 threads on the 2x64-core host: 12 s/image,
    # 16 threads: 0.2 s/image), so time one image at a few counts and keep the fastest
    best = None
    for n_threads in sorted({min(os.cpu_count() or 1, n) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(n_threads)
        one_image()                                # warm-up (oneDNN primitive caches)
        t1 = time.perf_counter()
        one_image()
        dt1 = time.perf_counter() - t1
        if best is None or dt1 < best[1]:
            best = (n_threads, dt1)
    cores = best[0]
    torch.set_num_threads(cores)
    one_image()
    t0 = time.perf_counter()
    n = 0
    while True:
        one_image()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or n >= 64:
            break
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d images of the same workload, one at a time (torch-CPU f32 conv stack with %d threads + "
                      "plain-C decode/NMS(300), single thread), %.1f s" % (n, cores, dt),
            "cpu": model_name}


def measured_traffic(kernel, precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, written by scripts/make_traffic.py under the same kernel names as this file's
    per-op table: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs of this same command) -- PMC
    counters cannot be collected from inside an un-profiled run, so the figure is read back from the newest committed
    profile and labelled with its file name (`traffic_source`).  (None, None) when no profile matches."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % precision)))
    for path in reversed(paths):
        try:
            with open(path) as f:
                return round(json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def exact_f32_leg(backbone, hp, weights, B, iou_threshold, imgs, steps=5, warmup=2):
    """The parity-clean arithmetic beside the headline (report both): the same workload with every conv
    on the exact float32 MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an ordered fmaf chain), timed over `steps` steps in
    this same run, with the dominant kernel's fraction of the 157.3 TFLOP/s f32-MFMA peak from a per-op event pass."""
    import torch

    from tf_rpn_amd.predictor import Proposer
    prop = Proposer(backbone, hyper_params=hp, weights=weights, precision="f32", max_batch=B,
                    iou_threshold=iou_threshold, overlap_nms=True)
    for _ in range(warmup):
        prop.propose_async(imgs)
    prop.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        prop.propose_async(imgs)
    prop.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model = prop.rpn_model
    ops = model.ops()
    model.set_profiling(3)
    for _ in range(3):
        prop.propose_async(imgs)
    prop.wait()
    torch.cuda.synchronize()
    ms, _ = model.profile_ms()
    model.set_profiling(0)
    tot = {}
    for op, t in zip(ops, ms):
        d = tot.setdefault(op["kernel"], [0.0, 0.0, 0])
        d[0] += t
        d[1] += op["flops_per_image"] * B
        d[2] += 1
    dom = max(tot, key=lambda k: tot[k][0])
    achieved = tot[dom][1] / (tot[dom][0] * 1e-3) / 1e12
    del prop
    torch.cuda.empty_cache()
    return {"value": round(B * steps / dt, 2), "unit": "images/s", "ms_per_step": round(1e3 * dt / steps, 4),
            "steps": steps, "dtype": "f32",
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_TFLOPS["f32"],
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS["f32"], 4),
                         "launches_per_step": tot[dom][2], "avg_launch_ms": round(tot[dom][0] / tot[dom][2], 4)}}


def config_leg(label, backbone, hp, B, precision, iou_threshold, steps=10, warmup=3):
    """One of the other BASELINE.json configs at its per-GPU batch, in this same run: whole hot path (conv stack ->
    head -> decode -> NMS(300), NMS of step k overlapped with the convs of step k+1), synthetic images and weights of
    that config's shape, `steps` timed steps bracketed by synchronize()."""
    import torch

    from tf_rpn_amd.models._rpn_model import synthetic_weights
    from tf_rpn_amd.predictor import Proposer
    weights = synthetic_weights(backbone, hp, seed=1)
    prop = Proposer(backbone, hyper_params=hp, weights=weights, precision=precision, max_batch=B,
                    iou_threshold=iou_threshold, overlap_nms=True)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    imgs = torch.rand((B, hp["img_size"], hp["img_size"], 3), generator=gen, device="cuda", dtype=torch.float32)
    for _ in range(warmup):
        prop.propose_async(imgs)
    prop.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        prop.propose_async(imgs)
    prop.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # latency of ONE step end to end (images resident -> proposals resident, nothing overlapped across steps, everything
    # on one stream = Proposer(overlap_nms=False)): HIP events around propose() on an idle device, median of `steps`
    # repeats -- what configs quoted per image (c1, c5) mean by it
    prop.overlap_nms = False
    lat = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        prop.propose(imgs)
        e1.record()
        torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat.sort()
    n_launches = len(prop.rpn_model.ops())
    del prop, imgs
    torch.cuda.empty_cache()
    return {"workload": "%s: %s, %dx%d, %d anchors/cell, batch %d per GPU" % (label, backbone, hp["img_size"], hp["img_size"],
                                                                             hp["anchor_count"], B),
            "value": round(B * steps / dt, 2), "unit": "images/s", "ms_per_step": round(1e3 * dt / steps, 4),
            "ms_per_image": round(1e3 * dt / steps / B, 4), "latency_ms_one_step_unpipelined": round(lat[len(lat) // 2], 4),
            "steps": steps, "dtype": precision, "conv_launches_per_step": n_launches}


def c3_leg(hp, n=30):
    """BASELINE.json configs[2]: batch 64, box path only (no backbone) -- the metric's "NMS boxes/sec".  Inputs as
    SURVEY.md 8(d): deltas ~ N(0,1) (x variances inside the kernel), scores = a seeded permutation of (i + 0.5) / A per
    image (tie-free), gt = 42 rows per image of which the first 10 are boxes and the rest zero padding.  Every kernel
    is timed with HIP events on the launch stream (torch's current stream) over `n` back-to-back launches; bandwidths
    are ALGORITHMIC bytes / time against the 8 TB/s HBM peak."""
    import numpy as np
    import torch

    from tf_rpn_amd import _lib as L
    from tf_rpn_amd.utils import bbox_utils
    B, G = 64, 42
    anchors = bbox_utils.generate_anchors(hp)
    A = int(anchors.shape[0])
    deltas = torch.from_numpy(np.random.RandomState(2).standard_normal((B, A, 4)).astype(np.float32)).cuda()
    rng = np.random.RandomState(3)
    base = ((np.arange(A) + 0.5) / A).astype(np.float32)
    scores = torch.from_numpy(np.stack([base[rng.permutation(A)] for _ in range(B)])).cuda()
    rng = np.random.RandomState(4)
    gt_np = np.zeros((B, G, 4), np.float32)
    y1, x1 = rng.uniform(0, 0.7, size=(2, B, 10))
    h, w = rng.uniform(0.05, 0.3, size=(2, B, 10))
    gt_np[:, :10] = np.stack([y1, x1, y1 + h, x1 + w], axis=-1)
    gt = torch.from_numpy(gt_np).cuda()
    boxes = torch.empty((B, A, 4), device="cuda")
    iou = torch.empty((B, A, G), device="cuda")
    ob, osc = torch.zeros((B, 300, 4), device="cuda"), torch.zeros((B, 300), device="cuda")
    oi = torch.zeros((B, 300), dtype=torch.int32, device="cuda")
    ov = torch.zeros((B,), dtype=torch.int32, device="cuda")
    _keep, vptr = L.host_floats(hp["variances"])
    lib = L.lib()

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {"workload": "configs[2]: batch=64, %d anchors per image, G=42 gt rows, no backbone" % A, "B": B, "A": A, "G": G,
           "hbm_peak_GBps": 8000.0}
    t = timed(lambda: L.check(lib.rpn_decode(L.ptr(anchors), 0, L.ptr(deltas), vptr, B, A, L.ptr(boxes), L.stream_ptr()),
                              "rpn_decode"), n)
    by = 32.0 * B * A + 16.0 * A
    out["decode"] = {"us": round(t * 1e6, 2), "GBps": round(by / t / 1e9, 1), "frac": round(by / t / 8e12, 4), "bytes": by}
    t = timed(lambda: L.check(lib.rpn_iou_map(L.ptr(anchors), 0, A, L.ptr(gt), B, G, L.ptr(iou), L.stream_ptr()),
                              "rpn_iou_map"), n)
    by = 4.0 * B * A * G + 16.0 * (A + B * G)
    out["iou_map"] = {"us": round(t * 1e6, 2), "GBps": round(by / t / 1e9, 1), "frac": round(by / t / 8e12, 4), "bytes": by}
    for thr in (0.5, 0.7):
        t = timed(lambda: L.check(lib.rpn_decode_nms(L.ptr(anchors), L.ptr(deltas), vptr, L.ptr(scores), B, A, 300, thr,
                                                     float("-inf"), 1, L.ptr(ob), L.ptr(osc), L.ptr(oi), L.ptr(ov), L.vp(0),
                                                     0, L.stream_ptr()), "rpn_decode_nms"), max(5, n // 2))
        by = B * (20.0 * A + 7204.0)
        out["decode_nms_iou%.1f" % thr] = {"us": round(t * 1e6, 2), "boxes_per_sec": round(B * A / t, 1),
                                           "GBps": round(by / t / 1e9, 2), "bytes": by,
                                           "mean_valid": round(float(ov.float().mean().item()), 1)}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher (one child per GPU) BEFORE anything initialises HIP in this process
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist

    from tf_rpn_amd import _lib
    from tf_rpn_amd.models._rpn_model import synthetic_weights
    from tf_rpn_amd.predictor import Proposer
    from tf_rpn_amd.utils import train_utils

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    if args.backbone is None:
        args.backbone = "mobilenet_v2" if args.config == "c5" else "vgg16"
    if args.batch is None:
        args.batch = {"c2": 8, "c4": 32, "c5": 1}[args.config]
    if args.config == "c5":     # 3 scales x 5 aspect ratios; BASELINE.json names no ratio set: {1, 2, 1/2, 3, 1/3}
        hp = dict(train_utils.get_hyper_params(args.backbone, img_size=1024, feature_map_shape=64,
                                               anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
    else:
        hp = dict(train_utils.get_hyper_params(args.backbone))
    weights = synthetic_weights(args.backbone, hp, seed=1)
    overlap = not args.serial_nms                     # 2-stage pipeline across steps (see Proposer.overlap_nms)
    prop = Proposer(args.backbone, hyper_params=hp, weights=weights, precision=args.precision,
                    max_batch=args.batch, iou_threshold=args.iou_threshold, overlap_nms=overlap)
    B = args.batch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(rank)
    imgs = torch.rand((B, hp["img_size"], hp["img_size"], 3), generator=gen, device="cuda", dtype=torch.float32)
    if world > 1 or args.force_dist:
        # The communicator is created AFTER the proposer's streams have run once: created first (eagerly, with
        # device_id), RCCL's internal streams take the hardware queues and the NMS side stream ends up sharing a
        # queue with the conv stream -- measured: the NMS/conv overlap disappears (3.25 vs 3.05 ms/step).
        prop.propose(imgs)
        torch.cuda.synchronize()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner through C stdio when the communicator is created (flushed at exit when stdout
        # is a pipe, i.e. AFTER the JSON line): send it to stderr so that stdout carries the one JSON line only.
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    M = prop.topn
    gathered = torch.empty((world * B, M * 5 + 1), dtype=torch.float32, device="cuda") if world > 1 else None
    gather_bufs = [torch.empty((world * B, M * 5 + 1), dtype=torch.float32, device="cuda") for _ in range(2)]
    use_dist = world > 1 or args.force_dist

    def step():
        if overlap and use_dist:
            return prop.propose_distributed_pipelined(imgs, gather_bufs)
        if overlap:
            return prop.propose_async(imgs)
        return prop.propose_distributed(imgs, gather_out=gathered)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides -------------------
    model = prop.rpn_model
    ops = model.ops()

    def dominant_kernel(ms_per_op):               # the conv kernel template with the largest total time
        tot = {}
        for op, ms in zip(ops, ms_per_op):
            tot[op["kernel"]] = tot.get(op["kernel"], 0.0) + ms
        return max(tot, key=tot.get)

    # untimed pre-pass: every op timed, to find the dominant kernel.  The timed region then carries HIP events only
    # around ITS launches (2 events per launch on the launch stream; events around all 20 ops cost ~2.5 % of a step).
    model.set_profiling(2)
    for _ in range(2):
        step()
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)
    torch.cuda.synchronize()
    dom = dominant_kernel(model.profile_ms()[0])
    dom_mask = [op["kernel"] == dom for op in ops]
    model.set_profiling_mask(dom_mask)
    # one of its launches per step, round robin (2 events per step), when K gives every launch >= 2 samples;
    # otherwise all of them in every step
    rotate = args.steps >= 2 * sum(dom_mask)
    model.set_profiling_rotate(rotate)
    model.set_profiling(args.steps)
    nms_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if overlap and use_dist:                  # + all-gather of step k-1 behind the convs of step k
            prop.propose_distributed_pipelined(imgs, gather_bufs)
            continue
        if overlap:
            prop.propose_async(imgs)              # convs on this stream, decode+NMS on the side stream
            continue
        deltas, scores = prop.forward(imgs)
        nms_ev[k][0].record()                     # same stream as the launches (torch's current stream)
        prop_out = prop._boxes[:B], prop._scores[:B], prop._idx[:B], prop._valid[:B]
        st = _lib.lib().rpn_decode_nms(_lib.ptr(prop.anchors), _lib.ptr(deltas), prop._vptr, _lib.ptr(scores), B,
                                       prop.total_anchors, M, prop.iou_threshold, prop.score_threshold, 1,
                                       _lib.ptr(prop_out[0]), _lib.ptr(prop_out[1]), _lib.ptr(prop_out[2]),
                                       _lib.ptr(prop_out[3]), _lib.vp(0), 0, _lib.stream_ptr())
        _lib.check(st, "rpn_decode_nms")
        nms_ev[k][1].record()
        if world > 1:
            rec = prop.pack_records(prop_out[0], prop_out[1], prop_out[3], out=prop._record_buffer(2, B))
            dist.all_gather_into_tensor(gathered, rec)
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)       # the last step's NMS + all-gather finish inside the timed region
    elif overlap:
        prop.wait()                               # the last step's NMS must finish inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    # mean durations of the dominant kernel's launches over the K timed steps (events were recorded on the launch
    # stream, read here); the other ops' durations come from an untimed post-pass with every op timed
    dom_ms, _kept = model.profile_ms()
    model.set_profiling_rotate(False)
    model.set_profiling_mask(None)
    model.set_profiling(3)
    for _ in range(3):
        step()
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)
    torch.cuda.synchronize()
    all_ms, _ = model.profile_ms()
    last_ms = [d if op["kernel"] == dom else a for op, d, a in zip(ops, dom_ms, all_ms)]
    model.set_profiling(0)
    if overlap:                                   # time decode+NMS alone, after the region, for the report
        torch.cuda.synchronize()
        d_, s_ = prop._bufs[0]["reg"][:B].view(B, -1, 4), prop._bufs[0]["cls"][:B].view(B, -1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(_lib.lib().rpn_decode_nms(_lib.ptr(prop.anchors), _lib.ptr(d_), prop._vptr, _lib.ptr(s_), B,
                                                 prop.total_anchors, M, prop.iou_threshold, prop.score_threshold, 1,
                                                 _lib.ptr(prop._boxes[:B]), _lib.ptr(prop._scores[:B]),
                                                 _lib.ptr(prop._idx[:B]), _lib.ptr(prop._valid[:B]), _lib.vp(0), 0,
                                                 _lib.stream_ptr()), "rpn_decode_nms")
        e1.record()
        torch.cuda.synchronize()
        nms_ms = e0.elapsed_time(e1) / 5
    else:
        nms_ms = sum(a.elapsed_time(b) for a, b in nms_ev) / args.steps

    gather_ms = None
    if use_dist:                                  # the all-gather alone, after the region, for the report
        src = prop._bufs[0] if overlap else {"boxes": prop._boxes, "scores": prop._scores, "valid": prop._valid}
        rec = prop.pack_records(src["boxes"][:B], src["scores"][:B], src["valid"][:B])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_gather_into_tensor(gather_bufs[0], rec)
        e1.record()
        torch.cuda.synchronize()
        gather_ms = e0.elapsed_time(e1) / 10

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    if rank == 0:
        # dominant kernel: the conv instantiation with the largest total time
        by_kernel = {}
        for op, ms in zip(ops, last_ms):
            d = by_kernel.setdefault(op["kernel"], {"ms": 0.0, "flops": 0.0, "launches": 0})
            d["ms"] += ms
            d["flops"] += op["flops_per_image"] * B
            d["launches"] += 1
        d = by_kernel[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # the peak of the arithmetic the DOMINANT KERNEL runs in: split-precision 3x3 kernels -> 16-bit MFMA / 3 products;
        # everything else (float32 implicit GEMM, the fused MobileNetV2 blocks) -> the float32 MFMA
        peak = PEAK_TFLOPS[args.precision] if "split" in dom else PEAK_TFLOPS["f32"]
        traffic, traffic_source = measured_traffic(dom, args.precision)
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "launches_per_step": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                    "flops_per_launch": d["flops"] / d["launches"],
                    "timing": ("HIP events on the launch stream around this kernel's launches in the K timed steps"
                               + (": one launch per step, round robin (every launch sampled >= 2 times)" if rotate else "")),
                    "conv_stack_ms": round(sum(last_ms), 3), "decode_nms_ms": round(nms_ms, 4)}
        out = {
            "metric": "proposal images/sec at 500x500x3 VOC batch; NMS boxes/sec",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "dtype_note": {"f32": "float32 in, float32 MFMA accumulate (exact)",
                           "f16x3": "float32 operands as hi+lo float16 halves, 3 MFMAs per product, float32 accumulate",
                           "bf16x3": "float32 operands as hi+lo bfloat16 halves, 3 MFMAs per product, float32 accumulate"
                           }[args.precision],
            "config": {"workload": "configs[%d]: %dx%dx3 synthetic batch=%d per GPU, %s backbone + RPN head, %d anchors/cell, "
                                   "decode + NMS(300), %dxMI355X" % ({"c2": 1, "c4": 3, "c5": 4}[args.config], hp["img_size"],
                                                                      hp["img_size"], B, args.backbone, hp["anchor_count"],
                                                                      world),
                       "per_gpu_batch": B, "global_batch": world * B, "img_size": hp["img_size"],
                       "anchors_per_image": prop.total_anchors, "nms_topn": M, "iou_threshold": args.iou_threshold,
                       "weights": "random-init (seeded He-normal)",
                       "parallelism": "image-sharded dp%d, one RCCL all-gather of proposals" % world,
                       "nms_overlap": ("decode+NMS of step k on a second HIP stream, overlapping the convs of step k+1"
                                       + ("; all-gather of step k-1 issued behind the convs of step k" if use_dist else ""))
                                      if overlap else "serial on the conv stream"},
            "allgather_ms": None if gather_ms is None else round(gather_ms, 4),
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "roofline": roofline,
        }
        if world == 1 and not args.no_extra_legs:
            # the metric's second half ("NMS boxes/sec") is quoted on configs[2] (B = 64, box path only)
            out["c3"] = c3_leg(hp)
            out["nms_boxes_per_sec"] = out["c3"]["decode_nms_iou0.7"]["boxes_per_sec"]
            out["nms_boxes_per_sec_note"] = "configs[2] (B=64), fused decode+NMS(300), iou 0.7; `c3` has iou 0.5 and the GB/s"
            if args.precision != "f32":
                out["exact_f32"] = exact_f32_leg(args.backbone, hp, weights, B, args.iou_threshold, imgs)
            if args.config == "c2" and args.backbone == "vgg16" and B == 8:
                # the other BASELINE.json configs (per-GPU shapes), so that every config has a driver-run figure
                hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2"))
                hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                                          anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
                out["other_configs"] = {
                    "c1": config_leg("configs[0]", "mobilenet_v2", hp_mn, 1, args.precision, args.iou_threshold, steps=20),
                    "c4": config_leg("configs[3] (batch 256 on 8 GPUs)", "vgg16", hp, 32, args.precision, args.iou_threshold,
                                     steps=5, warmup=2),
                    "c5": config_leg("configs[4] (batch 8 on 8 GPUs)", "mobilenet_v2", hp_c5, 1, args.precision,
                                     args.iou_threshold, steps=20),
                    "mobilenet_v2_b8": config_leg("configs[0] shape at batch 8", "mobilenet_v2", hp_mn, 8, args.precision,
                                                  args.iou_threshold, steps=20),
                }
        else:
            out["nms_boxes_per_sec"] = round(B * prop.total_anchors / (nms_ms * 1e-3), 1)
            out["nms_boxes_per_sec_note"] = "this run's own decode+NMS at its per-GPU batch (%d images)" % B
        if args.layers:
            for op, ms in zip(ops, last_ms):
                tf = op["flops_per_image"] * B / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                gbs = op["bytes_per_image"] * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                print("%-28s %-34s %8.3f ms %8.2f TF/s %9.1f GB/s" % (op["name"], op["kernel"], ms, tf, gbs),
                      file=sys.stderr)
            print("%-28s %-34s %8.3f ms" % ("decode+nms", "nms_kernel<decode>", nms_ms), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.backbone, hp, weights, args.iou_threshold, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
