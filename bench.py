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

PEAK_TFLOPS = {"f32": 157.3, "f32w": 157.3, "bf16x3": 2500.0 / 3.0, "f16x3": 2500.0 / 3.0}   # MI355X_MICROARCH.md: f32 MFMA; dense 16-bit MFMA / 3 products


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


def count_gpus_without_hip():
    """Number of GPUs this process may use, without touching the HIP runtime: the KFD topology nodes that have SIMDs
    (CPU nodes report simd_count 0), cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when
    one of them is set.  None when the topology cannot be read."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    try:
        for path in nodes:
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += 1 if int(line.split()[1]) > 0 else 0
                        break
    except (OSError, ValueError, IndexError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start one process per GPU.  Nothing here may initialise HIP
    (a process that has must not start GPU workers on this pool), so the devices are counted from the kernel driver's
    topology in sysfs; only when that is unreadable does it fall back to torch.cuda.device_count()."""
    have = count_gpus_without_hip()
    if have is None:
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
    ap.add_argument("--steps", type=int, default=100)      # (0.27 s of timed steps at configs[1]; 20 were 55 ms)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="BASELINE.json config: c2 = configs[1] (default: VGG16, 500x500, batch 8 per GPU); c4 = configs[3] "
                         "(VGG16, 32 images per GPU = batch 256 on 8 GPUs); c5 = configs[4] (MobileNetV2, 1024x1024, 15 "
                         "anchors per cell, 1 image per GPU = batch 8 on 8 GPUs)")
    ap.add_argument("--backbone", default=None, choices=["vgg16", "mobilenet_v2"])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f32w", "bf16x3", "f16x3"],
                    help="conv arithmetic: exact f32 MFMA, or f32 operands carried as hi+lo bf16/f16 halves with 3 MFMAs per "
                         "product and f32 accumulation (DESIGN.md 4.1: f16x3 measures as accurate as exact f32)")
    ap.add_argument("--iou-threshold", type=float, default=0.7)
    ap.add_argument("--serial-nms", action="store_true",
                    help="run decode+NMS on the conv stream (default: NMS of step k overlaps the convs of step k+1 on a "
                         "second HIP stream; every step's work still completes inside the timed region)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise a world-size-1 RCCL group and run the N > 1 code path (record packing + all-gather)")
    ap.add_argument("--no-dist-legs", action="store_true",
                    help="N > 1 (or --force-dist) line: skip the `multi_gpu_configs` legs (configs[3] / configs[4] sharded + all-gathered)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the configs[2] box-path leg (`c3`) and the exact-float32 leg (`exact_f32`) of the N = 1 line")
    ap.add_argument("--sustained-seconds", type=float, default=1.0,
                    help="also loop the headline step for at least this long (wall clock) and report it as `sustained`; 0: off")
    ap.add_argument("--cpu-seconds", type=float, default=18.0, help="time budget of the whole CPU-baseline leg")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    return ap.parse_args()


def _cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_CPU_CHILD = """
import sys, time, pickle, json
import numpy as np, torch
sys.path.insert(0, %r)
from oracle import bbox_oracle as bo
from oracle import c_oracle as co
from oracle import conv_oracle as cv
backbone, n_threads, size, thr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
blob = pickle.load(open(sys.argv[5], "rb"))
weights, hp = blob["weights"], blob["hp"]
torch.set_num_threads(n_threads)
anchors = bo.generate_anchors(hp)
var = np.float32(hp["variances"])
rng = np.random.RandomState(0)


def whole(nb):
    img = rng.uniform(0, 1, size=(nb, size, size, 3)).astype(np.float32)
    t0 = time.perf_counter()
    r, c = cv.rpn_forward(backbone, img, weights)
    t1 = time.perf_counter()
    bx = co.decode(anchors, r.reshape(nb, -1, 4), var)
    co.combined_nms(bx[:, :, None, :], c.reshape(nb, -1, 1), 300, 300, iou_threshold=thr)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0


print(json.dumps({"ready": True}), flush=True)
for line in sys.stdin:          # commands from bench.py: "sweep <budget s>" | "loop <nb> <budget s>" | "quit"
    cmd = line.split()
    if not cmd or cmd[0] == "quit":
        break
    if cmd[0] == "sweep":       # batch 1 (cold, then best of 2 warm), batch 8 (first call, then a second one if it is cheap)
        budget = float(cmd[1])
        c0, w0 = whole(1)
        print(json.dumps({"batch": 1, "warm": False, "conv_s": c0, "whole_s": w0}), flush=True)
        best = min((whole(1) for _ in range(2)), key=lambda t: t[1])
        print(json.dumps({"batch": 1, "warm": True, "conv_s": best[0], "whole_s": best[1]}), flush=True)
        c8, w8 = whole(8)
        print(json.dumps({"batch": 8, "warm": False, "conv_s": c8, "whole_s": w8}), flush=True)
        if w8 < 0.25 * budget:
            c8, w8 = whole(8)
            print(json.dumps({"batch": 8, "warm": True, "conv_s": c8, "whole_s": w8}), flush=True)
    else:                       # the bounded sample of the headline
        nb, budget = int(cmd[1]), float(cmd[2])
        whole(nb)
        t0 = time.perf_counter(); n = 0
        while True:
            whole(nb); n += nb
            dt = time.perf_counter() - t0
            if dt >= budget or n >= 64:
                break
        print(json.dumps({"loop_images": n, "loop_s": dt, "batch": nb}), flush=True)
    print(json.dumps({"done": True}), flush=True)
"""


class _CpuChild(object):
    """One CPU-baseline worker process (never touches the GPU): started early so that its interpreter + torch start-up
    overlaps the parent's own measurements, then driven by one-line commands; every reply is read under a deadline and
    the process is killed by PID when it misses one."""

    def __init__(self, backbone, n_threads, size, thr, blob_path, bind):
        env = dict(os.environ)
        if bind:
            env.update(OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_NUM_THREADS=str(n_threads))
        env.setdefault("OMP_WAIT_POLICY", "passive")        # an idle worker's threads sleep instead of spinning beside the next one
        self.threads = n_threads
        self.p = subprocess.Popen([sys.executable, "-c", _CPU_CHILD % ROOT, backbone, str(n_threads), str(size),
                                   repr(float(thr)), blob_path], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, env=env)
        self.ready = False
        self._buf = b""

    def _read_until(self, key, deadline):
        """Reply rows up to the row that carries `key` -> (rows, True); (rows so far, False) at the deadline or at EOF.
        Reads the pipe's file descriptor itself (select + os.read + own line buffer: a buffered reader would hide lines
        from select)."""
        import select
        rows = []
        fd = self.p.stdout.fileno()
        while True:
            while b"\n" in self._buf:
                line, self._buf = self._buf.split(b"\n", 1)
                try:
                    row = json.loads(line.decode("utf-8", "replace"))
                except ValueError:
                    continue
                if key in row:
                    return rows, True
                rows.append(row)
            left = deadline - time.time()
            if left <= 0:
                return rows, False
            if not select.select([fd], [], [], min(left, 0.25))[0]:
                continue
            chunk = os.read(fd, 65536)
            if not chunk:
                return rows, False
            self._buf += chunk

    def wait_ready(self, startup=30.0):
        if not self.ready and self.p.poll() is None:
            _rows, ok = self._read_until("ready", time.time() + startup)
            if ok:
                self.ready = True
            else:
                self.close(kill=True)
        return self.ready

    def command(self, text, limit, startup=20.0):
        """Send one command; -> (reply rows, finished in time).  A worker that misses its deadline is killed."""
        if self.p.poll() is not None:
            return [], False
        if not self.ready:
            _rows, ok = self._read_until("ready", time.time() + startup)
            if not ok:
                self.close(kill=True)
                return [], False
            self.ready = True
        try:
            self.p.stdin.write((text + "\n").encode())
            self.p.stdin.flush()
        except (OSError, ValueError):
            return [], False
        rows, ok = self._read_until("done", time.time() + limit)
        if not ok:
            self.close(kill=True)
        return rows, ok

    def close(self, kill=False):
        if self.p.poll() is None:
            if kill:
                self.p.kill()
            else:
                try:
                    self.p.stdin.write(b"quit\n")
                    self.p.stdin.flush()
                    self.p.stdin.close()
                except (OSError, ValueError):
                    pass
            try:
                self.p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                self.p.kill()
                self.p.wait()


def _cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota); None = unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_workers_start(backbone, hyper_params, weights, iou_threshold, child_threads=(64, 128)):
    """Start the CPU-baseline worker processes (64 / 128 threads with OMP_PROC_BIND=close, and n = 1): called before the
    GPU legs of the line, so that their interpreter + torch start-up (seconds, one core each) is over, and they sleep on
    their command pipe, by the time `cpu_baseline` times anything.  -> state for cpu_baseline(workers=...)."""
    import pickle
    import tempfile
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    blob = tempfile.NamedTemporaryFile(suffix=".pkl")
    pickle.dump({"weights": weights, "hp": dict(hyper_params)}, blob)
    blob.flush()
    size = hyper_params["img_size"]
    procs = {n: _CpuChild(backbone, n, size, iou_threshold, blob.name, bind=True) for n in child_threads if n <= usable}
    procs[1] = _CpuChild(backbone, 1, size, iou_threshold, blob.name, bind=False)
    return {"blob": blob, "procs": procs, "child_threads": tuple(child_threads)}


def cpu_baseline(backbone, hyper_params, weights, iou_threshold, target_seconds, child_threads=(64, 128), workers=None):
    """CPU oracle ("port": the TF2 reference cannot run here) on this box's host cores, as BASELINE.md section 3 lays it
    out.  The whole path = conv stack on torch-CPU float32 (oneDNN) + the plain-C decode / NMS(300) restatement.  Thread
    sweep: 16 and 32 threads in this process, 64 and 128 threads (and n = 1) each in a child process with
    OMP_PROC_BIND=close under a timeout (oneDNN collapses when a box is oversubscribed: 256 threads on the 2 x 64-core
    host did not finish one image in 8 s in round 3, so that point is not re-run); batch 1 and batch 8 at every count.
    The headline `value` is the best whole-path figure found, re-timed on a bounded sample at that thread count and
    batch shape; `cores` = the threads that figure used.  The leg is budgeted at ~`target_seconds`."""
    import numpy as np
    import torch

    from oracle import bbox_oracle as bo
    from oracle import c_oracle as co
    from oracle import conv_oracle as cv
    rng = np.random.RandomState(0)
    anchors = bo.generate_anchors(hyper_params)
    var = np.float32(hyper_params["variances"])
    size = hyper_params["img_size"]
    A = len(anchors)
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    variants = []
    cands = []                                      # (images/s whole path, threads, batch, in_process)
    if workers is None:
        workers = cpu_workers_start(backbone, hyper_params, weights, iou_threshold, child_threads)
    child_threads = workers["child_threads"]
    blob, workers = workers["blob"], workers["procs"]
    for w in workers.values():                      # start-up must be over before anything is timed (no-op when pre-started)
        w.wait_ready()
    t_leg = time.perf_counter()

    def whole(nb):
        img = rng.uniform(0, 1, size=(nb, size, size, 3)).astype(np.float32)
        t0 = time.perf_counter()
        r, c = cv.rpn_forward(backbone, img, weights)
        t1 = time.perf_counter()
        bx = co.decode(anchors, r.reshape(nb, -1, 4), var)
        co.combined_nms(bx[:, :, None, :], c.reshape(nb, -1, 1), 300, 300, iou_threshold=iou_threshold)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t0, (r, c)

    def note(threads, batch, conv_s, whole_s, where, warm=True):
        variants.append({"leg": "whole path (torch-CPU f32 conv stack + plain-C decode/NMS)", "where": where,
                         "threads": threads, "batch": batch, "warm": warm, "conv_seconds": round(conv_s, 4),
                         "seconds": round(whole_s, 4), "images_per_s": round(batch / whole_s, 3)})
        if warm or batch == 8:
            cands.append((batch / whole_s, threads, batch, where == "in process"))

    # (a) in this process: 16 and 32 threads, batch 1 (warm) and batch 8
    heads8 = None
    for n_threads in sorted({min(usable, n) for n in (16, 32)}):
        torch.set_num_threads(n_threads)
        whole(1)                                    # warm-up (oneDNN primitive caches)
        c1, w1, _ = whole(1)
        note(n_threads, 1, c1, w1, "in process")
        if time.perf_counter() - t_leg < 0.45 * target_seconds:
            c8, w8, heads8 = whole(8)
            note(n_threads, 8, c8, w8, "in process", warm=False)

    # (b) worker processes (started at the top of this leg): 64 and 128 threads with OMP_PROC_BIND=close, and n = 1
    per_child = max(4.0, 0.25 * target_seconds)
    for n_threads in child_threads:
        w = workers.get(n_threads)
        if w is None:
            variants.append({"leg": "whole path, child process", "threads": n_threads, "note": "more threads than usable CPUs: skipped"})
            continue
        rows, done = w.command("sweep %r" % per_child, per_child)
        for r in rows:
            note(n_threads, r["batch"], r["conv_s"], r["whole_s"], "child process, OMP_PROC_BIND=close", warm=r["warm"])
        if not done:
            variants.append({"leg": "whole path, child process", "threads": n_threads,
                             "note": "sweep cut off at its %.0f s limit after %d measurement(s)" % (per_child, len(rows))})
    rows, done = workers[1].command("loop 1 1.0", 6.0)
    for r in rows:
        variants.append({"leg": "whole path, child process, n = 1 thread", "threads": 1, "batch": 1,
                         "seconds": round(r["loop_s"] / r["loop_images"], 4),
                         "images_per_s": round(r["loop_images"] / r["loop_s"], 3)})
    if not rows:
        variants.append({"leg": "whole path, child process, n = 1 thread", "threads": 1, "note": "did not finish in 6 s"})
    variants.append({"leg": "whole path, child process", "threads": 256,
                     "note": "not re-run: did not finish one image in 8 s in round 3 (BENCH_r03.json)"})

    # (c) box path alone at batch 8 on the conv stack's own outputs: numpy "eager-like" restatement, then the plain-C one
    if heads8 is not None:
        reg, cls = heads8[0].reshape(8, -1, 4), heads8[1].reshape(8, -1)
        t0 = time.perf_counter()
        boxes_np = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(reg[:1], var))
        bo.combined_non_max_suppression(boxes_np[:, :, None, :], cls[:1, :, None], 300, 300, iou_threshold=iou_threshold)
        dt_np = time.perf_counter() - t0
        variants.append({"leg": "decode + NMS(300), numpy restatement (eager-like)", "threads": 1, "batch": 1,
                         "seconds": round(dt_np, 4), "images_per_s": round(1.0 / dt_np, 3), "boxes_per_s": round(A / dt_np, 1)})
        t0 = time.perf_counter()
        boxes_c = co.decode(anchors, reg, var)
        co.combined_nms(boxes_c[:, :, None, :], cls[:, :, None], 300, 300, iou_threshold=iou_threshold)
        dt_c = time.perf_counter() - t0
        variants.append({"leg": "decode + NMS(300), plain-C restatement", "threads": 1, "batch": 8, "seconds": round(dt_c, 4),
                         "images_per_s": round(8.0 / dt_c, 3), "boxes_per_s": round(8.0 * A / dt_c, 1)})

    # (d) headline: the best (threads, batch shape) found, re-timed on a bounded sample for what is left of the budget
    rate, cores, nb, in_proc = max(cands)
    left = max(2.0, target_seconds - (time.perf_counter() - t_leg))
    n = dt = None
    if not in_proc:
        rows, _done = workers[cores].command("loop %d %r" % (nb, left), left + 8.0)
        if rows:
            n, dt = rows[-1]["loop_images"], rows[-1]["loop_s"]
            where = "child process, OMP_PROC_BIND=close"
        else:                                       # the worker did not come back: fall back to the best in-process point
            rate, cores, nb, in_proc = max(c for c in cands if c[3])
    if n is None:
        where = "in process"
        torch.set_num_threads(cores)
        whole(nb)
        t0 = time.perf_counter()
        n = 0
        while True:
            whole(nb)
            n += nb
            dt = time.perf_counter() - t0
            if dt >= left or n >= 64:
                break
    for w in workers.values():
        w.close()
    blob.close()
    # `cores`: the CPUs this process can actually use = min(threads, cgroup quota) -- the GPU boxes grant 16 CPUs of time while 256
    # logical CPUs are visible, and the fastest configuration found runs MORE threads than that (`threads`)
    quota = _cpu_quota()
    granted = cores if not quota else max(1, min(cores, int(round(quota))))
    return {"value": n / dt, "unit": "images/s", "cores": granted, "threads": cores, "kind": "port",
            "sample": "%d images of the same workload, %d per call (torch-CPU f32 conv stack with %d threads on %d granted CPUs + "
                      "plain-C decode/NMS(300), single thread; %s), %.1f s" % (n, nb, cores, granted, where, dt),
            "cpu": _cpu_model_name(), "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
            "variants": variants, "leg_seconds": round(time.perf_counter() - t_leg, 1)}


def measured_traffic(kernel, precision, backbone, img_size, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, written by scripts/collect_profiles.py under the same kernel names as this file's
    per-op table: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs of this same command) -- PMC
    counters cannot be collected from inside an un-profiled run, so the figure is read back from the newest committed
    profile and labelled with its file name (`traffic_source`).  A profile only counts when its recorded `workload`
    (backbone, img_size, per-GPU batch, precision) is THIS run's: a figure measured on another shape is not this
    kernel's traffic.  (None, None) when no profile matches."""
    import glob
    want = {"backbone": backbone, "img_size": int(img_size), "batch": int(batch), "precision": precision}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            if prof.get("workload") != want:
                continue
            return round(prof["kernels"][kernel]["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def c3_traffic():
    """HBM bytes per launch of the configs[2] box kernels from the newest committed counter passes
    (profiles/*_c3_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of scripts/bench_bbox.py, B = 64)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)["kernels"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return {}, None


def c3_kernel_stat(pattern):
    """Average duration (us) of a configs[2] box kernel from the newest committed rocprofv3 --kernel-trace --stats summary
    (profiles/*_c3_kernel_stats.csv, scripts/bench_bbox.py at B = 64) -> (us, file) or (None, None)."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_kernel_stats.csv")), reverse=True):
        try:
            with open(path) as f:
                for row in csv.DictReader(f):
                    if pattern in row["Kernel"]:
                        return float(row["AverageNs"]) * 1e-3, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def winograd_reduction(op):
    """Direct-conv flops / flops the kernel executes: 1 for direct kernels; 2.25 for Winograd F(2x2,3x3) (16 instead of 36
    multiplications per 2 x 2 outputs and channel pair), 4 for F(4x4,3x3) (36 instead of 144 per 4 x 4 outputs)."""
    if op["arith"] != "f32w":
        return 1.0
    return 4.0 if op["kernel"].startswith("conv3x3_wino4") else 2.25


def exact_f32_leg(backbone, hp, weights, B, iou_threshold, imgs, steps=30, warmup=2, heads=None, precision="f32", keep_heads=None):
    """The parity-clean arithmetic beside the headline (report both): the same workload with every conv
    on the exact float32 MFMA (v_mfma_f32_32x32x2_f32: float32 products, float32 accumulation in a fixed order per
    output -- within a channel quad the products are taken in the order 0, 2, 1, 3 by conv_igemm_f32 / conv_igemm_f32_dma
    and 0, 1, 2, 3 by the generic layers; deterministic and batch-invariant, but not the bits of a sequential-K fmaf chain),
    timed over `steps` steps in this same run, with the dominant kernel's fraction of the 157.3 TFLOP/s f32-MFMA peak from
    a per-op event pass.  `heads` = the headline path's (reg, cls) head outputs for the same images: their largest
    absolute difference from this path's is returned as `max_abs_diff_vs_headline` (the bench line's self-check).
    precision="f32w" (the `f32_winograd` leg): the 3x3 convs as float32 Winograd F(4x4, 3x3) / F(2x2, 3x3) on the same float32 MFMA (float32
    operands and accumulation, 2.25 x -- F(4x4,3x3): 4 x -- fewer multiply-adds, another summation order: a precision of its own, never
    the parity-clean `exact_f32` row).  Its `roofline.achieved` prices the multiply-adds the kernel EXECUTES (`winograd_reduction`)
    against the 157.3 TFLOP/s peak; `effective_tflops` is the direct conv's flops over the same time.  `keep_heads`: a list that
    receives this path's (reg, cls) head outputs; `heads` may then be the exact-f32 leg's."""
    import torch

    from tf_rpn_amd.predictor import Proposer
    prop = Proposer(backbone, hyper_params=hp, weights=weights, precision=precision, max_batch=B,
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
        d = tot.setdefault(op["kernel"], [0.0, 0.0, 0, 0.0])
        d[0] += t
        d[1] += op["flops_per_image"] * B / winograd_reduction(op)                         # multiply-adds executed
        d[2] += op["launches"]
        d[3] += op["flops_per_image"] * B                                                  # the direct conv's
    dom = max(tot, key=lambda k: tot[k][0])
    achieved = tot[dom][1] / (tot[dom][0] * 1e-3) / 1e12
    diff = None
    if heads is not None or keep_heads is not None:
        d32, s32 = prop.forward(imgs)
        torch.cuda.synchronize()
        if heads is not None:
            diff = {"reg": float((d32.reshape(-1) - heads[0].reshape(-1)).abs().max().item()),
                    "cls": float((s32.reshape(-1) - heads[1].reshape(-1)).abs().max().item())}
        if keep_heads is not None:
            keep_heads.extend([d32.clone(), s32.clone()])
    del prop
    torch.cuda.empty_cache()
    out = {"value": round(B * steps / dt, 2), "unit": "images/s", "ms_per_step": round(1e3 * dt / steps, 4),
           "steps": steps, "dtype": precision, "max_abs_diff_vs_headline": diff,
           "roofline": {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_TFLOPS["f32"],
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS["f32"], 4),
                        "launches_per_step": tot[dom][2], "avg_launch_ms": round(tot[dom][0] / tot[dom][2], 4)}}
    if precision == "f32w":
        out["max_abs_diff_vs_exact_f32"] = out.pop("max_abs_diff_vs_headline")
        out["roofline"]["effective_tflops"] = round(tot[dom][3] / (tot[dom][0] * 1e-3) / 1e12, 3)
        out["note"] = ("3x3 convs as float32 Winograd on the f32 MFMA -- F(4x4,3x3) (1/4 of the direct conv's multiply-adds; per layer "
                       "on 16x16-pixel x 128-channel or 16x32 x 64 tiles, the 31x31 layers with their input channels split over two "
                       "workgroups), F(2x2,3x3) (1/2.25) on small grids: float32 operands and accumulation, another summation "
                       "order -- not the parity-clean `exact_f32` row")
    return out


def pool_leg(backbone, hp, weights, B, precision, iou_threshold, imgs, steps, warmup, in_flight=2):
    """The same `steps` steps over `in_flight` independent pipelines fed round robin (predictor.ProposerPool: each its own model
    handle and streams): the second batch's kernels fill the CUs the first leaves idle (the small kernels of a MobileNetV2 step;
    VGG16 at batch 8, whose persistent layers own whole CUs, does not gain: 2 888 vs 2 928 images/s, scripts/depth_probe.py vgg8).  Reported BESIDE the one-pipeline figure, which keeps its meaning from earlier
    rounds -- and its per-kernel roofline: under two pipelines a kernel's start-to-end time includes waiting for the other
    pipeline's workgroups to leave the CUs."""
    import torch

    from tf_rpn_amd.predictor import ProposerPool
    pool = ProposerPool(in_flight, backbone, hyper_params=hp, weights=weights, precision=precision, max_batch=B,
                        iou_threshold=iou_threshold)
    for _ in range(warmup):
        pool.propose_async(imgs)
    pool.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pool.propose_async(imgs)
    pool.wait()
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    res = {"value": round(B * steps / dt2, 2), "unit": "images/s", "ms_per_step": round(1e3 * dt2 / steps, 4), "steps": steps,
           "checks": {"f16_range": any(bool(q.rpn_model.status(reset=False)["f16_range"]) for q in pool.pipelines)
                      if precision == "f16x3" else False,
                      "valid_min": min(int(q._last["valid"][:B].min().item()) for q in pool.pipelines)}}
    del pool
    torch.cuda.empty_cache()
    return res


def config_leg(label, backbone, hp, B, precision, iou_threshold, steps=10, warmup=3, in_flight=0):
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
    for _ in range(min(steps, 50)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        prop.propose(imgs)
        e1.record()
        torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat.sort()
    n_launches = sum(op["launches"] for op in prop.rpn_model.ops())
    leg_checks = {"f16_range": bool(prop.rpn_model.status(reset=False)["f16_range"]) if precision == "f16x3" else False,
                  "valid_min": int(prop._valid[:B].min().item())}
    del prop
    torch.cuda.empty_cache()
    out = {"workload": "%s: %s, %dx%d, %d anchors/cell, batch %d per GPU" % (label, backbone, hp["img_size"], hp["img_size"],
                                                                            hp["anchor_count"], B),
           "value": round(B * steps / dt, 2), "unit": "images/s", "ms_per_step": round(1e3 * dt / steps, 4),
           "ms_per_image": round(1e3 * dt / steps / B, 4), "latency_ms_one_step_unpipelined": round(lat[len(lat) // 2], 4),
           "steps": steps, "dtype": precision, "conv_launches_per_step": n_launches, "checks": leg_checks}
    if in_flight > 1:
        out["pipelines_in_flight_%d" % in_flight] = pool_leg(backbone, hp, weights, B, precision, iou_threshold, imgs, steps,
                                                             max(warmup, 2 * in_flight), in_flight)
    del imgs
    return out


def dist_config_leg(label, backbone, hp, B, precision, iou_threshold, world, rank, steps=30, warmup=3):
    """One of BASELINE.json's 8-GPU configs (configs[3]: VGG16, 32 images per GPU; configs[4]: MobileNetV2 1024x1024, 15
    anchors per cell, ONE image per GPU) through the sharded path the N > 1 line is about: every rank proposes for its own
    images (`Proposer.propose_distributed_pipelined`: convs on the current stream; decode + NMS, record packing and the ONE
    RCCL all-gather of 6 KB records on the side stream, beside the convs of the next step), `steps` timed steps bracketed by
    barrier + synchronize on both sides, MAX over ranks.  Called by EVERY rank (collectives inside); rank 0 gets the dict.
    With `--force-dist` on one GPU the group has one rank: the same code path, the all-gather included.
    (The reference has no counterpart: one process, /root/reference/utils/io_utils.py:52-59.)"""
    import torch
    import torch.distributed as dist

    from tf_rpn_amd.models._rpn_model import synthetic_weights
    from tf_rpn_amd.predictor import Proposer
    weights = synthetic_weights(backbone, hp, seed=1)
    prop = Proposer(backbone, hyper_params=hp, weights=weights, precision=precision, max_batch=B,
                    iou_threshold=iou_threshold, overlap_nms=True)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1000 + rank)
    imgs = torch.rand((B, hp["img_size"], hp["img_size"], 3), generator=gen, device="cuda", dtype=torch.float32)
    M = prop.topn
    gather_bufs = [torch.empty((world * B, M * 5 + 1), dtype=torch.float32, device="cuda") for _ in range(2)]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 2)):
        prop.propose_distributed_pipelined(imgs, gather_bufs)
    prop.flush_distributed(gather_bufs)
    torch.cuda.synchronize()
    side_ok = prop.ensure_side_stream()           # the communicator's streams exist: still beside the conv stream?
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        prop.propose_distributed_pipelined(imgs, gather_bufs)
    gathered = prop.flush_distributed(gather_bufs)
    fence()
    elapsed = time.perf_counter() - t0
    # the gathered records of the last step: every rank's valid counts, boxes and scores, as every rank received them
    g_boxes, g_scores, g_valid = Proposer.unpack_records(gathered[:world * B], M)
    valid_min = int(g_valid.min().item())
    finite = bool(torch.isfinite(g_boxes).all().item() and torch.isfinite(g_scores).all().item())
    # rank r's rows must be rank r's own proposals: compare this rank's slice with its local outputs
    own = gathered[rank * B:(rank + 1) * B]
    mine = prop.pack_records(prop._last["boxes"][:B], prop._last["scores"][:B], prop._last["valid"][:B])
    own_rows_match = bool(torch.equal(own.view(torch.int32), mine.view(torch.int32)))
    f16_range = bool(prop.rpn_model.status(reset=False)["f16_range"]) if precision == "f16x3" else False
    # the all-gather alone, after the region
    rec = prop.pack_records(prop._last["boxes"][:B], prop._last["scores"][:B], prop._last["valid"][:B])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dist.all_gather_into_tensor(gather_bufs[0], rec)
    e1.record()
    torch.cuda.synchronize()
    gather_ms = e0.elapsed_time(e1) / 10
    mine_t = torch.tensor([elapsed, gather_ms, 1.0 if (f16_range or valid_min < 1 or not finite or not own_rows_match) else 0.0],
                          dtype=torch.float64, device="cuda")
    allr = [torch.zeros_like(mine_t) for _ in range(world)]
    dist.all_gather(allr, mine_t)
    t_max = max(float(v[0].item()) for v in allr)
    n_launches = sum(op["launches"] for op in prop.rpn_model.ops())
    del prop, gather_bufs, imgs
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {"workload": "%s: %s, %dx%d, %d anchors/cell, batch %d per GPU x %d GPU(s), image-sharded + one RCCL all-gather of "
                        "proposal records" % (label, backbone, hp["img_size"], hp["img_size"], hp["anchor_count"], B, world),
            "value": round(world * B * steps / t_max, 2), "unit": "images/s", "images_per_s": round(world * B * steps / t_max, 2),
            "ms_per_step": round(1e3 * t_max / steps, 4), "steps": steps, "n_gpus": world, "global_batch": world * B,
            "dtype": precision, "allgather_ms": round(gather_ms, 4), "record_bytes_per_rank": B * (M * 5 + 1) * 4,
            "rccl_ranks": dist.get_world_size(), "nms_stream_concurrent": side_ok, "conv_launches_per_step": n_launches,
            "per_rank": {"images_per_s": [round(B * steps / float(v[0].item()), 2) for v in allr],
                         "allgather_ms": [round(float(v[1].item()), 4) for v in allr]},
            "checks": {"f16_range": f16_range, "valid_min": valid_min, "proposals_finite": finite,
                       "own_rows_match": own_rows_match, "any_rank_failed": any(float(v[2].item()) > 0 for v in allr)}}


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
    # (`us` is launch-to-launch time over back-to-back launches on ONE stream: for this 18 MB kernel that is the ~2-3 us boundary
    # between dependent launches + the kernel; `kernel_us` below is the kernel alone, from the committed rocprofv3 summary)
    out["decode"] = {"us": round(t * 1e6, 2), "GBps": round(by / t / 1e9, 1), "frac": round(by / t / 8e12, 4), "bytes": by,
                     "timing": "launch-bound: events around n back-to-back launches on one stream (kernel + launch boundary)"}
    # the kernel's own duration: rocprofv3 --kernel-trace --stats of the same launches (the newest committed summary; spreading the
    # launches over several streams from Python measured the HOST, 56 us per launch)
    kstat, ksrc = c3_kernel_stat("decode_kernel")
    if kstat is not None:
        out["decode"].update({"kernel_us": round(kstat, 2), "kernel_GBps": round(by / (kstat * 1e-6) / 1e9, 1),
                              "kernel_frac": round(by / (kstat * 1e-6) / 8e12, 4), "kernel_timing": "rocprofv3 average, " + ksrc})
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
                                           "GBps": round(by / t / 1e9, 2), "frac": round(by / t / 8e12, 4), "bytes": by,
                                           "mean_valid": round(float(ov.float().mean().item()), 1)}
    # measured HBM traffic per launch (committed counter passes of the same kernels at this shape, labelled with the file)
    kernels, src = c3_traffic()
    for key, pat in (("decode", "decode_kernel"), ("iou_map", "iou_map"), ("decode_nms_iou0.5", "nms_kernel<true>"),
                     ("decode_nms_iou0.7", "nms_kernel<true>")):
        hit = [v for k, v in kernels.items() if pat in k]
        out[key]["traffic"] = hit[0]["hbm_bytes_per_launch"] if hit else None
        out[key]["traffic_source"] = src if hit else None
    return out


def _device_info():
    """Which device of the pool ran this line (rank 0's): the f16x3 path is power / clock bound and the pool's devices differ by up
    to ~8 % on it (NOTES.md round 6), the float32 paths by < 1 %."""
    try:
        import torch
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        return {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
                "max_clock_mhz": int(getattr(p, "clock_rate", 0)) // 1000 or None,
                "note": "devices of the pool measured 2 930-3 170 images/s on this workload with identical code (round 6)"}
    except Exception as exc:                       # never fail the bench line over a property query
        return {"error": str(exc)}


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

    model = prop.rpn_model
    ops = model.ops()

    def dominant_kernel(ms_per_op):               # the conv kernel template with the largest total time
        tot = {}
        for op, ms in zip(ops, ms_per_op):
            tot[op["kernel"]] = tot.get(op["kernel"], 0.0) + ms
        return max(tot, key=tot.get)

    # untimed pre-pass, BEFORE the warm-up: every op timed, to find the dominant kernel.  The timed region then carries HIP
    # events only around ITS launches (2 events per launch on the launch stream; events around all 20 ops cost ~2.5 % of a
    # step).  (Until round 3 this pass -- with its host-side read-back -- sat between the warm-up and the timed region: the
    # device idled for milliseconds right in front of the K timed steps.)
    model.set_profiling(2)
    for _ in range(2):
        step()
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)
    torch.cuda.synchronize()
    # the communicator's streams exist now (the steps above ran its first all-gathers): make sure the NMS side stream still runs
    # BESIDE the conv stream (HIP assigns its four hardware queues in creation order; a shared queue serialises the two silently)
    side_stream_ok = prop.ensure_side_stream() if overlap else None
    dom = dominant_kernel(model.profile_ms()[0])
    dom_mask = [op["kernel"] == dom for op in ops]
    model.set_profiling_mask(dom_mask)
    # one of its launches per step, round robin (2 events per step), when K gives every launch >= 2 samples;
    # otherwise all of them in every step
    rotate = args.steps >= 2 * sum(dom_mask)
    model.set_profiling_rotate(rotate)
    model.set_profiling(args.steps)               # (a ring of the last K forwards: the warm-up's fall out of it)
    # ---- W untimed warm-up steps, then the timed region: exactly K steps, barrier + synchronize on both sides ----------
    for _ in range(args.warmup):
        step()
    if overlap and use_dist:
        prop.flush_distributed(gather_bufs)
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
        prop.decode_nms(deltas, scores, B, *prop_out)
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
    # the conv stack alone (images resident -> reg / cls written), per-op events OFF: two events around a whole forward
    stack = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prop.forward(imgs)
        e1.record()
        torch.cuda.synchronize()
        stack.append(e0.elapsed_time(e1))
    conv_stack_ms = sorted(stack)[len(stack) // 2]
    if overlap:                                   # time decode+NMS alone, after the region, for the report
        torch.cuda.synchronize()
        d_, s_ = prop._bufs[0]["reg"][:B].view(B, -1, 4), prop._bufs[0]["cls"][:B].view(B, -1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            prop.decode_nms(d_, s_, B, prop._boxes[:B], prop._scores[:B], prop._idx[:B], prop._valid[:B])
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
    per_rank = None
    if world > 1:
        # every rank's own clock over the K steps and its own all-gather time, so that a scaling line explains itself
        mine = torch.tensor([elapsed, gather_ms if gather_ms is not None else 0.0], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"images_per_s": [round(B * args.steps / float(v[0].item()), 2) for v in allr],
                    "allgather_ms": [round(float(v[1].item()), 4) for v in allr]}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    # ---- sustained leg: the same step looped for >= 1 s of wall clock (the driver's K = 20 is ~56 ms: too short for the
    # clocks to settle or for a utilisation sampler to see the GPU).  Same fences as the timed region; every rank runs the
    # same count (derived from the max-over-ranks time above).  `value` stays the K-step figure the contract asks for.
    n_sus = max(args.steps, int(1.15 * args.sustained_seconds / (ms_per_step * 1e-3)) + 1) if args.sustained_seconds > 0 else 0
    sustained = None
    if n_sus:
        fence()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        if overlap and use_dist:
            prop.flush_distributed(gather_bufs)
        elif overlap:
            prop.wait()
        fence()
        ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        sustained = {"steps": n_sus, "seconds": round(float(ts.item()), 4),
                     "images_per_s": round(world * B * n_sus / float(ts.item()), 2),
                     "ms_per_step": round(1e3 * float(ts.item()) / n_sus, 4)}

    # ---- self-check of what was just timed (predictor.py:50-56: the loop's outputs must be usable proposals) --------
    # the float16 range word of every forward so far (sticky, raised on the device), the proposals of the last step, and
    # (N = 1, below) the head outputs against the exact-float32 path on the same images
    last = prop._last if overlap else {"valid": prop._valid, "scores": prop._scores, "boxes": prop._boxes}
    torch.cuda.synchronize()
    checks = {"f16_range": bool(model.status(reset=False)["f16_range"]) if args.precision == "f16x3" else False,
              "valid_min": int(last["valid"][:B].min().item()),
              "proposals_finite": bool(torch.isfinite(last["boxes"][:B]).all().item()
                                       and torch.isfinite(last["scores"][:B]).all().item()),
              "max_abs_diff_vs_exact_f32": None, "tolerance": 1e-4}
    if world > 1:
        bad = torch.tensor([1.0 if (checks["f16_range"] or checks["valid_min"] < 1 or not checks["proposals_finite"]) else 0.0],
                           device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        checks["any_rank_failed"] = bool(bad.item() > 0)

    # ---- BASELINE.json's own 8-GPU configs through the sharded + all-gathered path (every rank takes part) ----------------
    dist_legs = None
    if use_dist and not args.no_dist_legs and args.config == "c2" and args.backbone == "vgg16" and B == 8:
        hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                                  anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
        dist_legs = {
            "c4": dist_config_leg("configs[3] (batch %d on %d GPUs; 256 on 8)" % (32 * world, world), "vgg16", hp, 32, args.precision,
                                  args.iou_threshold, world, rank, steps=30, warmup=3),
            "c5": dist_config_leg("configs[4] (batch %d on %d GPUs; 8 on 8)" % (world, world), "mobilenet_v2", hp_c5, 1,
                                  args.precision, args.iou_threshold, world, rank, steps=400, warmup=10),
        }

    exit_code = 0
    if rank == 0:
        # dominant kernel: the conv instantiation with the largest total time
        by_kernel = {}
        for op, ms in zip(ops, last_ms):
            d = by_kernel.setdefault(op["kernel"], {"ms": 0.0, "flops": 0.0, "launches": 0})
            d["ms"] += ms
            d["flops"] += op["flops_per_image"] * B / winograd_reduction(op)   # (Winograd: the multiply-adds executed)
            d["launches"] += op["launches"]
        d = by_kernel[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # the peak of the arithmetic the DOMINANT KERNEL runs in: split-precision 3x3 kernels -> 16-bit MFMA / 3 products;
        # everything else (float32 implicit GEMM, the fused MobileNetV2 blocks) -> the float32 MFMA
        dom_arith = next(op["arith"] for op in ops if op["kernel"] == dom)
        peak = PEAK_TFLOPS[dom_arith]
        traffic, traffic_source = measured_traffic(dom, args.precision, args.backbone, hp["img_size"], B)
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "launches_per_step": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                    "flops_per_launch": d["flops"] / d["launches"],
                    "timing": ("HIP events on the launch stream around this kernel's launches in the K timed steps"
                               + (": one launch per step, round robin (every launch sampled >= 2 times, 12 at the default K)" if rotate else "")),
                    "arith": dom_arith,
                    "conv_stack_ms": round(conv_stack_ms, 4),
                    "conv_stack_ms_note": "HIP events around whole forwards, per-op events off (median of 5)",
                    "sum_of_op_ms_with_per_op_events": round(sum(last_ms), 3),
                    "conv_launches_per_step": sum(op["launches"] for op in ops),
                    "decode_nms_ms": round(nms_ms, 4)}
        out = {
            "metric": "proposal images/sec at 500x500x3 VOC batch; NMS boxes/sec",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "dtype_note": {"f32": "float32 in, float32 MFMA accumulate (exact)",
                           "f32w": "float32 in, float32 MFMA accumulate; the 3x3 convs as Winograd F(4x4,3x3) / F(2x2,3x3): 1/4 / 1/2.25 of the multiply-adds, another summation order",
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
            "device": _device_info(),
            "allgather_ms": None if gather_ms is None else round(gather_ms, 4),
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "nms_stream_concurrent": side_stream_ok,      # measured (predictor._streams_overlap), rank 0
            "per_rank": per_rank,
            "sustained": sustained,
            "sustained_images_per_s": None if sustained is None else sustained["images_per_s"],
            "checks": checks,
            "roofline": roofline,
        }
        if dist_legs is not None:
            out["multi_gpu_configs"] = dist_legs
        cpu_workers = None
        if world == 1 and not args.no_cpu_baseline:
            # (start-up of the CPU workers overlaps the GPU legs below; they never touch the GPU)
            cpu_workers = cpu_workers_start(args.backbone, hp, weights, args.iou_threshold)
        if world == 1 and not args.no_extra_legs:
            # the metric's second half ("NMS boxes/sec") is quoted on configs[2] (B = 64, box path only)
            out["c3"] = c3_leg(hp)
            out["nms_boxes_per_sec"] = out["c3"]["decode_nms_iou0.7"]["boxes_per_sec"]
            out["nms_boxes_per_sec_note"] = "configs[2] (B=64), fused decode+NMS(300), iou 0.7; `c3` has iou 0.5 and the GB/s"
            if args.precision != "f32":
                d_h, s_h = prop.forward(imgs)
                heads = (d_h.clone(), s_h.clone())
                exact_heads = []
                out["exact_f32"] = exact_f32_leg(args.backbone, hp, weights, B, args.iou_threshold, imgs, heads=heads, keep_heads=exact_heads)
                checks["max_abs_diff_vs_exact_f32"] = out["exact_f32"]["max_abs_diff_vs_headline"]
                if args.precision != "f32w":
                    # float32 Winograd beside it (a precision of its own; its check is against the exact-f32 head outputs)
                    out["f32_winograd"] = exact_f32_leg(args.backbone, hp, weights, B, args.iou_threshold, imgs, heads=tuple(exact_heads),
                                                        precision="f32w")
                    checks["f32_winograd_max_abs_diff_vs_exact_f32"] = out["f32_winograd"]["max_abs_diff_vs_exact_f32"]
                del exact_heads
                if args.precision == "f16x3":
                    checks["f16_range"] = checks["f16_range"] or bool(model.status(reset=False)["f16_range"])
            if args.config == "c2" and args.backbone == "vgg16" and B == 8:
                # the other BASELINE.json configs (per-GPU shapes), so that every config has a driver-run figure
                hp_mn = dict(train_utils.get_hyper_params("mobilenet_v2"))
                hp_c5 = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                                          anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.]))
                out["other_configs"] = {
                    # (one-image steps are 0.19 - 0.3 ms: 400 of them (0.08 - 0.12 s), batch 8 300 (0.13 s), so that the pipeline's fill
                    # and drain -- one exposed NMS of ~0.06 ms -- and the clocks' settling weigh nothing)
                    "c1": config_leg("configs[0]", "mobilenet_v2", hp_mn, 1, args.precision, args.iou_threshold, steps=400, warmup=10,
                                     in_flight=2),
                    "c4": config_leg("configs[3] (batch 256 on 8 GPUs)", "vgg16", hp, 32, args.precision, args.iou_threshold,
                                     steps=30, warmup=2),
                    "c5": config_leg("configs[4] (batch 8 on 8 GPUs)", "mobilenet_v2", hp_c5, 1, args.precision,
                                     args.iou_threshold, steps=400, warmup=10, in_flight=2),
                    "mobilenet_v2_b8": config_leg("configs[0] shape at batch 8", "mobilenet_v2", hp_mn, 8, args.precision,
                                                  args.iou_threshold, steps=300, warmup=10, in_flight=2),
                }
        else:
            out["nms_boxes_per_sec"] = round(B * prop.total_anchors / (nms_ms * 1e-3), 1)
            out["nms_boxes_per_sec_note"] = "this run's own decode+NMS at its per-GPU batch (%d images)" % B
        if args.layers:
            for op, ms in zip(ops, last_ms):
                if not op["launches"]:              # a pool that runs inside the previous conv's epilogue: no launch
                    continue
                tf = op["flops_per_image"] * B / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                gbs = op["bytes_per_image"] * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                print("%-28s %-34s %8.3f ms %8.2f TF/s %9.1f GB/s" % (op["name"], op["kernel"], ms, tf, gbs),
                      file=sys.stderr)
            print("%-28s %-34s %8.3f ms" % ("decode+nms", "nms_kernel<decode>", nms_ms), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.backbone, hp, weights, args.iou_threshold, args.cpu_seconds,
                                               workers=cpu_workers)
        else:
            out["cpu_baseline"] = None
        diff = checks["max_abs_diff_vs_exact_f32"]
        legs_bad = [k for k, v in out.get("other_configs", {}).items()
                    for c in [v["checks"]] + [w["checks"] for n, w in v.items() if n.startswith("pipelines_in_flight_")]
                    if c["f16_range"] or c["valid_min"] < 1]
        legs_bad += [k + "_dist" for k, v in (dist_legs or {}).items()
                     if v["checks"]["any_rank_failed"] or v["checks"]["valid_min"] < 1 or not v["checks"]["own_rows_match"]]
        legs_bad = sorted(set(legs_bad))
        checks["other_configs_failed"] = legs_bad
        checks["ok"] = not (bool(legs_bad) or checks["f16_range"] or checks["valid_min"] < 1 or not checks["proposals_finite"]
                            or checks.get("any_rank_failed", False)
                            or (diff is not None and max(diff.values()) > checks["tolerance"])
                            or (checks.get("f32_winograd_max_abs_diff_vs_exact_f32") is not None
                                and max(checks["f32_winograd_max_abs_diff_vs_exact_f32"].values()) > checks["tolerance"]))
        print(json.dumps(out), flush=True)
        if not checks["ok"]:
            sys.stderr.write("bench.py: self-check FAILED: %s\n" % json.dumps(checks))
            exit_code = 3
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
