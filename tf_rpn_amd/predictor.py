"""The proposal loop of the reference's ``predictor.py:41-60`` on MI355X.

Per batch (predictor.py:50-56): ``rpn_model.predict_on_batch(imgs)`` -> reshape the head
outputs to (B,A,4) / (B,A) -> ``deltas *= variances`` -> ``get_bboxes_from_deltas`` -> pick
proposals.  The reference picks ``tf.nn.top_k(rpn_labels, 10)`` for display
(predictor.py:58-60); the proposal path named by the north star substitutes
``non_max_suppression`` with ``test_nms_topn = 300`` (utils/train_utils.py:29,
utils/bbox_utils.py:48-70).  Both selectors are provided.

Data parallelism (not in the reference, which is single-process): images are independent, so
each rank takes a contiguous slice of the batch, runs the whole path locally, and one RCCL
all-gather of fixed-size proposal records collects the result.  There is no
other collective on the data path.
"""
import warnings

import torch

from . import _lib as L
from .utils import bbox_utils, train_utils


def _streams_overlap(a, b, microseconds=200):
    """True when work on HIP streams ``a`` and ``b`` runs CONCURRENTLY.  HIP maps streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, default 4, the default stream's included) in creation order -- measured (scripts/queue_probe.py): of ten
    streams {0, 7}, {1, 6}, {2, 5, 9} and {3, 4, 8} share one -- so whether two streams share a queue depends on what else the
    process has created; on a shared queue the NMS of batch k would run BEHIND the convs of batch k + 1 instead of beside them,
    silently.  One sleeping wave on each stream (rpn_stream_spin): beside each other they take ~1x, behind each other ~2x."""
    # device timestamps (events), not the host's clock: the host may be descheduled between the calls; and up to three rounds --
    # a shared queue NEVER shows the short time, so one short round settles it
    for _ in range(3):
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(a)
        L.check(L.lib().rpn_stream_spin(L.vp(a.cuda_stream), microseconds), "rpn_stream_spin")
        L.check(L.lib().rpn_stream_spin(L.vp(b.cuda_stream), microseconds), "rpn_stream_spin")
        ea.record(a)
        eb.record(b)
        torch.cuda.synchronize()
        if max(e0.elapsed_time(ea), e0.elapsed_time(eb)) < 1.5e-3 * microseconds:
            return True
    return False


def _concurrent_side_stream(must=None, should=(), tries=8):
    """A new stream that runs beside stream ``must`` (default: the current stream) and, if one of ``tries`` candidates does, beside
    every stream of ``should`` as well (torch hands out pooled streams round robin; each is tested with _streams_overlap).
    RuntimeWarning when no candidate runs beside ``must`` (e.g. GPU_MAX_HW_QUEUES=1): results stay correct, the overlap is lost."""
    must = must if must is not None else torch.cuda.current_stream()
    best, best_n = None, -1
    for _ in range(tries):
        s = torch.cuda.Stream()
        if not _streams_overlap(must, s):
            if best is None:
                best, best_n = s, -1
            continue
        n = sum(1 for o in should if _streams_overlap(o, s))
        if n == len(should):
            return s
        if n > best_n:
            best, best_n = s, n
    if best_n < 0:
        import warnings
        warnings.warn("tf_rpn_amd: no side stream runs concurrently with the conv stream (hardware queues shared: "
                      "GPU_MAX_HW_QUEUES?); the NMS / conv overlap is lost, results are unaffected", RuntimeWarning, stacklevel=3)
    return best


class Proposer(object):
    """anchors once (predictor.py:46), then ``propose(imgs)`` per batch."""

    def __init__(self, backbone="vgg16", hyper_params=None, weights="synthetic", precision="f32",
                 max_batch=8, iou_threshold=0.7, score_threshold=float("-inf"), seed=1, overlap_nms=False,
                 check_range=True, avoid_streams=()):
        if backbone == "mobilenet_v2":
            from .models.rpn_mobilenet_v2 import get_model
        else:
            from .models.rpn_vgg16 import get_model
        self.backbone = backbone
        self.hyper_params = hyper_params if hyper_params is not None else train_utils.get_hyper_params(backbone)
        hp = self.hyper_params
        self.rpn_model, self.feature_extractor = get_model(hp, weights=weights, precision=precision,
                                                           max_batch=max_batch, seed=seed)
        if self.rpn_model.feature_map_shape != int(hp["feature_map_shape"]):
            raise ValueError("hyper_params feature_map_shape=%s but the %s graph at img_size=%s yields %d"
                             % (hp["feature_map_shape"], backbone, hp["img_size"], self.rpn_model.feature_map_shape))
        self.anchors = bbox_utils.generate_anchors(hp)                       # predictor.py:46
        self.variances = [float(v) for v in hp["variances"]]
        self.topn = int(hp["test_nms_topn"])
        self.iou_threshold = float(iou_threshold)
        self.score_threshold = float(score_threshold)
        self.max_batch = int(max_batch)
        # precision "f16x3" only: ``wait()`` / ``flush_distributed()`` -- the points where a caller collects results --
        # read the device's float16 range word and raise FloatingPointError instead of handing out invalid proposals.
        # The launches themselves (``propose_async``, ``propose_distributed_pipelined``) never synchronise.
        self.check_range = bool(check_range)
        F, K = self.rpn_model.feature_map_shape, self.rpn_model.anchor_count
        self.total_anchors = F * F * K
        dev = "cuda"
        # preallocated outputs: a step performs no allocation and no host synchronisation
        self._reg = torch.empty((max_batch, F, F, 4 * K), dtype=torch.float32, device=dev)
        self._cls = torch.empty((max_batch, F, F, K), dtype=torch.float32, device=dev)
        M = self.topn
        self._boxes = torch.zeros((max_batch, M, 4), dtype=torch.float32, device=dev)
        self._scores = torch.zeros((max_batch, M), dtype=torch.float32, device=dev)
        self._idx = torch.full((max_batch, M), -1, dtype=torch.int32, device=dev)
        self._valid = torch.zeros((max_batch,), dtype=torch.int32, device=dev)
        _keep, self._vptr = L.host_floats(self.variances)
        self._vkeep = _keep
        # NMS scratch (rpn_nms_workspace_bytes): non-zero for few images with many anchors, where several workgroups per
        # image share the passes over the scores.  One buffer: a Proposer's NMS launches are ordered on one stream.
        # The size depends on the RUNTIME batch (the cluster scratch exists for few images with many anchors only), so it is
        # the maximum over every batch size this object accepts, not the value at max_batch.
        self._nms_ws_bytes = max(int(L.lib().rpn_nms_workspace_bytes(b, self.total_anchors, 1, M, M))
                                 for b in range(1, self.max_batch + 1))
        self._nms_ws = torch.empty((max(self._nms_ws_bytes, 16),), dtype=torch.uint8, device=dev)
        # optional 2-stage pipeline across batches: the NMS of batch k (one workgroup per image: 8 of 256 CUs)
        # runs on a side stream while the conv stack of batch k+1 runs on the main stream.  Head outputs and
        # proposal buffers are double buffered; ordering is by HIP events only (no host synchronisation).
        self.overlap_nms = bool(overlap_nms)
        if self.overlap_nms:
            import os
            self._nms_stream = _concurrent_side_stream(should=tuple(avoid_streams))
            # Hardware queues (measured): HIP maps streams onto four queues in creation order, and whatever the
            # process created before -- other streams, an RCCL communicator's internal ones -- decides whether a new stream shares
            # the conv stream's queue; on a shared queue the NMS runs BEHIND the convs (3.25 vs 3.05 ms per VGG16 step, 0.253 vs
            # 0.187 ms at one MobileNetV2 image), silently.  _concurrent_side_stream TESTS each candidate against the current
            # stream (one sleeping wave on each) and keeps the first that runs beside it; ``avoid_streams`` are tested too (best
            # effort: ProposerPool's other pipelines).  Both streams have run something when this returns, so that a communicator
            # created later cannot take the queue in between.
            with torch.cuda.stream(self._nms_stream):
                self._valid.zero_()
            torch.cuda.current_stream().wait_stream(self._nms_stream)
            # "go" handshake (speed only; RPN_NMS_HANDSHAKE=0 turns it off for A/B runs): the conv stack of batch k+1
            # is ordered behind the side stream's WAIT for batch k's head outputs, so that the NMS workgroups (1024
            # threads x 127 VGPRs: each needs a CU to itself) are dispatched at the kernel boundary, while the CUs are
            # empty.  Without it the first conv kernel of batch k+1 wins that race, fills every CU, and the NMS only
            # gets in at the NEXT boundary -- beside a persistent conv layer whose statically scheduled workgroups then
            # wait for "their" CU (block2_conv1: 0.21 -> 0.31 ms).  A higher stream priority alone does not help.
            # Needed where a step's first conv kernel fills every CU for most of the NMS's run time and is followed by
            # persistent layers (VGG16: block 1 in one launch, 0.44 ms).  MobileNetV2's first kernels are short and its
            # blocks are scheduled dynamically: there the extra event hop only costs (one image: 0.320 vs 0.303 ms).
            hs = os.environ.get("RPN_NMS_HANDSHAKE")
            self._nms_handshake = (int(hs) != 0) if hs is not None else (backbone == "vgg16")
            self._nms_go = None
            self._slot = 0
            self._bufs = []
            for _ in range(2):
                self._bufs.append({
                    "reg": torch.empty_like(self._reg), "cls": torch.empty_like(self._cls),
                    "boxes": torch.zeros_like(self._boxes), "scores": torch.zeros_like(self._scores),
                    "idx": torch.full_like(self._idx, -1), "valid": torch.zeros_like(self._valid),
                    "conv_done": torch.cuda.Event(), "nms_done": torch.cuda.Event(), "used": False})

    # -- the hot path ---------------------------------------------------------------------
    def forward(self, imgs):
        """imgs: CUDA float32 (B,img,img,3).  Returns views (deltas (B,A,4), objectness (B,A))."""
        B = self._check_imgs(imgs)
        reg, cls = self._reg[:B], self._cls[:B]
        self.rpn_model.forward_into(imgs, reg, cls)                           # predictor.py:50
        return reg.view(B, -1, 4), cls.view(B, -1)                             # predictor.py:52-53

    def decode_nms(self, deltas, scores, B, ob, osc, oi, ov):
        """Fused ``deltas *= variances`` -> decode -> NMS(topn) of (B,A,4) / (B,A) head outputs on the current stream
        (predictor.py:55-56 + utils/bbox_utils.py:48-70 -> rpn_decode_nms)."""
        st = L.lib().rpn_decode_nms(L.ptr(self.anchors), L.ptr(deltas), self._vptr, L.ptr(scores), B,
                                    self.total_anchors, self.topn, self.iou_threshold, self.score_threshold, 1,
                                    L.ptr(ob), L.ptr(osc), L.ptr(oi), L.ptr(ov), L.ptr(self._nms_ws), self._nms_ws_bytes,
                                    L.stream_ptr())
        L.check(st, "rpn_decode_nms")

    def _check_imgs(self, imgs):
        """The C side only sees a raw pointer: refuse anything that is not a contiguous CUDA float32
        (B, img_size, img_size, 3) batch with B <= max_batch (a wrong dtype / layout would be read out of bounds)."""
        s = int(self.hyper_params["img_size"])
        if not (isinstance(imgs, torch.Tensor) and imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
                and imgs.dim() == 4 and tuple(imgs.shape[1:]) == (s, s, 3) and 1 <= int(imgs.shape[0]) <= self.max_batch):
            raise ValueError("imgs must be a contiguous CUDA float32 tensor (B<=%d, %d, %d, 3) NHWC; got %s"
                             % (self.max_batch, s, s, "%s %s" % (tuple(imgs.shape), imgs.dtype)
                                if isinstance(imgs, torch.Tensor) else type(imgs)))
        return int(imgs.shape[0])

    def propose(self, imgs):
        """imgs -> (boxes (B,300,4), scores (B,300), valid (B,) int32, indices (B,300) int32)."""
        if self.overlap_nms:
            out = self.propose_async(imgs)
            self.wait()                               # consumers on the current stream see finished proposals
            return out
        B = int(imgs.shape[0])
        deltas, scores = self.forward(imgs)
        ob, osc, oi, ov = self._boxes[:B], self._scores[:B], self._idx[:B], self._valid[:B]
        self.decode_nms(deltas, scores, B, ob, osc, oi, ov)
        # (this non-overlapped form never synchronises, so it never polls the float16 range word: call ``wait()`` -- which polls --
        # before trusting the proposals of weights that may leave the float16 range)
        return ob, osc, ov, oi

    def propose_async(self, imgs):
        """Pipelined form (needs ``overlap_nms=True``): enqueues the conv stack on the current stream and decode+NMS
        on the side stream and returns the output tensors WITHOUT ordering the current stream behind the NMS, so that
        the next call's conv stack can start while this call's NMS is still running.  The tensors are complete only
        after ``wait()`` (which makes the current stream wait for the NMS of the last call); ``propose`` does both.
        Throughput loops call this and ``wait()`` once at the end."""
        if not self.overlap_nms:
            raise RuntimeError("propose_async needs overlap_nms=True")
        B = self._check_imgs(imgs)
        main = torch.cuda.current_stream()
        buf = self._bufs[self._slot]
        self._slot ^= 1
        if buf["used"]:
            main.wait_event(buf["nms_done"])          # the NMS that last read this slot's head outputs is done
        reg, cls = buf["reg"][:B], buf["cls"][:B]
        if self._nms_go is not None:
            main.wait_event(self._nms_go)
        self.rpn_model.forward_into(imgs, reg, cls)
        buf["conv_done"].record(main)
        ob, osc, oi, ov = buf["boxes"][:B], buf["scores"][:B], buf["idx"][:B], buf["valid"][:B]
        with torch.cuda.stream(self._nms_stream):
            self._nms_stream.wait_event(buf["conv_done"])
            if self._nms_handshake:
                if "go" not in buf:
                    buf["go"] = torch.cuda.Event()
                buf["go"].record(self._nms_stream)
                self._nms_go = buf["go"]
            self.decode_nms(reg.view(B, -1, 4), cls.view(B, -1), B, ob, osc, oi, ov)
            buf["nms_done"].record(self._nms_stream)
        buf["used"] = True
        self._last = buf
        self._last_batch = B
        return ob, osc, ov, oi

    def wait(self):
        """Make the current stream wait for the proposals returned by the last ``propose`` (pipelined mode).  Under
        precision "f16x3" (and ``check_range``) this is also where the float16 range word of the forwards so far is read
        back (one 4-byte copy, synchronising the current stream): FloatingPointError when an activation left the range --
        the proposals are invalid then; use "bf16x3" or "f32" for such weights."""
        if self.overlap_nms and getattr(self, "_last", None) is not None:
            torch.cuda.current_stream().wait_event(self._last["nms_done"])
        self._poll_range()

    def ensure_side_stream(self):
        """Re-test the NMS side stream against the current stream (e.g. after an RCCL communicator or other libraries created their
        streams: hardware queues are assigned in creation order) and draw a new one if the two have come to share a queue.
        Synchronises the device; not for the hot path.  Returns True when the side stream runs beside the current stream."""
        if not self.overlap_nms:
            return False
        cur = torch.cuda.current_stream()
        if _streams_overlap(cur, self._nms_stream):
            return True
        torch.cuda.synchronize()                                            # nothing in flight on the stream being retired
        self._nms_stream = _concurrent_side_stream(must=cur)
        with torch.cuda.stream(self._nms_stream):
            self._valid.zero_()
        cur.wait_stream(self._nms_stream)
        return _streams_overlap(cur, self._nms_stream)

    def _poll_range(self):
        """The float16 range poll copies one word back and synchronises the stream: impossible inside a stream capture (hipGraph),
        where it is skipped -- a captured step must be built with ``check_range=False`` semantics in mind and polled by the caller
        outside the graph (``rpn_model.raise_on_range_error()``)."""
        if self.check_range and self.rpn_model.precision in ("f16x3", "fp16x3"):
            if torch.cuda.is_current_stream_capturing():
                if not getattr(self, "_warned_capture_poll", False):
                    self._warned_capture_poll = True
                    warnings.warn("Proposer(check_range=True): the float16 range word cannot be polled inside a stream capture; "
                                  "a captured step never raises FloatingPointError -- call rpn_model.raise_on_range_error() "
                                  "after replaying the graph, or pass check_range=False to take that on knowingly",
                                  RuntimeWarning, stacklevel=3)
                return
            self.rpn_model.raise_on_range_error()

    def propose_unfused(self, imgs):
        """Same result through the reference's separate calls (decode, then NMS)."""
        B = int(imgs.shape[0])
        deltas, scores = self.forward(imgs)
        deltas = deltas * torch.tensor(self.variances, device=deltas.device)             # predictor.py:55
        boxes = bbox_utils.get_bboxes_from_deltas(self.anchors, deltas)                  # predictor.py:56
        nb, ns, _, nv, ni = bbox_utils.non_max_suppression(
            boxes.view(B, -1, 1, 4), scores.reshape(B, -1, 1), max_output_size_per_class=self.topn,
            max_total_size=self.topn, iou_threshold=self.iou_threshold, score_threshold=self.score_threshold,
            return_indices=True)
        return nb, ns, nv, ni

    def top_k(self, imgs, k=10, return_scores=False):
        """The reference's own selector: tf.nn.top_k + gather (predictor.py:58-60) -> (boxes (B,k,4), indices (B,k))
        [+ scores (B,k)]."""
        B = int(imgs.shape[0])
        deltas, scores = self.forward(imgs)
        boxes = bbox_utils.get_bboxes_from_deltas(self.anchors, deltas, variances=self.variances)
        order = torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :k]    # ties -> lower index
        top = torch.gather(boxes, 1, order.unsqueeze(-1).expand(B, k, 4))
        return (top, order, torch.gather(scores, 1, order)) if return_scores else (top, order)

    # -- multi-GPU collection ---------------------------------------------------------------
    def pack_records(self, boxes, scores, valid, out=None):
        """(B,M,4),(B,M),(B,) -> one record per image: M*5 float32 + the int32 valid count in the last 4 bytes
        (bit copy, no float round trip).  ``out``: a (rows >= B, M*5+1) float32 buffer to fill (rows beyond B are
        left as they are); without it a new tensor is allocated."""
        B, M = int(scores.shape[0]), int(scores.shape[1])
        rec = out[:B] if out is not None else torch.empty((B, M * 5 + 1), dtype=torch.float32, device=boxes.device)
        rec[:, :M * 4] = boxes.reshape(B, M * 4)
        rec[:, M * 4:M * 5] = scores
        rec.view(torch.int32)[:, M * 5] = valid
        return rec

    @staticmethod
    def unpack_records(rec, M):
        B = int(rec.shape[0])
        return (rec[:, :M * 4].reshape(B, M, 4), rec[:, M * 4:M * 5], rec.view(torch.int32)[:, M * 5])

    def _record_buffer(self, slot, rows):
        """Preallocated per-slot record buffers of the distributed step (no allocation inside a step); rows beyond the
        local batch stay zero = records with valid = 0 (padding of uneven shards)."""
        bufs = getattr(self, "_rec_bufs", None)
        if bufs is None:
            bufs = self._rec_bufs = {}
        key = (slot, rows)
        if key not in bufs:
            bufs[key] = torch.zeros((rows, self.topn * 5 + 1), dtype=torch.float32, device="cuda")
        return bufs[key]

    def propose_distributed_pipelined(self, local_imgs, gather_bufs):
        """Pipelined form for throughput runs (needs ``overlap_nms=True``): the convs of batch k run on the current
        stream; its decode+NMS, the record packing and the all-gather run on the side stream (RCCL orders itself
        behind that stream), concurrently with the convs of batch k+1.  Returns the gathered records of batch k-1
        (None on the first call) after making the current stream wait for them; ``flush_distributed`` returns the
        last batch.  ``gather_bufs``: two (world * B, M*5+1) float32 tensors used alternately.  Every rank must make
        the same sequence of calls.

        Lifetime: the returned tensor IS one of the caller's ``gather_bufs`` and is overwritten by the call after the
        next one; copy it if it must outlive that."""
        import torch.distributed as dist
        prev = getattr(self, "_dist_last", None)
        self.propose_async(local_imgs)
        buf, B = self._last, int(self._last_batch)
        slot = getattr(self, "_gather_slot", 0) ^ 1
        self._gather_slot = slot
        out = gather_bufs[slot]
        live = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if live else 1
        rows = int(out.shape[0]) // world                                # per-rank rows (>= B: uneven shards are padded)
        if rows < B:
            raise ValueError("gather buffer holds %d rows per rank, the local batch has %d" % (rows, B))
        with torch.cuda.stream(self._nms_stream):                        # behind this batch's NMS
            rec = self._record_buffer(slot, rows)
            self.pack_records(buf["boxes"][:B], buf["scores"][:B], buf["valid"][:B], out=rec)
            if B < rows:
                rec[B:].zero_()
            if live:
                dist.all_gather_into_tensor(out[:rows * world], rec)
            else:
                out[:rows].copy_(rec)
            if not hasattr(self, "_gather_done"):
                self._gather_done = [torch.cuda.Event(), torch.cuda.Event()]
            done = self._gather_done[slot]
            done.record(self._nms_stream)
        self._dist_last = (out, done)
        if prev is None:
            return None
        torch.cuda.current_stream().wait_event(prev[1])
        return prev[0]

    def flush_distributed(self, gather_bufs=None):
        """Gathered records of the last ``propose_distributed_pipelined`` call (current stream waits for them)."""
        prev = getattr(self, "_dist_last", None)
        self._dist_last = None
        if prev is None:
            return None
        torch.cuda.current_stream().wait_event(prev[1])
        self._poll_range()                             # (f16x3: FloatingPointError instead of invalid records, see wait())
        return prev[0]

    def propose_distributed(self, local_imgs, gather_out=None, total=None):
        """Each rank proposes for its slice; one all-gather (RCCL over xGMI when the backend is
        "nccl") returns every rank's records on every rank.

        ``total`` = global number of images when they were split with ``shard_bounds``: the all-gather needs equal
        sizes, so every rank pads its records to ceil(total / world) rows (zero rows, valid = 0) and the padding is
        dropped after the gather -> (total, M*5+1) in global image order.  Without ``total`` every rank must hold the
        same number of images -> (world * B_local, M*5+1).

        Lifetime: with a world of one process the result is a fresh copy; otherwise it is ``gather_out`` (or a new
        tensor when that is None).  The per-rank record buffer packed on the way is persistent scratch of this
        object and never returned."""
        import torch.distributed as dist
        boxes, scores, valid, _ = self.propose(local_imgs)
        B = int(scores.shape[0])
        live = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if live else 1
        rows = B if total is None else -(-int(total) // world)
        if rows < B:
            raise ValueError("this rank holds %d images but ceil(total / world) = %d" % (B, rows))
        rec = self._record_buffer(2, rows)
        self.pack_records(boxes, scores, valid, out=rec)
        if B < rows:
            rec[B:].zero_()
        if world == 1:
            return rec[:B].clone()                    # (not the hot path: never hand out the persistent scratch buffer)
        if gather_out is None:
            gather_out = torch.empty((world * rows, rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(gather_out[:world * rows], rec)
        return gather_out[:world * rows] if total is None else compact_gathered(gather_out[:world * rows], total, world)


class ProposerPool(object):
    """``n`` independent ``Proposer`` pipelines (each its own model handle, conv stream and NMS side stream) fed ROUND ROBIN: the
    conv stack of batch k + 1 runs beside the conv stack of batch k.  The reference's loop (predictor.py:46-60) handles its batches
    one after another and they are independent; on a 256-CU device the small kernels of a single-image MobileNetV2 step occupy a
    fraction of the chip and wait on one another, so a second batch in flight fills the gaps: one 500 x 500 image 0.186 -> 0.115 ms
    per step, one 1024 x 1024 image 0.302 -> 0.237, MobileNetV2 at batch 8 18.6 k -> 21.9 k images/s (n = 3: no further gain).
    VGG16 at batch 8 does not gain (its persistent kernels own whole CUs): use a plain ``Proposer`` there.

    ``propose_async(imgs)`` returns that batch's output tensors like ``Proposer.propose_async``; they are complete after ``wait()``
    (or after the pool has come round to the same pipeline's NEXT-but-one call, which reuses the buffers: copy what must live longer).
    ``imgs`` must stay unchanged until its step has run (the call only ORDERS the pipeline's stream behind the caller's current
    stream, it does not copy).  Results are bit-identical to a single ``Proposer``'s (same kernels, same handles' contracts)."""

    def __init__(self, n, *args, **kwargs):
        if n < 1:
            raise ValueError("ProposerPool needs n >= 1 pipelines")
        kwargs["overlap_nms"] = True
        # Hardware queues (four by default, the caller's stream holds one): each pipeline's conv stream gets a queue of its own,
        # the pipelines' NMS side streams may SHARE the fourth -- never the caller's: propose_async records its "images ready" event
        # on the caller's stream, and on a shared queue that event would wait behind an NMS (measured: configs[0] 0.164 instead of
        # 0.125 ms per step).  Each new stream is tested against the ones it must run beside (_streams_overlap).
        caller = torch.cuda.current_stream()
        self.pipelines, self.streams, side = [], [], []
        for _ in range(int(n)):
            s = _concurrent_side_stream(must=caller, should=tuple(self.streams + side))
            with torch.cuda.stream(s):
                p = Proposer(*args, avoid_streams=(caller,) + tuple(self.streams), **kwargs)
            caller.wait_stream(s)
            self.pipelines.append(p)
            self.streams.append(s)
            side.append(p._nms_stream)
        self._ready = [torch.cuda.Event() for _ in range(int(n))]
        self._k = 0
        p0 = self.pipelines[0]
        self.topn, self.anchors, self.total_anchors, self.rpn_model = p0.topn, p0.anchors, p0.total_anchors, p0.rpn_model

    def propose_async(self, imgs, ordered=True):
        """``ordered=False``: the caller vouches that ``imgs`` is already complete (e.g. it synchronised after producing it), and the
        pipeline's stream is not ordered behind the current stream (saves an event record + wait per call)."""
        i = self._k % len(self.pipelines)
        self._k += 1
        s = self.streams[i]
        if ordered:                                           # the images are ready where the caller produced them
            self._ready[i].record(torch.cuda.current_stream())
            s.wait_event(self._ready[i])
        # The conv stack reads ``imgs`` on the pipeline's stream, not on the stream that allocated it: tell the caching allocator,
        # or a caller that drops the tensor right after this call (``pool.propose_async(next_batch())``) gets the block handed to
        # its next allocation while the convs are still reading it.
        imgs.record_stream(s)
        with torch.cuda.stream(s):
            return self.pipelines[i].propose_async(imgs)

    def wait(self):
        """Make the current stream wait for every pipeline's last proposals (and poll their float16 range words)."""
        cur = torch.cuda.current_stream()
        for p, s in zip(self.pipelines, self.streams):
            with torch.cuda.stream(s):
                p.wait()
            cur.wait_stream(s)

    def propose(self, imgs):
        out = self.propose_async(imgs)
        self.wait()
        return out


def pad_records(rec, rows):
    """Pad (B, R) records with zero rows (valid = 0) to ``rows`` rows: equal-size contribution of an uneven shard."""
    B = int(rec.shape[0])
    if B == rows:
        return rec
    if B > rows:
        raise ValueError("%d records do not fit %d rows" % (B, rows))
    out = rec.new_zeros((rows, rec.shape[1]))
    out[:B] = rec
    return out


def compact_gathered(gathered, total, world):
    """Drop the padding rows of an all-gather of ``pad_records`` contributions: (world * ceil(total / world), R) ->
    (total, R) in global image order (rank r contributed the images ``shard_bounds(total, world, r)``)."""
    rows = -(-int(total) // int(world))
    if total % world == 0:
        return gathered[:total]
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, world, r)
        parts.append(gathered[r * rows:r * rows + (hi - lo)])
    return torch.cat(parts, dim=0)


def shard_bounds(total, world, rank):
    """Contiguous image slice of rank `rank`: [lo, hi)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def main(argv=None):
    """The reference's ``predictor.py`` script for its custom-image branch (predictor.py:8-9, 32-36, 41-60), without
    the drawing: ``python -m tf_rpn_amd.predictor --backbone vgg16 --images data/images/`` loads
    ``trained/rpn_<backbone>_model_weights.h5`` (Keras checkpoint, ``by_name``), batches the images of the folder
    (PIL, Lanczos resize to img_size), and prints / writes the top-k proposals per image in pixels of the resized image
    (``denormalize_bboxes``).  ``--nms`` returns the NMS(300) proposals instead of the reference's plain top-k."""
    import argparse
    import json
    import os
    import sys

    import numpy as np

    from .utils import data_utils, io_utils
    ap = argparse.ArgumentParser(description="RPN proposals on MI355X")
    ap.add_argument("-handle-gpu", action="store_true", help="accepted for compatibility; no effect")
    ap.add_argument("--backbone", default="mobilenet_v2", metavar=str(list(io_utils.BACKBONES)))
    ap.add_argument("--images", default="data/images/", help="folder of images (the reference's custom_image_path)")
    ap.add_argument("--weights", default=None, help="Keras .h5 / .npz weights (default: trained/rpn_<backbone>_model_weights.h5)")
    ap.add_argument("--synthetic-weights", action="store_true", help="seeded random weights instead of a checkpoint")
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--nms", action="store_true", help="NMS(test_nms_topn) proposals instead of the plain top-k")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f32w", "bf16x3", "f16x3"])
    ap.add_argument("--out", default=None, help="write the proposals as JSON here (default: stdout)")
    args = ap.parse_args(argv)
    io_utils.is_valid_backbone(args.backbone)
    hyper_params = train_utils.get_hyper_params(args.backbone)
    img_size = int(hyper_params["img_size"])
    img_paths = sorted(data_utils.get_custom_imgs(args.images))
    if not img_paths:
        raise SystemExit("no images in %r" % args.images)
    if args.synthetic_weights:
        weights = "synthetic"
    else:
        weights = args.weights or io_utils.get_model_path("rpn", args.backbone)
        if not os.path.exists(weights):
            raise SystemExit("weights file %r not found (train the reference, or pass --weights / --synthetic-weights)" % weights)
    B = max(1, int(args.batch_size))
    state = {"prop": Proposer(args.backbone, hyper_params=hyper_params, weights=weights, precision=args.precision,
                              max_batch=B)}
    results = []
    gen = data_utils.custom_data_generator(img_paths, img_size, img_size)
    batch, names = [], []

    def flush():
        if not batch:
            return
        imgs = torch.from_numpy(np.stack(batch)).cuda()
        n = len(batch)

        def run(prop):                              # ONE pass of the conv stack per batch
            if args.nms:
                boxes, scores, valid, _ = prop.propose(imgs)
                return boxes[:n].cpu().numpy(), scores[:n].cpu().numpy(), valid[:n].cpu().numpy()
            boxes, _order, scores = prop.top_k(imgs, args.top_k, return_scores=True)
            boxes, scores = boxes[:n].cpu().numpy(), scores[:n].cpu().numpy()
            return boxes, scores, np.full((n,), boxes.shape[1])

        prop = state["prop"]
        boxes, scores, valid = run(prop)
        # f16x3 only: the range flag of the pass that just ran (the .cpu() copies above already synchronised).  When this
        # checkpoint's activations leave the float16 range the batch is redone on bfloat16 halves (float32 range), and
        # the rest of the run stays there.
        if prop.rpn_model.precision == "f16x3" and prop.rpn_model.status(reset=True)["f16_range"]:
            print("precision f16x3: activations exceed the float16 range with these weights; switching to bf16x3",
                  file=sys.stderr)
            state["prop"] = prop = Proposer(args.backbone, hyper_params=hyper_params, weights=weights,
                                            precision="bf16x3", max_batch=B)
            boxes, scores, valid = run(prop)
        for i in range(n):
            k = int(valid[i])
            px = np.round(boxes[i, :k] * np.float32(img_size)).astype(int)          # denormalize_bboxes (bbox_utils.py:152-166)
            results.append({"image": names[i], "img_size": img_size, "boxes_y1x1y2x2": px.tolist(),
                            "scores": [float(v) for v in scores[i, :k]]})
        batch.clear()
        names.clear()

    for path, (img, _gt_boxes, _gt_labels) in zip(img_paths, gen):
        batch.append(img)
        names.append(path)
        if len(batch) == B:
            flush()
    flush()
    text = json.dumps(results, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    else:
        print(text)
    return results


if __name__ == "__main__":
    main()
