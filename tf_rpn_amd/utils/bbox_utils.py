"""Box math of the RPN proposal path on MI355X -- same call signatures as the reference's
``utils/bbox_utils.py``, each bound to one C-ABI entry point of ``librpn_hip.so``:

    generate_anchors(hyper_params)                  -> rpn_generate_anchors   (bbox_utils.py:23-46)
    get_bboxes_from_deltas(anchors, deltas)         -> rpn_decode             (bbox_utils.py:72-96)
    get_deltas_from_bboxes(bboxes, gt_boxes)        -> rpn_encode             (bbox_utils.py:98-124)
    generate_iou_map(bboxes, gt_boxes)              -> rpn_iou_map            (bbox_utils.py:126-150)
    normalize_bboxes / denormalize_bboxes(...)      -> rpn_scale_boxes        (bbox_utils.py:152-182)
    non_max_suppression(pred_bboxes, pred_labels, **kwargs) -> rpn_combined_nms (bbox_utils.py:48-70)

Arguments may be torch tensors (any device; results come back as CUDA tensors) or numpy
arrays (results come back as numpy).  Everything runs on the current torch HIP stream.
There is no CPU path: without a GPU these functions raise ``RuntimeError``.
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L


def generate_anchors(hyper_params, as_numpy=False):
    """All anchors for the feature map, (F*F*K, [y1, x1, y2, x2]) normalised to [0, 1]."""
    ratios = np.ascontiguousarray(hyper_params["anchor_ratios"], dtype=np.float64)
    scales = np.ascontiguousarray(hyper_params["anchor_scales"], dtype=np.float64)
    fm = int(hyper_params["feature_map_shape"])
    L.require_gpu()
    out = torch.empty((fm * fm * len(ratios) * len(scales), 4), dtype=torch.float32, device="cuda")
    st = L.lib().rpn_generate_anchors(float(hyper_params["img_size"]), fm,
                                      ratios.ctypes.data_as(L.c_double_p), len(ratios),
                                      scales.ctypes.data_as(L.c_double_p), len(scales), L.ptr(out), L.stream_ptr())
    L.check(st, "generate_anchors")
    return L.from_device(out, as_numpy)


def get_bboxes_from_deltas(anchors, deltas, variances=None):
    """Decode (B, A, [dy, dx, dh, dw]) against anchors (A,4) or (B,A,4) -> (B, A, [y1, x1, y2, x2]).

    ``variances`` (4 floats) optionally fuses the caller's ``deltas *= variances``
    (predictor.py:55) into the same kernel; the default keeps the reference's two-step contract.
    """
    d, was_np = L.to_device(deltas)
    a, _ = L.to_device(anchors)
    if d.dim() != 3 or d.shape[-1] != 4:
        raise ValueError("deltas must be (batch, total_bboxes, 4), got %s" % (tuple(d.shape),))
    B, A = int(d.shape[0]), int(d.shape[1])
    if a.dim() == 2:
        batched = 0
        ok = tuple(a.shape) == (A, 4)
    else:
        batched = 1
        ok = tuple(a.shape) == (B, A, 4)
    if not ok:
        raise ValueError("anchors %s do not match deltas %s" % (tuple(a.shape), tuple(d.shape)))
    out = torch.empty_like(d)
    vptr = None
    if variances is not None:
        _keep, vptr = L.host_floats(variances)
    st = L.lib().rpn_decode(L.ptr(a), batched, L.ptr(d), vptr, B, A, L.ptr(out), L.stream_ptr())
    L.check(st, "get_bboxes_from_deltas")
    return L.from_device(out, was_np)


def get_deltas_from_bboxes(bboxes, gt_boxes):
    """Encode gt_boxes (B,A,4) against bboxes (A,4) or (B,A,4) -> (B, A, [dy, dx, dh, dw])."""
    g, was_np = L.to_device(gt_boxes)
    b, _ = L.to_device(bboxes)
    if g.dim() != 3 or g.shape[-1] != 4:
        raise ValueError("gt_boxes must be (batch, total_bboxes, 4), got %s" % (tuple(g.shape),))
    B, A = int(g.shape[0]), int(g.shape[1])
    batched = 0 if b.dim() == 2 else 1
    if tuple(b.shape) != ((A, 4) if batched == 0 else (B, A, 4)):
        raise ValueError("bboxes %s do not match gt_boxes %s" % (tuple(b.shape), tuple(g.shape)))
    out = torch.empty_like(g)
    st = L.lib().rpn_encode(L.ptr(b), batched, L.ptr(g), B, A, L.ptr(out), L.stream_ptr())
    L.check(st, "get_deltas_from_bboxes")
    return L.from_device(out, was_np)


def generate_iou_map(bboxes, gt_boxes):
    """Pairwise IoU of bboxes (A,4) or (B,A,4) with gt_boxes (B,G,4) -> (B, A, G)."""
    g, was_np = L.to_device(gt_boxes)
    b, _ = L.to_device(bboxes)
    if g.dim() != 3 or g.shape[-1] != 4:
        raise ValueError("gt_boxes must be (batch, total_gt_boxes, 4), got %s" % (tuple(g.shape),))
    B, G = int(g.shape[0]), int(g.shape[1])
    batched = 0 if b.dim() == 2 else 1
    A = int(b.shape[-2])
    if b.shape[-1] != 4 or (batched and int(b.shape[0]) != B):
        raise ValueError("bboxes %s do not match gt_boxes %s" % (tuple(b.shape), tuple(g.shape)))
    out = torch.empty((B, A, G), dtype=torch.float32, device="cuda")
    st = L.lib().rpn_iou_map(L.ptr(b), batched, A, L.ptr(g), B, G, L.ptr(out), L.stream_ptr())
    L.check(st, "generate_iou_map")
    return L.from_device(out, was_np)


def _scale_boxes(bboxes, height, width, denormalize):
    b, was_np = L.to_device(bboxes)
    if b.shape[-1] != 4:
        raise ValueError("bboxes must end in 4 coordinates, got %s" % (tuple(b.shape),))
    out = torch.empty_like(b)
    st = L.lib().rpn_scale_boxes(L.ptr(b), b.numel() // 4, float(height), float(width), int(denormalize), L.ptr(out),
                                 L.stream_ptr())
    L.check(st, "denormalize_bboxes" if denormalize else "normalize_bboxes")
    return L.from_device(out, was_np)


def normalize_bboxes(bboxes, height, width):
    """(…, [y1, x1, y2, x2]) in pixels -> normalised [0, 1] (utils/bbox_utils.py:152-166)."""
    return _scale_boxes(bboxes, height, width, False)


def denormalize_bboxes(bboxes, height, width):
    """Normalised boxes -> pixels, rounded half-to-even like tf.round (utils/bbox_utils.py:168-182)."""
    return _scale_boxes(bboxes, height, width, True)


_NMS_KWARGS = ("max_output_size_per_class", "max_total_size", "iou_threshold", "score_threshold",
               "pad_per_class", "clip_boxes", "name")


def non_max_suppression(pred_bboxes, pred_labels, **kwargs):
    """Per-image, per-class greedy NMS with the keyword arguments of
    ``tf.image.combined_non_max_suppression`` (the reference passes **kwargs straight through).

    pred_bboxes (B, N, q, 4) with q == 1 or q == total_labels; pred_labels (B, N, total_labels).
    Returns (nmsed_boxes (B,M,4), nmsed_scores (B,M), nmsed_classes (B,M), valid_detections (B,) int32);
    entries past valid_detections[i] are zero padding.  ``return_indices=True`` (extension, not in
    TF) appends the selected box indices (B,M) int32, -1 padded.
    """
    return_indices = bool(kwargs.pop("return_indices", False))
    unknown = [k for k in kwargs if k not in _NMS_KWARGS]
    if unknown:
        raise TypeError("non_max_suppression() got unexpected keyword argument(s) %s" % unknown)
    if "max_output_size_per_class" not in kwargs or "max_total_size" not in kwargs:
        raise TypeError("non_max_suppression() needs max_output_size_per_class and max_total_size")
    max_per_class = int(kwargs["max_output_size_per_class"])
    max_total = int(kwargs["max_total_size"])
    iou_thr = float(kwargs.get("iou_threshold", 0.5))
    score_thr = float(kwargs.get("score_threshold", float("-inf")))
    pad_per_class = bool(kwargs.get("pad_per_class", False))
    clip_boxes = bool(kwargs.get("clip_boxes", True))

    boxes, was_np = L.to_device(pred_bboxes)
    scores, _ = L.to_device(pred_labels)
    if boxes.dim() != 4 or boxes.shape[-1] != 4 or scores.dim() != 3:
        raise ValueError("expected pred_bboxes (B,N,q,4) and pred_labels (B,N,C), got %s and %s"
                         % (tuple(boxes.shape), tuple(scores.shape)))
    B, N, q = int(boxes.shape[0]), int(boxes.shape[1]), int(boxes.shape[2])
    C = int(scores.shape[2])
    if tuple(scores.shape[:2]) != (B, N) or q not in (1, C):
        raise ValueError("pred_bboxes %s and pred_labels %s are inconsistent" % (tuple(boxes.shape), tuple(scores.shape)))
    if max_per_class < 0 or max_total < 0:
        raise ValueError("max_output_size_per_class and max_total_size must be >= 0")
    M = min(max_total, max_per_class * C) if pad_per_class else max_total

    out_boxes = torch.zeros((B, M, 4), dtype=torch.float32, device="cuda")
    out_scores = torch.zeros((B, M), dtype=torch.float32, device="cuda")
    out_classes = torch.zeros((B, M), dtype=torch.float32, device="cuda")
    out_idx = torch.full((B, M), -1, dtype=torch.int32, device="cuda")
    out_valid = torch.zeros((B,), dtype=torch.int32, device="cuda")
    if B > 0 and M > 0:
        lib = L.lib()
        ws_bytes = int(lib.rpn_nms_workspace_bytes(B, N, C, max_per_class, M))
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device="cuda")
        st = lib.rpn_combined_nms(L.ptr(boxes), L.ptr(scores), B, N, q, C, max_per_class, M, iou_thr, score_thr,
                                  int(clip_boxes), L.ptr(out_boxes), L.ptr(out_scores), L.ptr(out_classes),
                                  L.ptr(out_idx), L.ptr(out_valid), L.ptr(ws), ws_bytes, L.stream_ptr())
        L.check(st, "non_max_suppression")
    res = (out_boxes, out_scores, out_classes, out_valid)
    if return_indices:
        res = res + (out_idx,)
    return tuple(L.from_device(t, was_np) for t in res)


def decode_and_nms(anchors, deltas, scores, variances, max_total_size, iou_threshold=0.5,
                   score_threshold=float("-inf"), clip_boxes=True):
    """predictor.py:52-56 followed by NMS, in one kernel: raw head deltas (B,A,4) and objectness
    (B,A) -> (boxes (B,M,4), scores (B,M), indices (B,M) int32, valid (B,) int32).  Decoded boxes
    are never written to HBM (-> rpn_decode_nms)."""
    d, was_np = L.to_device(deltas)
    s, _ = L.to_device(scores)
    a, _ = L.to_device(anchors)
    B, A = int(d.shape[0]), int(d.shape[1])
    if tuple(a.shape) != (A, 4) or tuple(s.shape) != (B, A) or d.shape[-1] != 4:
        raise ValueError("inconsistent shapes: anchors %s deltas %s scores %s"
                         % (tuple(a.shape), tuple(d.shape), tuple(s.shape)))
    M = int(max_total_size)
    out_boxes = torch.zeros((B, M, 4), dtype=torch.float32, device="cuda")
    out_scores = torch.zeros((B, M), dtype=torch.float32, device="cuda")
    out_idx = torch.full((B, M), -1, dtype=torch.int32, device="cuda")
    out_valid = torch.zeros((B,), dtype=torch.int32, device="cuda")
    vptr = None
    if variances is not None:
        _keep, vptr = L.host_floats(variances)
    if B > 0 and M > 0:
        ws_bytes = int(L.lib().rpn_nms_workspace_bytes(B, A, 1, M, M))
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device="cuda")
        st = L.lib().rpn_decode_nms(L.ptr(a), L.ptr(d), vptr, L.ptr(s), B, A, M, float(iou_threshold),
                                    float(score_threshold), int(bool(clip_boxes)), L.ptr(out_boxes),
                                    L.ptr(out_scores), L.ptr(out_idx), L.ptr(out_valid), L.ptr(ws), ws_bytes,
                                    L.stream_ptr())
        L.check(st, "decode_and_nms")
    return tuple(L.from_device(t, was_np) for t in (out_boxes, out_scores, out_idx, out_valid))
