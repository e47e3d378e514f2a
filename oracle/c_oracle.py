"""ctypes front-end of ``oracle/rpn_oracle.c`` (the plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED.
Used where the numpy/pure-python oracle would be too slow (full-size NMS) and as
the ``cpu_baseline`` "port" in ``bench.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("RPN_ORACLE_SO") or os.path.join(_HERE, "_build", "librpn_oracle.so")   # (env: the sanitizer build)
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "rpn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["san"] if _SO.endswith("_san.so") else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_nms_iou.restype = ctypes.c_float
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def generate_anchors(hyper_params):
    ratios = np.asarray(hyper_params["anchor_ratios"], np.float64)
    scales = np.asarray(hyper_params["anchor_scales"], np.float64)
    fm = int(hyper_params["feature_map_shape"])
    out = np.empty((fm * fm * len(ratios) * len(scales), 4), np.float32)
    lib().orc_generate_anchors(ctypes.c_double(hyper_params["img_size"]), fm, _p(ratios, _f64p), len(ratios),
                               _p(scales, _f64p), len(scales), _p(out))
    return out


def decode(anchors, deltas, variances=None):
    anchors, deltas = _f32(anchors), _f32(deltas)
    B, A = deltas.shape[0], deltas.shape[1]
    out = np.empty_like(deltas)
    v = _f32(variances) if variances is not None else None
    lib().orc_decode(_p(anchors), int(anchors.ndim == 3), _p(deltas), _p(v) if v is not None else None,
                     B, A, _p(out))
    return out


def encode(bboxes, gt):
    bboxes, gt = _f32(bboxes), _f32(gt)
    B, A = gt.shape[0], gt.shape[1]
    out = np.empty_like(gt)
    lib().orc_encode(_p(bboxes), int(bboxes.ndim == 3), _p(gt), B, A, _p(out))
    return out


def scale_boxes(bboxes, height, width, denormalize):
    b = _f32(bboxes)
    out = np.empty_like(b)
    lib().orc_scale_boxes(_p(b), ctypes.c_longlong(b.size // 4), ctypes.c_float(height), ctypes.c_float(width),
                          int(bool(denormalize)), _p(out))
    return out


def iou_map(bboxes, gt):
    bboxes, gt = _f32(bboxes), _f32(gt)
    B, G = gt.shape[0], gt.shape[1]
    A = bboxes.shape[-2]
    out = np.empty((B, A, G), np.float32)
    lib().orc_iou_map(_p(bboxes), int(bboxes.ndim == 3), A, _p(gt), B, G, _p(out))
    return out


def combined_nms(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold=0.5,
                 score_threshold=float("-inf"), pad_per_class=False, clip_boxes=True):
    boxes, scores = _f32(boxes), _f32(scores)
    B, N, q, _ = boxes.shape
    C = scores.shape[2]
    M = int(max_total_size)
    if pad_per_class:
        M = min(M, int(max_output_size_per_class) * C)
    ob = np.empty((B, M, 4), np.float32)
    osc = np.empty((B, M), np.float32)
    oc = np.empty((B, M), np.float32)
    oi = np.empty((B, M), np.int32)
    ov = np.empty((B,), np.int32)
    lib().orc_combined_nms(_p(boxes), _p(scores), B, N, q, C, int(max_output_size_per_class), M,
                           ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold), int(bool(clip_boxes)),
                           _p(ob), _p(osc), _p(oc), _p(oi, _i32p), _p(ov, _i32p))
    return ob, osc, oc, ov, oi


_ACT = {"linear": 0, None: 0, "relu": 1, "sigmoid": 2, "relu6": 3}


def conv2d(x, w, bias, stride=1, pad_t=0, pad_l=0, out_hw=None, depthwise=False, act="linear"):
    """NHWC x HWIO direct convolution with double accumulation (small inputs only)."""
    x, w = _f32(x), _f32(w)
    B, H, W, Cin = x.shape
    R, S = w.shape[0], w.shape[1]
    Cout = Cin if depthwise else w.shape[3]
    OH, OW = out_hw
    out = np.empty((B, OH, OW, Cout), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv2d(_p(x), B, H, W, Cin, _p(w), _p(b) if b is not None else None, R, S, Cout, stride,
                     pad_t, pad_l, OH, OW, int(depthwise), _ACT[act], _p(out))
    return out


def maxpool2x2(x):
    x = _f32(x)
    B, H, W, C = x.shape
    out = np.empty((B, H // 2, W // 2, C), np.float32)
    lib().orc_maxpool2x2(_p(x), B, H, W, C, _p(out))
    return out
