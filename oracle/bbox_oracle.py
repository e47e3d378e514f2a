"""numpy float32 restatement of the reference's ``utils/bbox_utils.py`` and the
config dict of ``utils/train_utils.py``.  TEST INFRASTRUCTURE ONLY (see
``oracle/__init__.py``); PARITY UNPINNED (no reference test pins these results).

Every function follows the reference op-for-op, including the dtype each
intermediate has in TF 2.0 eager mode:

* python-float arithmetic stays float64 until it meets a tensor,
* ``tf.sqrt(<python float>)`` converts to float32 first, then takes a float32 sqrt,
* ``int32_tensor / python_int`` is ``tf.truediv`` -> float64,
* everything else is float32 with one rounding per op (no FMA contraction).

Box order is always ``[y1, x1, y2, x2]``.
"""
import copy

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------
# config  (reference: utils/train_utils.py:5-38)
# --------------------------------------------------------------------------
RPN = {
    "vgg16": {
        "img_size": 500,
        "feature_map_shape": 31,
        "anchor_ratios": [1., 2., 1. / 2.],
        "anchor_scales": [128, 256, 512],
    },
    "mobilenet_v2": {
        "img_size": 500,
        "feature_map_shape": 32,
        "anchor_ratios": [1., 2., 1. / 2.],
        "anchor_scales": [128, 256, 512],
    },
}


def get_hyper_params(backbone, **kwargs):
    """utils/train_utils.py:20-38.  kwargs override only keys that already
    exist and only with truthy values (:33-35).  Unlike the reference this
    works on a copy, so tests cannot leak state through the module dict (:28)."""
    hyper_params = copy.deepcopy(RPN[backbone])
    hyper_params["test_nms_topn"] = 300
    hyper_params["total_pos_bboxes"] = 128
    hyper_params["total_neg_bboxes"] = 128
    hyper_params["variances"] = [0.1, 0.1, 0.2, 0.2]
    for key, value in kwargs.items():
        if key in hyper_params and value:
            hyper_params[key] = value
    hyper_params["anchor_count"] = len(hyper_params["anchor_ratios"]) * len(hyper_params["anchor_scales"])
    return hyper_params


# --------------------------------------------------------------------------
# anchors  (reference: utils/bbox_utils.py:3-46)
# --------------------------------------------------------------------------
def generate_base_anchors(hyper_params):
    """utils/bbox_utils.py:3-21."""
    img_size = hyper_params["img_size"]
    base_anchors = []
    for scale in hyper_params["anchor_scales"]:
        scale = scale / img_size                      # :16  python double
        for ratio in hyper_params["anchor_ratios"]:
            w = np.sqrt(F32(scale ** 2 / ratio))      # :18  double -> f32 -> f32 sqrt
            h = F32(w * F32(ratio))                   # :19  f32 * f32(ratio)
            base_anchors.append([-h / F32(2), -w / F32(2), h / F32(2), w / F32(2)])  # :20
    return np.asarray(base_anchors, dtype=F32)        # :21


def generate_anchors(hyper_params):
    """utils/bbox_utils.py:23-46.  Flat index = (y*F + x)*K + k."""
    fm = int(hyper_params["feature_map_shape"])
    stride = 1 / fm                                                   # :35  python double
    grid64 = np.arange(0, fm, dtype=np.int32).astype(np.float64) / np.float64(fm) + stride / 2
    grid_coords = grid64.astype(F32)                                  # :36
    grid_x, grid_y = np.meshgrid(grid_coords, grid_coords)            # :38  'xy' indexing
    flat_x, flat_y = grid_x.reshape(-1), grid_y.reshape(-1)           # :39
    grid_map = np.stack([flat_y, flat_x, flat_y, flat_x], axis=-1)    # :40
    base = generate_base_anchors(hyper_params)                        # :42
    anchors = base.reshape(1, -1, 4) + grid_map.reshape(-1, 1, 4)     # :44  f32 add
    anchors = anchors.reshape(-1, 4).astype(F32)                      # :45
    return np.clip(anchors, F32(0), F32(1))                           # :46


# --------------------------------------------------------------------------
# delta decode / encode  (reference: utils/bbox_utils.py:72-124)
# --------------------------------------------------------------------------
def scale_deltas(deltas, variances):
    """predictor.py:55 -- ``rpn_bbox_deltas *= hyper_params["variances"]``."""
    return (np.asarray(deltas, F32) * np.asarray(variances, F32)).astype(F32)


def get_bboxes_from_deltas(anchors, deltas):
    """utils/bbox_utils.py:72-96.  anchors (A,4) or (B,A,4); deltas (B,A,4)."""
    anchors = np.asarray(anchors, F32)
    deltas = np.asarray(deltas, F32)
    half = F32(0.5)
    w = anchors[..., 3] - anchors[..., 1]                # :81
    h = anchors[..., 2] - anchors[..., 0]                # :82
    cx = anchors[..., 1] + half * w                      # :83
    cy = anchors[..., 0] + half * h                      # :84
    bw = np.exp(deltas[..., 3]) * w                      # :86
    bh = np.exp(deltas[..., 2]) * h                      # :87
    bcx = (deltas[..., 1] * w) + cx                      # :88
    bcy = (deltas[..., 0] * h) + cy                      # :89
    y1 = bcy - (half * bh)                               # :91
    x1 = bcx - (half * bw)                               # :92
    y2 = bh + y1                                         # :93
    x2 = bw + x1                                         # :94
    return np.stack([y1, x1, y2, x2], axis=-1).astype(F32)


def get_deltas_from_bboxes(bboxes, gt_boxes):
    """utils/bbox_utils.py:98-124 (zero-width guards at :117-122)."""
    bboxes = np.asarray(bboxes, F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    half = F32(0.5)
    bw = bboxes[..., 3] - bboxes[..., 1]
    bh = bboxes[..., 2] - bboxes[..., 0]
    bcx = bboxes[..., 1] + half * bw
    bcy = bboxes[..., 0] + half * bh
    gw = gt_boxes[..., 3] - gt_boxes[..., 1]
    gh = gt_boxes[..., 2] - gt_boxes[..., 0]
    gcx = gt_boxes[..., 1] + half * gw
    gcy = gt_boxes[..., 0] + half * gh
    bw = np.where(bw == 0, F32(1e-3), bw).astype(F32)    # :117
    bh = np.where(bh == 0, F32(1e-3), bh).astype(F32)    # :118
    with np.errstate(divide="ignore", invalid="ignore"):
        dx = np.where(gw == 0, F32(0), (gcx - bcx) / bw)         # :119
        dy = np.where(gh == 0, F32(0), (gcy - bcy) / bh)         # :120
        dw = np.where(gw == 0, F32(0), np.log(gw / bw))          # :121
        dh = np.where(gh == 0, F32(0), np.log(gh / bh))          # :122
    return np.stack([dy, dx, dh, dw], axis=-1).astype(F32)


# --------------------------------------------------------------------------
# pairwise IoU  (reference: utils/bbox_utils.py:126-150)
# --------------------------------------------------------------------------
def generate_iou_map(bboxes, gt_boxes):
    """utils/bbox_utils.py:126-150.  bboxes (A,4) or (B,A,4); gt (B,G,4) -> (B,A,G).
    Plain divide: no epsilon, no zero-area guard, no corner canonicalisation."""
    bboxes = np.asarray(bboxes, F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    by1, bx1, by2, bx2 = [bboxes[..., i:i + 1] for i in range(4)]         # :135
    gy1, gx1, gy2, gx2 = [gt_boxes[..., i:i + 1] for i in range(4)]       # :136
    gt_area = ((gy2 - gy1) * (gx2 - gx1))[..., 0]                         # :138
    bbox_area = ((by2 - by1) * (bx2 - bx1))[..., 0]                       # :139
    t = lambda a: np.swapaxes(a, -1, -2)                                  # tf.transpose [0,2,1]
    x_top = np.maximum(bx1, t(gx1))                                       # :141
    y_top = np.maximum(by1, t(gy1))                                       # :142
    x_bot = np.minimum(bx2, t(gx2))                                       # :143
    y_bot = np.minimum(by2, t(gy2))                                       # :144
    inter = np.maximum(x_bot - x_top, F32(0)) * np.maximum(y_bot - y_top, F32(0))   # :146
    union = bbox_area[..., :, None] + gt_area[..., None, :] - inter       # :148
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(F32)                                # :150


# --------------------------------------------------------------------------
# normalise / denormalise  (reference: utils/bbox_utils.py:152-182)
# --------------------------------------------------------------------------
def normalize_bboxes(bboxes, height, width):
    """utils/bbox_utils.py:152-166."""
    b = np.asarray(bboxes, F32)
    return np.stack([b[..., 0] / F32(height), b[..., 1] / F32(width),
                     b[..., 2] / F32(height), b[..., 3] / F32(width)], axis=-1).astype(F32)


def denormalize_bboxes(bboxes, height, width):
    """utils/bbox_utils.py:168-182; tf.round is round-half-to-even, as np.round."""
    b = np.asarray(bboxes, F32)
    out = np.stack([b[..., 0] * F32(height), b[..., 1] * F32(width),
                    b[..., 2] * F32(height), b[..., 3] * F32(width)], axis=-1).astype(F32)
    return np.round(out).astype(F32)


# --------------------------------------------------------------------------
# NMS  (reference: utils/bbox_utils.py:48-70 -> tf.image.combined_non_max_suppression,
#       TF 2.0.0 tensorflow/core/kernels/non_max_suppression_op.cc, un-vendored;
#       semantics restated from SURVEY.md section 8c)
# --------------------------------------------------------------------------
def _mn(a, b):
    """Eigen::numext::mini / std::min: (b < a) ? b : a  (a NaN second operand is ignored)."""
    return b if b < a else a


def _mx(a, b):
    """Eigen::numext::maxi / std::max: (a < b) ? b : a."""
    return b if a < b else a


def nms_iou(bi, bj):
    """TF's NMS-internal IoU: corners canonicalised with min/max, 0 if either
    area <= 0, else inter / (area_i + area_j - inter), all float32."""
    bi = [F32(v) for v in bi]
    bj = [F32(v) for v in bj]
    ymin_i, xmin_i = _mn(bi[0], bi[2]), _mn(bi[1], bi[3])
    ymax_i, xmax_i = _mx(bi[0], bi[2]), _mx(bi[1], bi[3])
    ymin_j, xmin_j = _mn(bj[0], bj[2]), _mn(bj[1], bj[3])
    ymax_j, xmax_j = _mx(bj[0], bj[2]), _mx(bj[1], bj[3])
    with np.errstate(invalid="ignore", over="ignore"):
        area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
        area_j = F32(F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j))
        if area_i <= 0 or area_j <= 0:
            return F32(0)
        iymin, ixmin = _mx(ymin_i, ymin_j), _mx(xmin_i, xmin_j)
        iymax, ixmax = _mn(ymax_i, ymax_j), _mn(xmax_i, xmax_j)
        inter = F32(_mx(F32(iymax - iymin), F32(0)) * _mx(F32(ixmax - ixmin), F32(0)))
        return F32(inter / F32(F32(area_i + area_j) - inter))


def _nms_one_class(boxes, scores, max_out, iou_threshold, score_threshold):
    """Greedy NMS for one (image, class).  Candidates: score > score_threshold
    (strict; NaN never qualifies), visited in descending score, ties lower index
    first (restatement-defined).  A candidate is suppressed iff
    IoU(candidate, s) > iou_threshold (strict) for some already selected s."""
    iou_threshold = F32(iou_threshold)
    score_threshold = F32(score_threshold)
    cand = [i for i in range(len(scores)) if scores[i] > score_threshold]
    cand.sort(key=lambda i: (-float(scores[i]), i))
    selected = []
    for i in cand:
        if len(selected) >= max_out:
            break
        keep = True
        for j in reversed(selected):
            if nms_iou(boxes[i], boxes[j]) > iou_threshold:
                keep = False
                break
        if keep:
            selected.append(i)
    return selected


def combined_non_max_suppression(boxes, scores, max_output_size_per_class, max_total_size,
                                 iou_threshold=0.5, score_threshold=float("-inf"),
                                 pad_per_class=False, clip_boxes=True, return_indices=False):
    """boxes (B,N,q,4) with q in {1,C}; scores (B,N,C).
    Returns (nmsed_boxes (B,M,4) f32, nmsed_scores (B,M) f32, nmsed_classes (B,M) f32,
    valid_detections (B,) int32) [+ indices (B,M) int32, -1 padded].
    M = max_total_size, or min(max_total_size, max_output_size_per_class*C) if pad_per_class."""
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    B, N, q, _ = boxes.shape
    C = scores.shape[2]
    assert q in (1, C)
    M = int(max_total_size)
    if pad_per_class:
        M = min(M, int(max_output_size_per_class) * C)
    out_boxes = np.zeros((B, M, 4), F32)
    out_scores = np.zeros((B, M), F32)
    out_classes = np.zeros((B, M), F32)
    out_idx = np.full((B, M), -1, np.int32)
    valid = np.zeros((B,), np.int32)
    for b in range(B):
        entries = []
        for c in range(C):
            cls_boxes = boxes[b, :, 0 if q == 1 else c, :]
            sel = _nms_one_class(cls_boxes, scores[b, :, c], int(max_output_size_per_class),
                                 iou_threshold, score_threshold)
            for rank, i in enumerate(sel):
                entries.append((-float(scores[b, i, c]), c, rank, i))
        entries.sort()          # score desc; ties: class asc, then selection order (restatement-defined)
        entries = entries[:M]
        valid[b] = len(entries)
        for r, (_, c, _, i) in enumerate(entries):
            bx = boxes[b, i, 0 if q == 1 else c, :]
            if clip_boxes:
                bx = np.minimum(np.maximum(bx, F32(0)), F32(1))
            out_boxes[b, r] = bx
            out_scores[b, r] = scores[b, i, c]
            out_classes[b, r] = F32(c)
            out_idx[b, r] = i
    res = (out_boxes, out_scores, out_classes, valid)
    return res + (out_idx,) if return_indices else res


def non_max_suppression(pred_bboxes, pred_labels, **kwargs):
    """utils/bbox_utils.py:48-70: thin **kwargs pass-through."""
    return combined_non_max_suppression(pred_bboxes, pred_labels, **kwargs)


# --------------------------------------------------------------------------
# predictor glue  (reference: predictor.py:50-60)
# --------------------------------------------------------------------------
def top_k_indices(scores, k):
    """tf.nn.top_k: descending, ties -> lower index (predictor.py:58)."""
    scores = np.asarray(scores, F32)
    order = np.argsort(-scores, axis=-1, kind="stable")
    return order[..., :k].astype(np.int32)


# --------------------------------------------------------------------------
# input preprocessing  (reference: utils/data_utils.py:25-28, 54-68; TF 2.0 kernels restated)
# --------------------------------------------------------------------------
def convert_image_dtype_uint8(img_u8):
    """tf.image.convert_image_dtype(img, tf.float32) for uint8 input (data_utils.py:25):
    cast to float32, then ONE float32 multiply by float32(1/255)."""
    return (np.asarray(img_u8, np.uint8).astype(F32) * F32(1.0 / 255.0)).astype(F32)


def resize_bilinear(img, out_h, out_w):
    """tf.image.resize(img, (h, w)) of TF 2.x (data_utils.py:26): bilinear, half-pixel centres, no antialias.
    Restated from resize_bilinear_op.cc: in = (i + 0.5) * scale - 0.5 (float32, scale = in_size / out_size),
    lower = max(floor(in), 0), upper = min(ceil(in), in_size - 1), lerp = in - floor(in);
    out = top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp, all float32, one rounding per op."""
    img = np.asarray(img, F32)
    H, W = img.shape[0], img.shape[1]

    def weights(out_size, in_size):
        scale = F32(in_size) / F32(out_size)
        i = np.arange(out_size, dtype=F32)
        src = ((i + F32(0.5)) * scale - F32(0.5)).astype(F32)
        fl = np.floor(src)
        lower = np.maximum(fl.astype(np.int64), 0)
        upper = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
        return lower, upper, (src - fl).astype(F32)

    ylo, yhi, yl = weights(out_h, H)
    xlo, xhi, xl = weights(out_w, W)
    tl, tr = img[ylo][:, xlo], img[ylo][:, xhi]
    bl, br = img[yhi][:, xlo], img[yhi][:, xhi]
    xl3, yl3 = xl[None, :, None], yl[:, None, None]
    top = (tl + ((tr - tl) * xl3).astype(F32)).astype(F32)
    bottom = (bl + ((br - bl) * xl3).astype(F32)).astype(F32)
    return (top + ((bottom - top) * yl3).astype(F32)).astype(F32)


def preprocess_image(img_u8, final_height, final_width, flip=False):
    """data_utils.py:25-28: convert to float32 [0,1], resize, optional tf.image.flip_left_right."""
    out = resize_bilinear(convert_image_dtype_uint8(img_u8), final_height, final_width)
    return out[:, ::-1, :].copy() if flip else out


def flip_boxes_horizontally(gt_boxes):
    """data_utils.py:64-67: [y1, 1 - x2, y2, 1 - x1]."""
    g = np.asarray(gt_boxes, F32)
    return np.stack([g[..., 0], F32(1.0) - g[..., 3], g[..., 2], F32(1.0) - g[..., 1]], axis=-1).astype(F32)


# --------------------------------------------------------------------------
# training-target assignment  (reference: utils/train_utils.py:50-65, 84-144)
# --------------------------------------------------------------------------
def randomly_select_xyz_mask(mask, select_xyz, random_ints):
    """utils/train_utils.py:50-65 with the random tensor made an explicit input (``tf.random.uniform(shape,
    minval=1, maxval=max(select)*10, int32)`` cannot be reproduced outside TF).  multiplied = mask * random;
    rank = argsort(argsort(multiplied, DESCENDING)); keep rank < select.  tf.argsort DESCENDING is built on
    top_k, whose ties go to the lower index, i.e. a stable sort of the negated values."""
    mask = np.asarray(mask, bool)
    multiplied = mask.astype(np.int64) * np.asarray(random_ints, np.int64)                       # :61
    sorted_mask = np.argsort(-multiplied, axis=-1, kind="stable")                                # :62
    ranks = np.argsort(sorted_mask, axis=-1, kind="stable")                                      # :63
    select = np.asarray(select_xyz, np.int64).reshape(-1, 1)                                     # :64 expand_dims
    return np.logical_and(mask, ranks < select)                                                  # :65


def calculate_rpn_actual_outputs(anchors, gt_boxes, gt_labels, hyper_params, random_pos, random_neg):
    """utils/train_utils.py:84-144.  anchors (A,4), gt_boxes (B,G,4), gt_labels (B,G) int32 (-1 = padding),
    random_pos / random_neg (B,A) int32 >= 1 stand in for the two tf.random.uniform draws.
    Returns (bbox_deltas (B,A,4) f32, bbox_labels (B,F,F,K) f32 in {1, 0, -1})."""
    anchors = np.asarray(anchors, F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    gt_labels = np.asarray(gt_labels, np.int32)
    B = gt_boxes.shape[0]
    fm, K = hyper_params["feature_map_shape"], hyper_params["anchor_count"]
    total_pos, total_neg = hyper_params["total_pos_bboxes"], hyper_params["total_neg_bboxes"]
    variances = np.asarray(hyper_params["variances"], F32)
    iou_map = generate_iou_map(anchors, gt_boxes)                                                 # :106
    max_indices_each_row = np.argmax(iou_map, axis=2).astype(np.int32)                            # :108 first max
    max_indices_each_column = np.argmax(iou_map, axis=1).astype(np.int32)                         # :110
    merged_iou_map = np.max(iou_map, axis=2)                                                      # :112
    pos_mask = merged_iou_map > F32(0.7)                                                          # :114
    max_pos_mask = np.zeros_like(pos_mask)
    for b in range(B):                                                                            # :116-121 scatter
        for g in range(gt_labels.shape[1]):
            if gt_labels[b, g] != -1:
                max_pos_mask[b, max_indices_each_column[b, g]] = True
    pos_mask = np.logical_or(pos_mask, max_pos_mask)                                              # :122
    pos_mask = randomly_select_xyz_mask(pos_mask, np.array([total_pos]), random_pos)              # :123
    pos_count = pos_mask.sum(axis=-1).astype(np.int32)                                            # :125
    neg_count = (total_pos + total_neg) - pos_count                                               # :126
    neg_mask = np.logical_and(merged_iou_map < F32(0.3), np.logical_not(pos_mask))                # :128
    neg_mask = randomly_select_xyz_mask(neg_mask, neg_count, random_neg)                          # :129
    pos_labels = np.where(pos_mask, F32(1.0), F32(-1.0))                                          # :131
    bbox_labels = (pos_labels + neg_mask.astype(F32)).astype(F32)                                 # :132-133
    gt_boxes_map = np.take_along_axis(gt_boxes, max_indices_each_row[..., None].astype(np.int64).repeat(4, -1), axis=1)
    expanded_gt_boxes = np.where(pos_mask[..., None], gt_boxes_map, F32(0.0)).astype(F32)         # :137
    bbox_deltas = (get_deltas_from_bboxes(anchors, expanded_gt_boxes) / variances).astype(F32)    # :139
    return bbox_deltas, bbox_labels.reshape(B, fm, fm, K)                                         # :142
