"""Configuration of the RPN proposal path: the host-side counterpart of the reference's
``utils/train_utils.py:5-38`` (the ``RPN`` table and ``get_hyper_params``).  No device work.

Only the keys the forward / proposal path reads are documented here; the training-only
counters (``total_pos_bboxes`` / ``total_neg_bboxes``) are carried so that a dict produced
here can be handed to the reference's trainer unchanged.
"""

# stride-16 feature-map side at img_size 500: VGG16 floors 500/16 -> 31 (four 'valid' pools),
# MobileNetV2 ceils -> 32 (four padded stride-2 convs).   utils/train_utils.py:8,14
_FEATURE_MAP_SIDE = {"vgg16": 31, "mobilenet_v2": 32}

RPN = {
    backbone: dict(img_size=500,
                   feature_map_shape=side,
                   anchor_ratios=[1.0, 2.0, 0.5],       # h / w
                   anchor_scales=[128, 256, 512])       # pixels at img_size
    for backbone, side in _FEATURE_MAP_SIDE.items()
}

_FIXED = dict(test_nms_topn=300, total_pos_bboxes=128, total_neg_bboxes=128, variances=[0.1, 0.1, 0.2, 0.2])


def get_hyper_params(backbone, **kwargs):
    """Return the hyper-parameter dict for ``backbone`` ("vgg16" | "mobilenet_v2").

    Behaviour kept from the reference (utils/train_utils.py:20-38), quirks included:
      * the module-level ``RPN[backbone]`` dict itself is updated and returned (:28);
      * a keyword overrides a value only if the key already exists AND the new value is
        truthy (:33-35) -- ``img_size=0`` or an unknown key is silently ignored;
      * ``anchor_count`` is always recomputed as len(ratios) * len(scales) (:37).
    """
    params = RPN[backbone]
    params.update({k: (list(v) if isinstance(v, list) else v) for k, v in _FIXED.items()})
    params.update({k: v for k, v in kwargs.items() if k in params and v})
    params["anchor_count"] = len(params["anchor_ratios"]) * len(params["anchor_scales"])
    return params


def get_step_size(total_items, batch_size):
    """Batches per epoch: ceil(total_items / batch_size) (utils/train_utils.py:40-48)."""
    return -(-int(total_items) // int(batch_size))


def rpn_generator(dataset, anchors, hyper_params):
    """Endless ``(img, (bbox_deltas, bbox_labels))`` stream over ``dataset`` -- any re-iterable of
    ``(img, gt_boxes, gt_labels)`` batches -- with the targets computed on the device
    (utils/train_utils.py:67-82, the consumer of ``calculate_rpn_actual_outputs``)."""
    while True:
        for img, gt_boxes, gt_labels in dataset:
            bbox_deltas, bbox_labels = calculate_rpn_actual_outputs(anchors, gt_boxes, gt_labels, hyper_params)
            yield img, (bbox_deltas, bbox_labels)


def calculate_rpn_actual_outputs(anchors, gt_boxes, gt_labels, hyper_params, random_pos=None, random_neg=None):
    """Training targets for one batch on the device -- utils/train_utils.py:84-144 (with
    ``randomly_select_xyz_mask``, :50-65).

    anchors (A,4); gt_boxes (B,G,4) padded with zeros; gt_labels (B,G) int, -1 = padding.
    Returns (bbox_deltas (B,A,4), bbox_labels (B,F,F,K)) with labels 1 (positive), 0 (negative), -1 (ignored).
    The reference draws two ``tf.random.uniform`` int tensors to subsample positives / negatives; pass them as
    ``random_pos`` / ``random_neg`` ((B,A) int32 >= 1) for reproducible results, otherwise they come from torch's
    generator.  The (B,A,G) IoU map is never materialised.
    """
    import torch

    from .. import _lib as L
    g, was_np = L.to_device(gt_boxes)
    a, _ = L.to_device(anchors)
    lab, _ = L.to_device(gt_labels, dtype=torch.int32)
    if g.dim() != 3 or g.shape[-1] != 4 or tuple(lab.shape) != tuple(g.shape[:2]) or a.dim() != 2 or a.shape[-1] != 4:
        raise ValueError("expected anchors (A,4), gt_boxes (B,G,4), gt_labels (B,G); got %s %s %s"
                         % (tuple(a.shape), tuple(g.shape), tuple(lab.shape)))
    B, G, A = int(g.shape[0]), int(g.shape[1]), int(a.shape[0])
    fm, K = int(hyper_params["feature_map_shape"]), int(hyper_params["anchor_count"])
    if fm * fm * K != A:
        raise ValueError("anchors (%d) do not match feature_map_shape^2 * anchor_count (%d)" % (A, fm * fm * K))
    total_pos, total_neg = int(hyper_params["total_pos_bboxes"]), int(hyper_params["total_neg_bboxes"])

    def given(r, what):
        if not isinstance(r, torch.Tensor):
            import numpy as np
            r = torch.from_numpy(np.ascontiguousarray(np.asarray(r)))
        if tuple(r.shape) != (B, A):
            raise ValueError("random_pos / random_neg must be (B, A)")
        if r.numel() and int(r.min()) < 1:          # key 0 marks a non-candidate: a priority <= 0 would silently drop it
            raise ValueError("%s: priorities must be >= 1 (the reference draws them from [1, maxval))" % what)
        return r.to(device="cuda", dtype=torch.int32).contiguous()

    def draw(maxval):
        """tf.random.uniform(shape, minval=1, maxval=maxval, dtype=int32) (utils/train_utils.py:59-60); ``maxval``
        may be a device scalar (no host synchronisation)."""
        span = (torch.as_tensor(maxval, device="cuda") - 1).clamp(min=1).to(torch.float32)
        return (torch.rand((B, A), device="cuda") * span).floor().to(torch.int32).clamp_(max=span.to(torch.int32) - 1) + 1

    deltas = torch.empty((B, A, 4), dtype=torch.float32, device="cuda")
    labels = torch.empty((B, A), dtype=torch.float32, device="cuda")
    if B > 0:
        if G == 0:
            raise ValueError("gt_boxes needs at least one (possibly padded) row per image")
        lib = L.lib()
        ws_bytes = int(lib.rpn_targets_workspace_bytes(B, A, G))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device="cuda")
        _keep, vptr = L.host_floats(hyper_params["variances"])

        def run(rp, rn):
            st = lib.rpn_rpn_targets(L.ptr(a), L.ptr(g), L.ptr(lab), B, A, G, total_pos, total_neg, vptr, L.ptr(rp),
                                     L.ptr(rn), L.ptr(deltas), L.ptr(labels), L.ptr(ws), ws_bytes, L.stream_ptr())
            L.check(st, "calculate_rpn_actual_outputs")

        # randomly_select_xyz_mask draws with maxval = reduce_max(select_xyz) * 10 PER CALL (:59): total_pos * 10 for the
        # positives (:119), max over the batch of neg_count * 10 for the negatives (:122-125)
        rp = given(random_pos, "random_pos") if random_pos is not None else draw(total_pos * 10)
        if random_neg is not None:
            run(rp, given(random_neg, "random_neg"))
        else:
            # neg_count depends on how many positives survive the subsampling: one pass with placeholder negative
            # priorities yields the positive labels (they do not depend on random_neg), then the real draw
            run(rp, torch.ones((B, A), dtype=torch.int32, device="cuda"))
            pos_count = (labels == 1).sum(dim=1)
            neg_maxval = ((total_pos + total_neg) - pos_count).max() * 10
            run(rp, draw(neg_maxval))
    labels = labels.view(B, fm, fm, K)
    return L.from_device(deltas, was_np), L.from_device(labels, was_np)
