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
