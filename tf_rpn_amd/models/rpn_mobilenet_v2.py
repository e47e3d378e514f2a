"""MobileNetV2-backbone RPN on MI355X -- counterpart of the reference's ``models/rpn_mobilenet_v2.py``.

``get_model(hyper_params)`` keeps the reference's signature and return pair
(models/rpn_mobilenet_v2.py:6-22): an object with ``predict_on_batch(imgs) -> [rpn_reg, rpn_cls]``
and a handle on the feature-extractor tap (``block_13_expand_relu``, 576 channels, stride 16).  The graph (stem conv, 13 inverted-residual
blocks with BatchNorm folded into the weights, ``rpn_conv`` / ``rpn_cls`` / ``rpn_reg``) is built by the native graph builder in
``csrc/model.hip`` and runs as hand-written gfx950 kernels.
"""
from ._rpn_model import FeatureExtractor, RPNModel, synthetic_weights

BACKBONE = "mobilenet_v2"


def get_model(hyper_params, weights="synthetic", precision="f32", max_batch=8, keep_activations=False, seed=1):
    """weights: "synthetic" (seeded He-normal; the reference would download ImageNet weights,
    which needs network access), None (set them later with ``set_weights`` / ``load_weights``),
    a ``{layer: {...}}`` dict, or a path: a Keras ``.h5`` weights file (the reference's checkpoint; loaded ``by_name``
    like predictor.py:43-44) or a ``.npz`` written by ``RPNModel.save_weights``."""
    rpn_model = RPNModel(BACKBONE, hyper_params, precision=precision, max_batch=max_batch,
                         keep_activations=keep_activations)
    if isinstance(weights, str) and weights == "synthetic":
        rpn_model.set_weights(synthetic_weights(BACKBONE, hyper_params, seed=seed))
    elif isinstance(weights, dict):
        rpn_model.set_weights(weights)
    elif isinstance(weights, str):
        rpn_model.load_weights(weights)
    return rpn_model, FeatureExtractor(rpn_model, rpn_model.tap_layer)


def init_model(model):
    """The reference builds the Keras graph with one dummy call (models/rpn_mobilenet_v2.py:24-29);
    here it warms the kernels up (first launch loads the code object)."""
    import torch
    model.predict_on_batch(torch.rand((1, model.img_size, model.img_size, 3), device="cuda"))
