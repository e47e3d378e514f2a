"""VGG16-backbone RPN on MI355X -- counterpart of the reference's ``models/rpn_vgg16.py``.

``get_model(hyper_params)`` keeps the reference's signature and return pair
(models/rpn_vgg16.py:6-22): an object with ``predict_on_batch(imgs) -> [rpn_reg, rpn_cls]``
and a handle on the feature-extractor tap (``block5_conv3``).  The graph (13 conv3x3+ReLU,
4 max-pools, ``rpn_conv`` / ``rpn_cls`` / ``rpn_reg``) is built by the native graph builder in
``csrc/model.hip`` and runs as hand-written gfx950 kernels.
"""
from ._rpn_model import FeatureExtractor, RPNModel, synthetic_weights

BACKBONE = "vgg16"


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
    """The reference builds the Keras graph with one dummy call (models/rpn_vgg16.py:24-29);
    here it warms the kernels up (first launch loads the code object)."""
    import torch
    model.predict_on_batch(torch.rand((1, model.img_size, model.img_size, 3), device="cuda"))
