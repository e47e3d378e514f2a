"""tf_rpn_amd -- MI355X (gfx950) native Region Proposal Network forward / proposal path.

Host-side mirror of the reference's interface for that path (same module and function
names as FurkanOM/tf-rpn): ``utils.bbox_utils``, ``utils.train_utils``,
``models.rpn_vgg16`` / ``models.rpn_mobilenet_v2`` and the ``predictor`` loop, all bound
to the C ABI of ``csrc/librpn_hip.so`` (``include/rpn_hip.h``).
"""
__version__ = "0.1.0"
