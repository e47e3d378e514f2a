"""Command line and file locations of the proposal path -- host-side counterpart of the reference's
``utils/io_utils.py`` (``get_model_path`` :17-29, ``handle_args`` :31-43, ``is_valid_backbone`` :45-50,
``handle_gpu_compatibility`` :52-59).  No device work."""
import argparse
import os

BACKBONES = ("vgg16", "mobilenet_v2")


def get_model_path(model_type, backbone="vgg16"):
    """``trained/<model_type>_<backbone>_model_weights.h5`` relative to the working directory; the ``trained``
    folder is created when missing, as the reference does (io_utils.py:25-28).  ``model_type``: "rpn" | "faster_rcnn"."""
    folder = "trained"
    os.makedirs(folder, exist_ok=True)
    return os.path.join(folder, "%s_%s_model_weights.h5" % (model_type, backbone))


def handle_args(argv=None):
    """The reference's two flags (io_utils.py:36-42): ``-handle-gpu`` (accepted, nothing to do on ROCm) and
    ``--backbone`` (default ``mobilenet_v2``)."""
    parser = argparse.ArgumentParser(description="Region Proposal Network on MI355X")
    parser.add_argument("-handle-gpu", action="store_true", help="accepted for compatibility; no effect")
    parser.add_argument("--backbone", required=False, default="mobilenet_v2", metavar=str(list(BACKBONES)),
                        help="which backbone the RPN uses")
    return parser.parse_args(argv)


def is_valid_backbone(backbone):
    """AssertionError unless ``backbone`` is one of the two the reference knows (io_utils.py:50)."""
    assert backbone in BACKBONES


def handle_gpu_compatibility():
    """TensorFlow memory-growth workaround in the reference (io_utils.py:52-59); there is nothing to configure here:
    device memory is one arena per model handle, sized at creation."""
    return None
