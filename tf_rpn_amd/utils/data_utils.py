"""Input preprocessing of the RPN path on MI355X -- counterpart of the part of the reference's
``utils/data_utils.py`` that sits immediately in front of the model (``preprocessing``, :7-29, and
``flip_horizontally``, :54-68) and of its custom-image branch (``get_custom_imgs`` :106-118,
``custom_data_generator`` :120-136).  Dataset loading through tensorflow_datasets is out of scope.
"""
import os

import numpy as np
import torch

from .. import _lib as L


def preprocess_image(img_u8, final_height, final_width, flip=False, out=None):
    """uint8 (H,W,3) image -> float32 (final_height, final_width, 3) in [0,1] on the device:
    ``tf.image.convert_image_dtype`` + ``tf.image.resize`` (bilinear, half-pixel centres) and, when ``flip``,
    ``tf.image.flip_left_right`` (data_utils.py:25-28).  ``out`` may be a slice of a preallocated batch."""
    x, was_np = L.to_device(img_u8, dtype=torch.uint8)
    if x.dim() != 3 or x.shape[2] != 3:
        raise ValueError("image must be (H, W, 3) uint8, got %s" % (tuple(x.shape),))
    if out is None:
        out = torch.empty((int(final_height), int(final_width), 3), dtype=torch.float32, device="cuda")
    elif tuple(out.shape) != (int(final_height), int(final_width), 3) or not out.is_contiguous():
        raise ValueError("out must be a contiguous (%d,%d,3) float32 tensor" % (final_height, final_width))
    st = L.lib().rpn_preprocess_image(L.ptr(x), int(x.shape[0]), int(x.shape[1]), int(final_height), int(final_width),
                                      int(bool(flip)), L.ptr(out), L.stream_ptr())
    L.check(st, "preprocess_image")
    return L.from_device(out, was_np)


def preprocess_batch(images_u8, final_height, final_width, flips=None):
    """List of uint8 images of any sizes -> one (B, final_height, final_width, 3) float32 device batch
    (what ``padded_batch`` yields after ``preprocessing``, predictor.py:37-39)."""
    L.require_gpu()
    batch = torch.empty((len(images_u8), int(final_height), int(final_width), 3), dtype=torch.float32, device="cuda")
    for i, img in enumerate(images_u8):
        preprocess_image(img, final_height, final_width, flip=bool(flips[i]) if flips is not None else False,
                         out=batch[i])
    return batch


def flip_boxes_horizontally(gt_boxes):
    """(…, [y1, x1, y2, x2]) -> [y1, 1 - x2, y2, 1 - x1] (data_utils.py:64-67); plain tensor arithmetic."""
    g = gt_boxes if isinstance(gt_boxes, torch.Tensor) else torch.as_tensor(gt_boxes, dtype=torch.float32)
    return torch.stack([g[..., 0], 1.0 - g[..., 3], g[..., 2], 1.0 - g[..., 1]], dim=-1)


def get_custom_imgs(custom_image_path):
    """Paths of the files directly inside ``custom_image_path`` (not recursive; data_utils.py:113-118)."""
    for path, _dirs, filenames in os.walk(custom_image_path):
        return [os.path.join(path, name) for name in filenames]
    return []


def custom_data_generator(img_paths, final_height, final_width):
    """Yield ``(img, gt_boxes, gt_labels)`` per file like the reference's generator for its own images
    (data_utils.py:131-136): the image is opened with PIL, resized to (final_width, final_height) with **Lanczos**
    (not the bilinear resize of the dataset branch), converted to float32 in [0,1] (non-RGB files are converted to RGB
    first; the reference assumes RGB); boxes / labels are empty
    placeholders of shapes (1,0) float32 and (0,) int32.  Host arrays: batching moves them to the device."""
    from PIL import Image
    for img_path in img_paths:
        image = Image.open(img_path).convert("RGB")
        resized = image.resize((int(final_width), int(final_height)), Image.LANCZOS)
        img = np.asarray(resized, dtype=np.uint8).astype(np.float32) * np.float32(1.0 / 255.0)   # convert_image_dtype
        yield img, np.zeros((1, 0), np.float32), np.zeros((0,), np.int32)
