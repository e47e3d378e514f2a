"""Independent pin of the bilinear-resize restatement: scikit-image's resize (order 1, no anti-aliasing, edge mode)
uses the same half-pixel-centre sampling grid as TF 2.x's tf.image.resize(method="bilinear") without antialiasing.
Run with the interpreter that has scikit-image:  /opt/conda/bin/python3.9 tests/golden/make_resize_golden.py
(writes tests/golden/resize_skimage.npz: uint8 inputs and float64 outputs)."""
import os

import numpy as np
from skimage.transform import resize

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)
out = {}
for i, ((h, w), (oh, ow)) in enumerate([((17, 23), (30, 30)), ((32, 24), (15, 20)), ((12, 12), (12, 12)), ((5, 9), (20, 18)),
                                        ((48, 36), (20, 20))]):
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    ref = resize(img.astype(np.float64) / 255.0, (oh, ow), order=1, mode="edge", anti_aliasing=False, preserve_range=True)
    out["in%d" % i] = img
    out["out%d" % i] = ref
np.savez_compressed(os.path.join(HERE, "resize_skimage.npz"), **out)
print("wrote", len(out) // 2, "cases")
