"""Where does conv_cin3_f32_mfma_kernel differ from float64 / from itself?  (diagnostic, round 6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_gpu_conv as T
from oracle import conv_oracle as cv
rng = np.random.RandomState(1)
for (B, H, W) in ((3, 203, 203), (8, 500, 500)):
    x = rng.uniform(0, 1, size=(B, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64)) * 0.3).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, size=64).astype(np.float32)
    ref = cv.conv2d_nhwc(x, w, b, pad=(1, 1, 1, 1), act="relu", dtype=torch.float64)
    gots = [T._conv_gpu(x, w, b, 1, 1, 1, H, W, "relu") for _ in range(3)]
    for i, got in enumerate(gots):
        err = np.abs(got - ref)
        bad = np.argwhere(err > 1e-4)
        print("shape", (B, H, W), "run", i, "max err", err.max(), "nan", np.isnan(got).sum(), "bad", len(bad), "same as run 0:", np.array_equal(got, gots[0]))
        if len(bad):
            print("  bad by img", np.bincount(bad[:, 0], minlength=B))
            print("  bad rows", np.unique(bad[:, 1])[:40])
            print("  bad cols", np.unique(bad[:, 2])[:40])
    if not np.array_equal(gots[1], gots[0]):
        d = np.argwhere(gots[1] != gots[0])
        print("  run1 != run0 at", len(d), "rows", np.unique(d[:, 1])[:30], "cols", np.unique(d[:, 2])[:30], "ch", np.unique(d[:, 3])[:30])
