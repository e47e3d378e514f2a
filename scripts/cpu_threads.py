"""Find a sane torch thread count for the CPU baseline on the GPU box's host (256 logical cores)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo, conv_oracle as cv
from tf_rpn_amd.models._rpn_model import synthetic_weights
hp = bo.get_hyper_params("vgg16")
w = synthetic_weights("vgg16", hp)
img = np.random.RandomState(0).uniform(0, 1, (1, 500, 500, 3)).astype(np.float32)
for n in (16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    cv.rpn_forward("vgg16", img, w)
    t0 = time.perf_counter(); cv.rpn_forward("vgg16", img, w); cv.rpn_forward("vgg16", img, w)
    print(n, "threads: %.3f s/img" % ((time.perf_counter() - t0) / 2), flush=True)
