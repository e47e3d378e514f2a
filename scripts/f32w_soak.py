"""f32w soak: the same batch through one model handle N times; every output must equal the first bit for bit (the split-channel
layers' tickets must be back at zero after every launch).  usage: python scripts/f32w_soak.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hp = bo.get_hyper_params("vgg16", img_size=500, feature_map_shape=31)
w = synthetic_weights("vgg16", hp, seed=1)
imgs = np.random.RandomState(0).uniform(0, 1, size=(8, 500, 500, 3)).astype(np.float32)
m = RPNModel("vgg16", hp, precision="f32w", max_batch=8)
m.set_weights(w)
reg0, cls0 = m.predict_on_batch(imgs)
bad = 0
for i in range(N):
    b = 8 if i % 3 else (1 + i % 7)              # smaller batches in between (the workspace is shared by all batch sizes)
    reg, cls = m.predict_on_batch(imgs[:b])
    if not (np.array_equal(reg, reg0[:b]) and np.array_equal(cls, cls0[:b])):
        bad += 1
print("f32w soak: %d forwards, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
