"""MobileNetV2 probe: fused inverted-residual kernels vs the layer-by-layer path vs the float64 oracle (small sizes),
then per-op timings at B=8 500x500 and at C5 (1024x1024, 15 anchors, B=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo
from oracle import conv_oracle as cv
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights

quick = "--quick" in sys.argv
for img, B in ([(64, 2), (125, 1)] if quick else [(64, 2), (125, 1), (150, 3), (500, 1)]):
    hp = bo.get_hyper_params("mobilenet_v2", img_size=img, feature_map_shape=None)
    weights = synthetic_weights("mobilenet_v2", hp, seed=7)
    imgs = np.random.RandomState(3).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    ref = cv.rpn_forward("mobilenet_v2", imgs, weights, dtype=torch.float64, return_features=True)
    for keep in (False, True):
        m = RPNModel("mobilenet_v2", hp, precision="f32", max_batch=B, keep_activations=keep)
        m.set_weights(weights)
        reg, cls = m.predict_on_batch(imgs)
        feat = m.get_activation(m.tap_layer, batch=B).cpu().numpy()
        print("img %d B %d %s: ops %d  |feat err| %.3e (scale %.2f)  reg %.3e cls %.3e" % (
            img, B, "layerwise" if keep else "fused", len(m.ops()), np.abs(feat - ref[2]).max(), np.abs(ref[2]).max(),
            np.abs(reg - ref[0]).max(), np.abs(cls - ref[1]).max()), flush=True)

def bench(hp, B, precision, tag):
    weights = synthetic_weights("mobilenet_v2", hp, seed=1)
    m = RPNModel("mobilenet_v2", hp, precision=precision, max_batch=B)
    m.set_weights(weights)
    x = torch.rand((B, hp["img_size"], hp["img_size"], 3), device="cuda")
    F, K = m.feature_map_shape, m.anchor_count
    reg = torch.empty((B, F, F, 4 * K), device="cuda"); cls = torch.empty((B, F, F, K), device="cuda")
    for _ in range(5): m.forward_into(x, reg, cls)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n): m.forward_into(x, reg, cls)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    m.set_profiling(5)
    for _ in range(5): m.forward_into(x, reg, cls)
    torch.cuda.synchronize()
    ms, _ = m.profile_ms()
    print("== %s %s B=%d: %.3f ms per forward = %.0f images/s (conv stack only), %d launches" % (tag, precision, B, dt * 1e3, B / dt, len(ms)))
    for op, t in zip(m.ops(), ms):
        tf = op["flops_per_image"] * B / (t * 1e-3) / 1e12 if t > 0 else 0
        gb = op["bytes_per_image"] * B / (t * 1e-3) / 1e9 if t > 0 else 0
        print("   %-26s %-34s %7.3f ms %7.2f TF/s %8.1f GB/s" % (op["name"], op["kernel"], t, tf, gb))

hp500 = dict(bo.get_hyper_params("mobilenet_v2"))
hpc5 = dict(bo.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1. / 3.]))
for prec in ("f32", "f16x3"):
    bench(hp500, 8, prec, "500x500")
    bench(hpc5, 1, prec, "c5 1024x1024")
