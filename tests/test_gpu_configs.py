"""GPU parity at the BASELINE.json configurations that round 1 left untested on the device, and the float16-range
safety of the ``f16x3`` conv arithmetic.

* C5 = configs[4]: MobileNetV2, 1024x1024, 3 scales x 5 aspect ratios = 15 anchors per cell (F = 64, A = 61 440), one
  image per GPU.  ``get_hyper_params`` is driven exactly as /root/reference/utils/train_utils.py:33-37 allows
  (keyword overrides of existing keys; ``anchor_count`` recomputed).
* C4 = configs[3]: VGG16 at 500x500, 32 images per GPU (batch 256 on 8 GPUs).
"""
import numpy as np
import pytest
import torch

from oracle import bbox_oracle as bo
from oracle import c_oracle as co
from oracle import conv_oracle as cv
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils

pytestmark = pytest.mark.gpu

C5_RATIOS = [1.0, 2.0, 1.0 / 2.0, 3.0, 1.0 / 3.0]


def _c5_hyper_params():
    hp = dict(train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                           anchor_ratios=list(C5_RATIOS)))
    train_utils.get_hyper_params("mobilenet_v2", img_size=500, feature_map_shape=32,        # the table is a mutated
                                 anchor_ratios=[1.0, 2.0, 0.5])                               # global: put it back
    assert hp["anchor_count"] == 15 and hp["img_size"] == 1024 and hp["feature_map_shape"] == 64
    return hp


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_c5_mobilenet_v2_1024_k15_against_oracle(precision):
    """configs[4], one image: conv stack + head within 1e-4 of torch-CPU (float64 oracle AND float32), then the
    proposal stage bit-exact (indices, valid counts) against the C oracle fed the GPU head outputs."""
    hp = _c5_hyper_params()
    weights = synthetic_weights("mobilenet_v2", hp, seed=1)
    prop = Proposer("mobilenet_v2", hyper_params=dict(hp), weights=weights, precision=precision, max_batch=1,
                    iou_threshold=0.7)
    assert prop.total_anchors == 64 * 64 * 15 == 61440
    imgs = np.random.RandomState(0).uniform(0, 1, size=(1, 1024, 1024, 3)).astype(np.float32)
    x = torch.from_numpy(imgs).cuda()
    boxes, scores, valid, idx = [t.cpu().numpy() for t in prop.propose(x)]
    deltas, obj = [t.cpu().numpy() for t in prop.forward(x)]
    ref = cv.rpn_forward("mobilenet_v2", imgs, weights, dtype=torch.float64)
    assert ref[0].shape == (1, 64, 64, 60) and ref[1].shape == (1, 64, 64, 15)
    e_reg = np.abs(deltas.reshape(ref[0].shape) - ref[0]).max()
    e_cls = np.abs(obj.reshape(ref[1].shape) - ref[1]).max()
    print("c5 %s: max|err| vs float64 oracle: reg %.3e cls %.3e" % (precision, e_reg, e_cls))
    assert e_reg <= 1e-4 and e_cls <= 1e-4
    if precision == "f16x3":
        assert not prop.rpn_model.status()["f16_range"]
    anchors = bo.generate_anchors(hp)
    assert np.array_equal(prop.anchors.cpu().numpy(), anchors)
    dec = co.decode(anchors, deltas, np.float32(hp["variances"]))
    rb, rs, _rc, rv, ri = co.combined_nms(dec[:, :, None, :], obj[:, :, None], 300, 300, iou_threshold=0.7)
    assert np.array_equal(valid, rv) and np.array_equal(idx, ri)
    assert np.abs(boxes - rb).max() <= 1e-4 and np.array_equal(scores, rs)


def test_c4_vgg16_batch32_invariance_and_properties():
    """configs[3] per-GPU shard: 32 images through the f16x3 path.  Image i inside the batch of 32 gives bit-identical
    head outputs and proposals to image i alone (batch 1 handle, same weights); output contract as in C2."""
    hp = dict(bo.get_hyper_params("vgg16"))
    weights = synthetic_weights("vgg16", hp, seed=1)
    big = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=32, iou_threshold=0.7)
    imgs = torch.rand((32, 500, 500, 3), generator=torch.Generator().manual_seed(0)).cuda()
    boxes, scores, valid, idx = [t.clone() for t in big.propose(imgs)]
    deltas, obj = [t.clone() for t in big.forward(imgs)]
    assert not big.rpn_model.status()["f16_range"]
    assert boxes.shape == (32, 300, 4) and valid.dtype == torch.int32
    assert (valid > 0).all() and (valid <= 300).all() and boxes.min() >= 0 and boxes.max() <= 1
    assert torch.isfinite(deltas).all() and torch.isfinite(obj).all()
    for b in range(32):
        v = int(valid[b])
        s = scores[b, :v]
        assert (s[:-1] >= s[1:]).all() and (scores[b, v:] == 0).all() and (idx[b, v:] == -1).all()
        assert len(set(idx[b, :v].tolist())) == v
    one = Proposer("vgg16", hyper_params=dict(hp), weights=weights, precision="f16x3", max_batch=1, iou_threshold=0.7)
    for i in (0, 13, 31):
        xi = imgs[i:i + 1].contiguous()
        b1, s1, v1, i1 = [t.clone() for t in one.propose(xi)]
        d1, o1 = one.forward(xi)
        assert torch.equal(d1[0], deltas[i]) and torch.equal(o1[0], obj[i])
        assert torch.equal(b1[0], boxes[i]) and torch.equal(i1[0], idx[i]) and int(v1[0]) == int(valid[i])
    # oracle on one image of the batch: the box stage bit-exact against the C restatement fed the GPU head outputs
    anchors = bo.generate_anchors(hp)
    dec = co.decode(anchors, deltas[13:14].cpu().numpy(), np.float32(hp["variances"]))
    _rb, _rs, _rc, rv, ri = co.combined_nms(dec[:, :, None, :], obj[13:14].cpu().numpy()[:, :, None], 300, 300,
                                            iou_threshold=0.7)
    assert int(rv[0]) == int(valid[13]) and np.array_equal(ri[0], idx[13].cpu().numpy())


# ---- float16 range safety of the f16x3 arithmetic ------------------------------------------------------------------
def _scaled_pair(weights, first, second, s):
    """ReLU is positively homogeneous: scaling layer `first` (kernel and bias) by s and the kernel of the layer that
    consumes it by 1/s leaves every later activation mathematically unchanged -- only `first`'s output grows by s."""
    w = {k: dict(v) for k, v in weights.items()}
    w[first]["kernel"] = (weights[first]["kernel"] * np.float32(s)).astype(np.float32)
    w[first]["bias"] = (weights[first]["bias"] * np.float32(s)).astype(np.float32)
    w[second]["kernel"] = (weights[second]["kernel"] / np.float32(s)).astype(np.float32)
    return w


def test_f16x3_overflow_is_flagged_never_silent():
    """Activations pushed past 65504 (block2_conv1 scaled by 2^20, block2_conv2 by 2^-20): the exact-f32 and bf16x3
    paths still meet the 1e-4 bound; f16x3 raises the device flag, ``predict_on_batch`` on numpy input refuses to
    return, and the flag is sticky until reset."""
    hp = bo.get_hyper_params("vgg16", img_size=96, feature_map_shape=6)
    base = synthetic_weights("vgg16", hp, seed=1)
    weights = _scaled_pair(base, "block2_conv1", "block2_conv2", 2.0 ** 20)
    imgs = np.random.RandomState(0).uniform(0, 1, size=(2, 96, 96, 3)).astype(np.float32)
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    for precision in ("f32", "bf16x3"):
        model = RPNModel("vgg16", hp, precision=precision, max_batch=2)
        model.set_weights(weights)
        reg, cls = model.predict_on_batch(imgs)
        assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4, precision
        assert not model.status()["f16_range"]
    model = RPNModel("vgg16", hp, precision="f16x3", max_batch=2)
    model.set_weights(weights)
    with pytest.raises(FloatingPointError, match="float16 range"):
        model.predict_on_batch(imgs)
    x = torch.from_numpy(imgs).cuda()
    model.predict_on_batch(x)                                   # CUDA tensors: no host read-back on the hot path ...
    assert model.status()["f16_range"]                          # ... the flag is there for whoever polls
    assert model.status(reset=True)["f16_range"] and not model.status()["f16_range"]
    # the unscaled weights on the same handle: clean again, and within the bound
    model.set_weights(base)
    reg, cls = model.predict_on_batch(imgs)
    ref0 = cv.rpn_forward("vgg16", imgs, base, dtype=torch.float64)
    assert np.abs(reg - ref0[0]).max() <= 1e-4 and np.abs(cls - ref0[1]).max() <= 1e-4


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_split_precision_tiny_inputs_subnormal_halves(precision):
    """Images ~1e-6: the float16 hi halves of the first activations are subnormal and their lo halves vanish.  The
    absolute error this leaves on the head outputs must stay inside the 1e-4 bound (it is absolute, not relative)."""
    hp = bo.get_hyper_params("vgg16", img_size=96, feature_map_shape=6)
    weights = synthetic_weights("vgg16", hp, seed=2)
    for w in weights.values():                                  # biases off so that the tiny signal is all there is
        if "bias" in w:
            w["bias"] = np.zeros_like(w["bias"])
    imgs = (np.random.RandomState(1).uniform(0, 1, size=(1, 96, 96, 3)) * 1e-6).astype(np.float32)
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    model = RPNModel("vgg16", hp, precision=precision, max_batch=1)
    model.set_weights(weights)
    reg, cls = model.predict_on_batch(imgs)
    assert np.isfinite(reg).all() and np.isfinite(cls).all()
    assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_mobilenet_v2_wide_batchnorm_scales(precision):
    """bn_Conv1 scales gamma / sqrt(var + 1e-3) spanning 1e-2 ... 50 (a Keras checkpoint with small moving variances
    gives ~31 already): the power-of-two pre-scale of the float16 weights is derived from the FOLDED kernel, so the hi
    halves stay finite and the outputs stay within the bound -- on the layer-by-layer path (where Conv1 runs on the
    split-precision MFMA kernel under f16x3) and on the fused-block path."""
    hp = bo.get_hyper_params("mobilenet_v2", img_size=96, feature_map_shape=6)
    weights = synthetic_weights("mobilenet_v2", hp, seed=5)
    rng = np.random.RandomState(9)
    # bn_Conv1 is the BatchNorm that is folded into a split-precision (float16-halves) kernel under f16x3.  Only it
    # gets the wide scales: a gain of 50 on EVERY layer would turn the network into a chaotic map in which float32 and
    # float64 arithmetic part ways by construction (rounding errors amplified 50x per layer), which tests nothing.
    w = weights["bn_Conv1"]
    n = w["gamma"].shape[0]
    scale = np.exp(rng.uniform(np.log(1e-2), np.log(50.0), size=n))
    scale[:2] = [1e-2, 50.0]
    w["var"] = rng.uniform(1e-4, 2e-3, size=n).astype(np.float32)                  # tiny moving variances
    w["gamma"] = (scale * np.sqrt(w["var"] + 1e-3)).astype(np.float32)
    imgs = np.random.RandomState(2).uniform(0, 1, size=(2, 96, 96, 3)).astype(np.float32)
    ref = cv.rpn_forward("mobilenet_v2", imgs, weights, dtype=torch.float64)
    for keep in (True, False):                  # layer by layer (Conv1 on its own kernel) / fused blocks
        model = RPNModel("mobilenet_v2", hp, precision=precision, max_batch=2, keep_activations=keep)
        model.set_weights(weights)
        reg, cls = model.predict_on_batch(imgs)
        assert np.isfinite(reg).all() and np.isfinite(cls).all()
        scale = max(1.0, float(np.abs(ref[0]).max()))
        assert np.abs(reg - ref[0]).max() <= 1e-4 * scale and np.abs(cls - ref[1]).max() <= 1e-4, keep
        assert not model.status()["f16_range"]


def test_hot_path_rejects_malformed_batches():
    """``Proposer.forward`` / ``forward_into`` hand raw pointers to the C side: dtype, device, layout and shape are
    checked in Python first (a uint8 / NCHW / oversized batch would otherwise be read out of bounds)."""
    hp = dict(bo.get_hyper_params("vgg16", img_size=64, feature_map_shape=4))
    prop = Proposer("vgg16", hyper_params=dict(hp), max_batch=2)
    good = torch.rand((2, 64, 64, 3), device="cuda")
    prop.propose(good)
    bad = [good.cpu(), good.to(torch.float16), good.permute(0, 3, 1, 2), good[:, :32], torch.rand((3, 64, 64, 3), device="cuda"),
           (good * 255).to(torch.uint8), good.transpose(1, 2)]   # the last one: right shape, not contiguous
    for x in bad:
        with pytest.raises(ValueError):
            prop.propose(x)
    reg = torch.empty((2, 4, 4, 36), device="cuda")
    with pytest.raises(ValueError):
        prop.rpn_model.forward_into(good, reg, torch.empty((2, 4, 4, 8), device="cuda"))


def test_target_assignment_priority_contract():
    """utils/train_utils.py:59-60: each randomly_select_xyz_mask call draws from [1, reduce_max(select_xyz) * 10).
    User-supplied priorities must be >= 1 (0 marks a non-candidate); the default draws honour the per-call maxval and
    select exactly min(total_pos, candidates) positives and 256 - pos negatives."""
    hp = dict(bo.get_hyper_params("vgg16"))
    anchors = bo.generate_anchors(hp)
    rng = np.random.RandomState(4)
    B, G, A = 3, 6, anchors.shape[0]
    gt = np.zeros((B, G, 4), np.float32)
    for b in range(B):
        for g in range(4):
            y, x = rng.uniform(0.05, 0.5, size=2)
            h, w = rng.uniform(0.2, 0.45, size=2)
            gt[b, g] = [y, x, y + h, x + w]
    labels = np.full((B, G), -1, np.int32)
    labels[:, :4] = 1
    deltas, lab = train_utils.calculate_rpn_actual_outputs(anchors, gt, labels, hp)
    lab = lab.reshape(B, -1)
    pos, neg = (lab == 1).sum(1), (lab == 0).sum(1)
    assert (pos <= 128).all() and (pos >= 1).all() and ((pos + neg) == 256).all()
    assert np.isfinite(deltas).all() and (np.abs(deltas[lab != 1]) == 0).all()
    bad = np.ones((B, A), np.int32)
    bad[1, 7] = 0
    with pytest.raises(ValueError, match=">= 1"):
        train_utils.calculate_rpn_actual_outputs(anchors, gt, labels, hp, random_pos=bad, random_neg=np.ones((B, A), np.int32))


H5PY_PYTHON = "/opt/conda/bin/python3.9"          # the image's interpreter that has the real h5py


def _trained_like_vgg16(hp, seed=7):
    """VGG16 + RPN weights with the STATISTICS of a trained ImageNet checkpoint rather than a He initialisation: each
    backbone layer amplifies (kernel gain 1.6 over He-normal, small positive biases), so that on the reference's
    un-normalised [0, 1] input (utils/data_utils.py:25-26: no mean subtraction) the activations grow layer by layer to
    O(10^2 - 10^3) in block 5, as a real VGG16's do; the RPN head brings them back to O(1) deltas and unsaturated
    objectness (rpn_conv gain 0.002), as a trained head does."""
    base = synthetic_weights("vgg16", hp, seed=seed)
    rng = np.random.RandomState(seed)
    w = {}
    for name, entry in base.items():
        e = dict(entry)
        if name.startswith("block"):
            e["kernel"] = (entry["kernel"] * np.float32(1.6)).astype(np.float32)
            e["bias"] = rng.uniform(0.0, 0.1, size=entry["bias"].shape).astype(np.float32)
        elif name == "rpn_conv":
            e["kernel"] = (entry["kernel"] * np.float32(0.002)).astype(np.float32)
        w[name] = e
    return w


def test_f16x3_on_trained_like_statistics_through_the_h5_path(tmp_path):
    """f16x3 on weights with trained-like per-layer gains, loaded the way the reference loads a checkpoint
    (``load_weights(path, by_name=True)``, predictor.py:43-44, Keras .h5 written by the real h5py): head outputs within
    1e-4 of the float64 oracle with the float16 range flag clear -- or, when the flag is raised, the documented bf16x3
    fallback (float32 range) within the bound.  All other f16x3 evidence uses He-init weights whose activations stay O(1)."""
    import os
    import subprocess
    hp = bo.get_hyper_params("vgg16")
    weights = _trained_like_vgg16(hp)
    imgs = np.random.RandomState(3).uniform(0, 1, size=(1, 500, 500, 3)).astype(np.float32)
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64, return_features=True)
    feat_max = float(np.abs(ref[2]).max())
    print("block5_conv3 activations: max %.1f, mean %.2f; |deltas| max %.3f; objectness in [%.3f, %.3f]"
          % (feat_max, float(np.abs(ref[2]).mean()), float(np.abs(ref[0]).max()), float(ref[1].min()), float(ref[1].max())))
    assert 1e2 <= feat_max <= 6e4                       # trained-like magnitudes, still inside float16's range
    assert 0.01 < float(ref[1].min()) and float(ref[1].max()) < 0.99      # an unsaturated head: the 1e-4 bound is meaningful
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        have_h5py = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except OSError:
        have_h5py = False
    model = RPNModel("vgg16", hp, precision="f16x3", max_batch=1)
    if have_h5py:
        npz, h5 = str(tmp_path / "w.npz"), str(tmp_path / "w.h5")
        RPNModel.save_weights(weights, npz)
        r = subprocess.run([H5PY_PYTHON, os.path.join(here, "golden", "npz_to_keras_h5.py"), npz, h5, "fixed"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        done = model.load_weights(h5, by_name=True)
        assert sorted(done) == sorted(l["name"] for l in model.layers)
    else:
        model.set_weights(weights)                      # (no interpreter with h5py here: same arrays, set directly)
    x = torch.from_numpy(imgs).cuda()
    reg, cls = [t.cpu().numpy() for t in model.predict_on_batch(x)]
    flagged = model.status(reset=True)["f16_range"]
    if flagged:                                         # the documented fallback: bfloat16 halves, float32 range
        model = RPNModel("vgg16", hp, precision="bf16x3", max_batch=1)
        model.set_weights(weights)
        reg, cls = model.predict_on_batch(imgs)
    e_reg, e_cls = float(np.abs(reg.reshape(ref[0].shape) - ref[0]).max()), float(np.abs(cls.reshape(ref[1].shape) - ref[1]).max())
    print("f16x3%s: max|err| vs float64 oracle: deltas %.3e, objectness %.3e" % (" -> bf16x3 (range flag)" if flagged else "",
                                                                                 e_reg, e_cls))
    assert e_reg <= 1e-4 and e_cls <= 1e-4
    assert not flagged or feat_max > 6e4                # the flag may only rise when something really left the range


def _trained_like_mobilenet_v2(hp, seed=7):
    """MobileNetV2 + RPN weights with the BatchNorm STATISTICS of a trained checkpoint instead of the near-identity ones of
    ``synthetic_weights`` (gamma, var in [0.8, 1.2]): every conv output channel c carries its own scale s_c, log-uniform in
    [0.01, 100] (kernel column x s_c), and its BatchNorm has co-adapted statistics, moving_variance = s_c^2 v0 and
    moving_mean = s_c m0 (v0 in [0.5, 2], m0 in [-0.3, 0.3]), with gamma in [0.4, 1.2] and beta in [-0.5, 0.5] -- so the
    fold scale gamma / sqrt(var + eps) spans ~3e-3 ... 38 (its top is capped by eps = 1e-3: channels with var << eps are
    where the eps term matters), while the folded network stays as well conditioned as a trained one is (float32 torch-CPU
    agrees with float64 to ~2e-7 on it).  rpn_conv x 0.3: an unsaturated head, so the 1e-4 bound is meaningful."""
    rng = np.random.RandomState(seed)
    w = {}
    ratios = []

    def conv_bn(name, bn_name, shape, fan, ch, depthwise=False):
        s = np.exp(rng.uniform(np.log(0.01), np.log(100.0), size=ch)).astype(np.float32)
        k = (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
        w[name] = {"kernel": (k * (s.reshape(1, 1, ch, 1) if depthwise else s.reshape(1, 1, 1, ch))).astype(np.float32)}
        v0 = rng.uniform(0.5, 2.0, size=ch).astype(np.float32)
        gamma = rng.uniform(0.4, 1.2, size=ch).astype(np.float32)
        w[bn_name] = {"gamma": gamma, "beta": rng.uniform(-0.5, 0.5, size=ch).astype(np.float32),
                      "mean": (s * rng.uniform(-0.3, 0.3, size=ch)).astype(np.float32), "var": (s * s * v0).astype(np.float32)}
        ratios.append(gamma / np.sqrt(w[bn_name]["var"] + np.float32(cv.BN_EPS)))

    conv_bn("Conv1", "bn_Conv1", (3, 3, 3, 32), 27, 32)
    for (bid, cin, t, cout, _stride) in cv.MNV2_BLOCKS:
        p = "expanded_conv_" if bid == 0 else "block_%d_" % bid
        ce = cin * t
        if bid != 0:
            conv_bn(p + "expand", p + "expand_BN", (1, 1, cin, ce), cin, ce)
        conv_bn(p + "depthwise", p + "depthwise_BN", (3, 3, ce, 1), 9, ce, depthwise=True)
        conv_bn(p + "project", p + "project_BN", (1, 1, ce, cout), ce, cout)
    conv_bn("block_13_expand", "block_13_expand_BN", (1, 1, 96, 576), 96, 576)
    K = int(hp["anchor_count"])
    he = lambda shape, fan: (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
    w["rpn_conv"] = {"kernel": he((3, 3, 576, 512), 9 * 576) * np.float32(0.3), "bias": rng.uniform(-0.05, 0.05, 512).astype(np.float32)}
    w["rpn_cls"] = {"kernel": he((1, 1, 512, K), 512), "bias": rng.uniform(-0.05, 0.05, K).astype(np.float32)}
    w["rpn_reg"] = {"kernel": he((1, 1, 512, 4 * K), 512) * np.float32(0.1),
                    "bias": rng.uniform(-0.05, 0.05, 4 * K).astype(np.float32)}
    r = np.concatenate(ratios)
    return w, (float(r.min()), float(r.max()))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_mobilenet_v2_trained_like_batchnorm_through_the_h5_path(tmp_path, precision):
    """The BatchNorm-folded backbone on trained-like statistics (fold scales over four decades, beta != 0, variances below
    eps), loaded the way the reference loads a checkpoint (``load_weights(path, by_name=True)``, predictor.py:43-44; Keras
    .h5 written by the real h5py when an interpreter with it exists): head outputs within 1e-4 of the float64 oracle with
    the float16 range flag clear -- or, when the flag is raised, the documented bf16x3 fallback within the bound."""
    import os
    import subprocess
    hp = bo.get_hyper_params("mobilenet_v2")
    weights, (r_lo, r_hi) = _trained_like_mobilenet_v2(hp)
    assert r_lo < 1e-2 and r_hi > 20.0                   # the fold scales really span the decades
    imgs = np.random.RandomState(3).uniform(0, 1, size=(2, 500, 500, 3)).astype(np.float32)
    ref = cv.rpn_forward("mobilenet_v2", imgs, weights, dtype=torch.float64, return_features=True)
    print("fold scale gamma/sqrt(var+eps) in [%.2e, %.1f]; block_13_expand_relu: max %.2f mean %.3f; |deltas| max %.3f; "
          "objectness in [%.3f, %.3f]" % (r_lo, r_hi, float(ref[2].max()), float(ref[2].mean()), float(np.abs(ref[0]).max()),
                                          float(ref[1].min()), float(ref[1].max())))
    assert 0.01 < float(ref[1].min()) and float(ref[1].max()) < 0.99      # an unsaturated head
    assert float((ref[2] > 0).mean()) > 0.2                                # a live feature map, not a dead one
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        have_h5py = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except OSError:
        have_h5py = False
    model = RPNModel("mobilenet_v2", hp, precision=precision, max_batch=2)
    if have_h5py:
        npz, h5 = str(tmp_path / "w.npz"), str(tmp_path / "w.h5")
        RPNModel.save_weights(weights, npz)
        r = subprocess.run([H5PY_PYTHON, os.path.join(here, "golden", "npz_to_keras_h5.py"), npz, h5, "fixed"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        done = model.load_weights(h5, by_name=True)
        assert sorted(done) == sorted(l["name"] for l in model.layers)
    else:
        model.set_weights(weights)                      # (no interpreter with h5py here: same arrays, set directly)
    x = torch.from_numpy(imgs).cuda()
    reg, cls = [t.cpu().numpy() for t in model.predict_on_batch(x)]
    flagged = precision == "f16x3" and model.status(reset=True)["f16_range"]
    if flagged:                                         # the documented fallback: bfloat16 halves, float32 range
        model = RPNModel("mobilenet_v2", hp, precision="bf16x3", max_batch=2)
        model.set_weights(weights)
        reg, cls = model.predict_on_batch(imgs)
    e_reg = float(np.abs(reg.reshape(ref[0].shape) - ref[0]).max())
    e_cls = float(np.abs(cls.reshape(ref[1].shape) - ref[1]).max())
    print("%s%s: max|err| vs float64 oracle: deltas %.3e, objectness %.3e" % (precision, " -> bf16x3 (range flag)" if flagged else "",
                                                                              e_reg, e_cls))
    assert e_reg <= 1e-4 and e_cls <= 1e-4
    assert not flagged                                   # nothing here leaves the float16 range (ReLU6-bounded activations)


def test_proposer_wait_raises_on_f16_range_error():
    """CUDA-tensor callers of the Proposer: the hot launches never synchronise, but the points where results are collected
    -- ``wait()`` (hence ``propose()`` in pipelined mode) and ``flush_distributed()`` -- read the float16 range word under
    f16x3 and raise FloatingPointError instead of handing out invalid proposals; ``check_range=False`` keeps the old
    poll-it-yourself behaviour; clean weights never raise."""
    hp = bo.get_hyper_params("vgg16", img_size=96, feature_map_shape=6)
    base = synthetic_weights("vgg16", hp, seed=1)
    bad = _scaled_pair(base, "block2_conv1", "block2_conv2", 2.0 ** 20)
    x = torch.from_numpy(np.random.RandomState(0).uniform(0, 1, size=(2, 96, 96, 3)).astype(np.float32)).cuda()
    prop = Proposer("vgg16", hyper_params=dict(hp), weights=bad, precision="f16x3", max_batch=2, overlap_nms=True)
    prop.propose_async(x)                               # the launch itself does not raise (no host read-back)
    with pytest.raises(FloatingPointError, match="float16 range"):
        prop.wait()
    with pytest.raises(FloatingPointError, match="float16 range"):
        prop.propose(x)
    gather = [torch.empty((2, prop.topn * 5 + 1), device="cuda") for _ in range(2)]
    prop.propose_distributed_pipelined(x, gather)
    with pytest.raises(FloatingPointError, match="float16 range"):
        prop.flush_distributed(gather)
    quiet = Proposer("vgg16", hyper_params=dict(hp), weights=bad, precision="f16x3", max_batch=2, overlap_nms=True,
                     check_range=False)
    quiet.propose(x)
    assert quiet.rpn_model.status(reset=True)["f16_range"]
    for precision in ("f16x3", "bf16x3"):
        good = Proposer("vgg16", hyper_params=dict(hp), weights=base if precision == "f16x3" else bad, precision=precision,
                        max_batch=2, overlap_nms=True)
        boxes, scores, valid, idx = good.propose(x)
        assert int(valid.min()) > 0
