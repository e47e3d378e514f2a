"""GPU parity, conv stack: HIP kernels (through the C ABI) against torch-CPU conv (oracle/conv_oracle.py)
on seeded inputs.  Floating point: tolerance 1e-4 absolute on head outputs (north star); single layers are
held to a relative bound derived from the f32 MFMA's ordered-fmaf numerics."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import bbox_oracle as bo
from oracle import c_oracle as co
from oracle import conv_oracle as cv
from tf_rpn_amd import _lib as L
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights

pytestmark = pytest.mark.gpu


def _conv_gpu(x, w, b, stride, pad_t, pad_l, OH, OW, act, precision="f32"):
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    bd = torch.from_numpy(b).cuda() if b is not None else None
    B, H, W, Cin = x.shape
    R, S, _, Cout = w.shape
    out = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
    st = L.lib().rpn_conv2d(L.ptr(xd), B, H, W, Cin, L.ptr(wd), L.ptr(bd), R, S, Cout, stride, pad_t, pad_l, OH, OW,
                            L.ACTS[act], L.PRECISIONS[precision], L.ptr(out), L.stream_ptr())
    L.check(st, "rpn_conv2d")
    torch.cuda.synchronize()
    return out.cpu().numpy()


CONV_CASES = [
    # B, H, W, Cin, Cout, R, stride, (pt, pb, pl, pr), act
    (1, 16, 16, 16, 32, 3, 1, (1, 1, 1, 1), "relu"),          # one exact tile
    (2, 20, 37, 64, 128, 3, 1, (1, 1, 1, 1), "relu"),         # ragged tiles, 128-wide N tile
    (1, 31, 31, 512, 512, 3, 1, (1, 1, 1, 1), "relu"),        # block5 / rpn_conv shape
    (2, 19, 23, 3, 64, 3, 1, (1, 1, 1, 1), "relu"),           # block1_conv1: Cin = 3 (generic gather)
    (1, 40, 40, 3, 32, 3, 2, (0, 1, 0, 1), "relu6"),          # MobileNetV2 stem, even size: pad (0,1)
    (1, 41, 41, 3, 32, 3, 2, (1, 1, 1, 1), "relu6"),          # odd size: pad (1,1)
    (2, 31, 31, 512, 45, 1, 1, (0, 0, 0, 0), "sigmoid"),      # head-sized 1x1, Cout = 45
    (1, 33, 17, 24, 144, 1, 1, (0, 0, 0, 0), "relu6"),        # MNv2 expand: Cin = 24 (partial K slice)
    (1, 33, 17, 144, 24, 1, 1, (0, 0, 0, 0), "linear"),       # MNv2 project: Cout = 24
    (3, 9, 9, 96, 576, 1, 1, (0, 0, 0, 0), "relu6"),
    (1, 8, 16, 20, 36, 3, 1, (1, 1, 1, 1), "linear"),         # Cin % 16 != 0 with 9 taps
    # grids of >= 512 workgroups of 128 x 128: the LDS-DMA staged kernel (conv_igemm_f32_dma)
    (4, 96, 96, 24, 256, 1, 1, (0, 0, 0, 0), "relu6"),        # 1x1, a partial 16-channel slice (quads beyond Cin are zeros)
    (4, 90, 100, 20, 136, 3, 1, (1, 1, 1, 1), "relu"),        # ragged tiles, Cin % 16 != 0 with 9 taps, a ragged N tile
    (4, 191, 191, 32, 256, 3, 2, (0, 1, 0, 1), "linear"),     # stride 2, asymmetric padding
    (2, 125, 125, 128, 256, 3, 1, (1, 1, 1, 1), "relu"),      # block3_conv1's shape
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_conv2d_single_layer(case):
    B, H, W, Cin, Cout, R, stride, pad, act = case
    rng = np.random.RandomState(hash(case[:7]) % (2 ** 31))
    x = rng.uniform(-1, 1, size=(B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((R, R, Cin, Cout)) * np.sqrt(2.0 / (R * R * Cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, size=(Cout,)).astype(np.float32)
    ref = cv.conv2d_nhwc(x, w, b, stride=stride, pad=pad, act=act, dtype=torch.float64)
    OH, OW = ref.shape[1:3]
    got = _conv_gpu(x, w, b, stride, pad[0], pad[2], OH, OW, act)
    assert not np.isnan(got).any(), "some outputs were never written"
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref).max()
    assert err <= 2e-5 * scale, "max abs err %.3e (scale %.2f)" % (err, scale)
    if Cout >= 128 and B >= 2 and H >= 90:      # (these are meant for the DMA-staged kernel: conv_f32_tile_n's rule, restated)
        n_cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert -(-OW // 16) * -(-OH // 8) * B * -(-Cout // 128) >= 2 * n_cus
    if R * R * Cin <= 600 and B * H * W <= 40000:   # cross-check the independent plain-C direct conv on small cases
        cref = co.conv2d(x, w, b, stride=stride, pad_t=pad[0], pad_l=pad[2], out_hw=(OH, OW), act=act)
        assert np.abs(got - cref).max() <= 2e-5 * scale


def test_conv2d_no_bias_matches():
    rng = np.random.RandomState(5)
    x = rng.uniform(-1, 1, size=(1, 12, 12, 32)).astype(np.float32)
    w = rng.standard_normal((3, 3, 32, 16)).astype(np.float32) * 0.1
    ref = cv.conv2d_nhwc(x, w, None, pad=(1, 1, 1, 1), dtype=torch.float64)
    assert np.abs(_conv_gpu(x, w, None, 1, 1, 1, 12, 12, "linear") - ref).max() <= 2e-5


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 15), (1, 5, 16), (1, 4, 17), (2, 9, 64), (1, 8, 65), (1, 2, 66), (3, 7, 130),
                                   (1, 13, 500), (1, 17, 70), (2, 16, 129)], ids=lambda s: "x".join(map(str, s)))
def test_first_layer_float32_matrix_pipe_kernel(shape):
    """conv_cin3_f32_mfma_kernel (VGG16 block1_conv1 of the float32 graphs: Cin 3 -> 64, 'same'): every border case of its 64-pixel row
    segments and 8-row tiles (widths 1 .. 500 around the 16 / 64 steps, heights that leave waves without a row), ReLU and linear,
    with and without a bias, against float64 and against the plain-C direct convolution."""
    B, H, W = shape
    rng = np.random.RandomState(B * 1000 + H * 31 + W)
    x = rng.uniform(-1, 1, size=(B, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64)) * np.sqrt(2.0 / 27)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, size=(64,)).astype(np.float32)
    for act, bias in (("relu", b), ("linear", b), ("linear", None)):
        ref = cv.conv2d_nhwc(x, w, bias, pad=(1, 1, 1, 1), act=act, dtype=torch.float64)
        got = _conv_gpu(x, w, bias, 1, 1, 1, H, W, act)
        assert not np.isnan(got).any(), "some outputs were never written"
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))
        if bias is not None:
            cref = co.conv2d(x, w, bias, stride=1, pad_t=1, pad_l=1, out_hw=(H, W), act=act)
            assert np.abs(got - cref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max()))     # (both are bias-first fma chains in tap order)


def test_first_layer_float32_matrix_pipe_kernel_is_exact_on_integers_at_full_size():
    """Batch 4 of 500 x 500 (the persistent kernel's tile loop, ~4 tiles per workgroup, the chip's store path busy) with small-integer
    images, weights and biases: every product and sum is exact in float32, so the kernel must match torch's convolution BIT FOR BIT,
    twice.  (The first build of this kernel passed every tolerance test of small shapes and stored garbage into 0.006 % of these
    outputs, differently from run to run: a store-data hazard hipcc does not guard -- tests/test_host.py scans for it.)"""
    rng = np.random.RandomState(12)
    B, H, W = 4, 500, 500
    x = rng.randint(-4, 5, size=(B, H, W, 3)).astype(np.float32)
    w = rng.randint(-3, 4, size=(3, 3, 3, 64)).astype(np.float32)
    b = rng.randint(-2, 3, size=(64,)).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b),
                                     padding=1).clamp_(min=0).permute(0, 2, 3, 1).numpy()
    for _ in range(2):
        got = _conv_gpu(x, w, b, 1, 1, 1, H, W, "relu")
        wrong = np.argwhere(got != ref)
        assert len(wrong) == 0, "%d wrong values, first at %s: got %r want %r" % (len(wrong), wrong[0], got[tuple(wrong[0])], ref[tuple(wrong[0])])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("shape", [(8, 125, 125, 256, 256), (4, 250, 250, 64, 128), (8, 31, 31, 512, 512)], ids=lambda s: "x".join(map(str, s)))
def test_full_size_layers_are_exact_on_integers(shape, precision):
    """Race / hazard detector for the persistent conv kernels at the sizes the bench runs (a chip full of workgroups, several tiles per
    workgroup): small-integer inputs, weights and biases make every product and every partial sum an integer below 2^24 -- exact in
    float32 in ANY summation order, and exact in the f16x3 arithmetic too (an integer below 2048 is its own float16 `hi` half, `lo`
    = 0; the power-of-two weight scale is exact) -- so the layer must equal torch's convolution BIT FOR BIT, twice.  (Tolerance tests
    of small shapes did not see the store-data hazard of round 6: tests/test_host.py::test_no_unguarded_store_data_hazard.)"""
    B, H, W, Cin, Cout = shape
    rng = np.random.RandomState(Cin + Cout + H)
    x = rng.randint(-3, 4, size=(B, H, W, Cin)).astype(np.float32)
    w = rng.randint(-2, 3, size=(3, 3, Cin, Cout)).astype(np.float32)
    b = rng.randint(-5, 6, size=(Cout,)).astype(np.float32)
    assert 9 * Cin * 3 * 2 + 5 < 2 ** 24
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b),
                                     padding=1).clamp_(min=0).permute(0, 2, 3, 1).contiguous().numpy()
    for _ in range(2):
        got = _conv_gpu(x, w, b, 1, 1, 1, H, W, "relu", precision=precision)
        wrong = np.argwhere(got != ref)
        assert len(wrong) == 0, "%d wrong values, first at %s: got %r want %r" % (len(wrong), wrong[0], got[tuple(wrong[0])], ref[tuple(wrong[0])])


def test_maxpool_and_depthwise():
    rng = np.random.RandomState(6)
    x = rng.uniform(-1, 1, size=(2, 125, 37, 64)).astype(np.float32)            # odd sizes: 'valid' floors
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((2, 62, 18, 64), device="cuda")
    L.check(L.lib().rpn_maxpool2x2(L.ptr(xd), 2, 125, 37, 64, L.ptr(out), L.stream_ptr()), "rpn_maxpool2x2")
    assert np.array_equal(out.cpu().numpy(), co.maxpool2x2(x))
    for (H, stride, pad) in ((21, 1, (1, 1, 1, 1)), (20, 2, (0, 1, 0, 1)), (21, 2, (1, 1, 1, 1))):
        x = rng.uniform(-1, 1, size=(2, H, H, 96)).astype(np.float32)
        w = rng.standard_normal((3, 3, 96, 1)).astype(np.float32)
        b = rng.uniform(-1, 1, size=(96,)).astype(np.float32)
        ref = cv.conv2d_nhwc(x, w, b, stride=stride, pad=pad, act="relu6", depthwise=True, dtype=torch.float64)
        OH = ref.shape[1]
        xd, wd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(w.reshape(3, 3, 96)).cuda(), torch.from_numpy(b).cuda()
        out = torch.empty((2, OH, OH, 96), device="cuda")
        L.check(L.lib().rpn_dwconv3x3(L.ptr(xd), 2, H, H, 96, L.ptr(wd), L.ptr(bd), stride, pad[0], pad[2], OH, OH,
                                      L.ACTS["relu6"], L.ptr(out), L.stream_ptr()), "rpn_dwconv3x3")
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5


# ---- whole models ---------------------------------------------------------------------------------
def _model_case(backbone, img, B, seed=1, fm=None):
    hp = bo.get_hyper_params(backbone, img_size=img, feature_map_shape=fm)
    weights = synthetic_weights(backbone, hp, seed=seed)
    model = RPNModel(backbone, hp, max_batch=B, keep_activations=True)
    model.set_weights(weights)
    imgs = np.random.RandomState(0).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    return hp, weights, model, imgs


@pytest.mark.parametrize("backbone,img,B", [("vgg16", 64, 2), ("vgg16", 150, 1), ("mobilenet_v2", 64, 2),
                                             ("mobilenet_v2", 150, 3), ("mobilenet_v2", 125, 1)])
def test_model_forward_small(backbone, img, B):
    hp, weights, model, imgs = _model_case(backbone, img, B)
    reg, cls = model.predict_on_batch(imgs)
    ref64 = cv.rpn_forward(backbone, imgs, weights, dtype=torch.float64, return_features=True)
    ref32 = cv.rpn_forward(backbone, imgs, weights, dtype=torch.float32)
    F = model.feature_map_shape
    assert reg.shape == (B, F, F, 36) and cls.shape == (B, F, F, 9) and ref64[0].shape == reg.shape
    feat = model.get_activation(model.tap_layer, batch=B).cpu().numpy()
    fscale = max(1.0, float(np.abs(ref64[2]).max()))
    assert np.abs(feat - ref64[2]).max() <= 5e-5 * fscale
    assert np.abs(reg - ref64[0]).max() <= 1e-4 and np.abs(cls - ref64[1]).max() <= 1e-4
    # no further from the float64 reference than torch's own float32 path is (x4 slack)
    assert np.abs(reg - ref64[0]).max() <= 4 * np.abs(ref32[0] - ref64[0]).max() + 1e-6


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("img,B", [(64, 2), (125, 1), (150, 3), (224, 2), (500, 1)])
def test_mobilenet_v2_fused_blocks(img, B, precision):
    """The production MobileNetV2 graph: stem + expanded_conv and every inverted-residual block run as ONE launch each
    (mnv2_block_kernels.hip; <= 16 launches per forward).  Held to the float64 oracle (1e-4) and to the layer-by-layer
    path (keep_activations=True: the unfused kernels) on the same weights -- odd sizes exercise the stride-2
    correct_pad cases (125 -> 63 -> 32) and ragged 4 x 8 tiles."""
    hp = bo.get_hyper_params("mobilenet_v2", img_size=img, feature_map_shape=None)
    weights = synthetic_weights("mobilenet_v2", hp, seed=7)
    imgs = np.random.RandomState(3).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    fused = RPNModel("mobilenet_v2", hp, precision=precision, max_batch=B)
    fused.set_weights(weights)
    kernels = [op["kernel"] for op in fused.ops()]
    assert sum(k.startswith("ir_block") for k in kernels) == 13 and len(kernels) <= 16, kernels
    reg, cls = fused.predict_on_batch(imgs)
    ref = cv.rpn_forward("mobilenet_v2", imgs, weights, dtype=torch.float64, return_features=True)
    feat = fused.get_activation(fused.tap_layer, batch=B).cpu().numpy()
    fscale = max(1.0, float(np.abs(ref[2]).max()))
    assert np.abs(feat - ref[2]).max() <= 5e-5 * fscale
    assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4
    layerwise = RPNModel("mobilenet_v2", hp, precision=precision, max_batch=B, keep_activations=True)
    layerwise.set_weights(weights)
    assert not any(op["kernel"].startswith("ir_block") for op in layerwise.ops())
    reg_u, cls_u = layerwise.predict_on_batch(imgs)
    assert np.abs(reg - reg_u).max() <= 5e-5 and np.abs(cls - cls_u).max() <= 5e-5
    # batch invariance of the fused path: image 0 alone, bit for bit
    r1, c1 = fused.predict_on_batch(imgs[:1])
    assert np.array_equal(r1[0], reg[0]) and np.array_equal(c1[0], cls[0])


def test_model_layerwise_vgg16():
    """Every intermediate activation against the float64 oracle (catches an O(1)-wrong layer that a
    whole-model tolerance could hide)."""
    hp, weights, model, imgs = _model_case("vgg16", 96, 1)
    model.predict_on_batch(imgs)
    x = torch.from_numpy(imgs).to(torch.float64).permute(0, 3, 1, 2)
    import torch.nn.functional as Fnn
    for layer in cv.VGG16_LAYERS:
        if layer == "pool":
            x = Fnn.max_pool2d(x, 2, 2)
            continue
        name = layer[0]
        x = torch.relu(cv._conv(x, weights[name]["kernel"], weights[name]["bias"], padding=1, dtype=torch.float64))
        got = model.get_activation(name, batch=1).cpu().numpy()
        ref = x.permute(0, 2, 3, 1).numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 3e-5 * scale, name


@pytest.mark.parametrize("backbone", ["vgg16", "mobilenet_v2"])
def test_model_forward_full_size_one_image(backbone):
    """The reference's own input size (500x500), one image, against torch-CPU float32."""
    hp, weights, model, imgs = _model_case(backbone, 500, 1)
    reg, cls = model.predict_on_batch(imgs)
    ref = cv.rpn_forward(backbone, imgs, weights, dtype=torch.float32)
    assert reg.shape == ref[0].shape == (1, hp["feature_map_shape"], hp["feature_map_shape"], 36)
    assert np.abs(reg - ref[0]).max() <= 1e-4, np.abs(reg - ref[0]).max()
    assert np.abs(cls - ref[1]).max() <= 1e-4, np.abs(cls - ref[1]).max()
    assert 0.0 < cls.min() and cls.max() < 1.0


@pytest.mark.parametrize("img,B", [(96, 2), (150, 1), (203, 3)])
def test_f32_fused_pool_equals_layerwise(img, B):
    """Exact-float32 VGG16 graph: block*_pool runs inside the preceding conv's epilogue (the 2 x 2 window sits in one lane's
    accumulators; ConvArgs::pool) unless every activation is kept.  Same bits as conv -> HBM -> pool kernel, odd edges
    included (150 -> 75 -> 37 -> 18 -> 9: every pool floors), and within 1e-4 of the float64 oracle."""
    hp = bo.get_hyper_params("vgg16", img_size=img, feature_map_shape=None)
    weights = synthetic_weights("vgg16", hp, seed=4)
    imgs = np.random.RandomState(5).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    outs = []
    for keep in (False, True):
        model = RPNModel("vgg16", hp, precision="f32", max_batch=B, keep_activations=keep)
        model.set_weights(weights)
        kernels = [op["kernel"] for op in model.ops()]
        assert ("fused:maxpool_f32" in kernels) == (not keep)
        outs.append(model.predict_on_batch(imgs))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    assert np.abs(outs[0][0] - ref[0]).max() <= 1e-4 and np.abs(outs[0][1] - ref[1]).max() <= 1e-4


def test_model_batch_invariance_full_size():
    """BASELINE config C2 (B=8, VGG16, 500x500): image i alone gives bit-identical outputs to image i in
    the batch (no cross-image term anywhere on the path)."""
    hp = bo.get_hyper_params("vgg16")
    model = RPNModel("vgg16", hp, max_batch=8)
    model.set_weights(synthetic_weights("vgg16", hp, seed=1))
    imgs = torch.rand((8, 500, 500, 3), generator=torch.Generator().manual_seed(0)).cuda()
    reg, cls = model.predict_on_batch(imgs)
    reg, cls = reg.clone(), cls.clone()
    for i in (0, 5):
        r1, c1 = model.predict_on_batch(imgs[i:i + 1].contiguous())
        assert torch.equal(r1[0], reg[i]) and torch.equal(c1[0], cls[i])
    assert torch.isfinite(reg).all() and torch.isfinite(cls).all()


@pytest.mark.parametrize("img,fm", [(500, 32), (1024, 64)])
def test_mobilenet_v2_batch_invariance_across_ksplit_factors(img, fm):
    """MobileNetV2, f16x3: the fused blocks on small grids split their expanded channels over 3 or 6 workgroups per tile
    (one 500 x 500 image: 32 tiles -> 6; one 1024 x 1024 image: 128 -> 3), large grids do not.  The projection tree is the
    same for every factor, so image i alone (split) must give BIT-identical outputs to image i inside a batch (not split,
    or split by another factor)."""
    hp = bo.get_hyper_params("mobilenet_v2", img_size=img, feature_map_shape=fm)
    B = 8 if img == 500 else 3
    model = RPNModel("mobilenet_v2", hp, precision="f16x3", max_batch=B)
    model.set_weights(synthetic_weights("mobilenet_v2", hp, seed=1))
    imgs = torch.rand((B, img, img, 3), generator=torch.Generator().manual_seed(0)).cuda()
    reg, cls = model.predict_on_batch(imgs)
    reg, cls = reg.clone(), cls.clone()
    for i in (0, B - 1):
        r1, c1 = model.predict_on_batch(imgs[i:i + 1].contiguous())
        assert torch.equal(r1[0], reg[i]) and torch.equal(c1[0], cls[i])
        r2, c2 = model.predict_on_batch(imgs[i:i + 1].contiguous())          # and run to run (the tickets are left at zero)
        assert torch.equal(r2[0], r1[0]) and torch.equal(c2[0], c1[0])
    if B >= 3:
        r3, c3 = model.predict_on_batch(imgs[:2].contiguous())                # another grid size, maybe another factor
        assert torch.equal(r3, reg[:2]) and torch.equal(c3, cls[:2])
    assert torch.isfinite(reg).all() and not model.status()["f16_range"]
    ref = cv.rpn_forward("mobilenet_v2", imgs[:1].cpu().numpy(), synthetic_weights("mobilenet_v2", hp, seed=1), dtype=torch.float64)
    assert np.abs(reg[:1].cpu().numpy() - ref[0]).max() <= 1e-4 and np.abs(cls[:1].cpu().numpy() - ref[1]).max() <= 1e-4


def test_mobilenet_v2_handles_of_different_max_batch_agree_within_the_bound():
    """Kernel choices that change the summation order are made ONCE PER HANDLE, from the grid at its max_batch (block 3's
    chunk size, blocks 4-5 on the 256-thread or the 512-thread block kernel): every batch size of one handle gives the same
    bits (test above), two handles of different max_batch agree within the float bound, not necessarily bit for bit -- the
    contract include/rpn_hip.h states.  Both stay within 1e-4 of the float64 oracle."""
    hp = bo.get_hyper_params("mobilenet_v2")
    weights = synthetic_weights("mobilenet_v2", hp, seed=2)
    imgs = torch.rand((1, 500, 500, 3), generator=torch.Generator().manual_seed(3)).cuda()
    outs = []
    for mb in (1, 8):
        model = RPNModel("mobilenet_v2", hp, precision="f16x3", max_batch=mb)
        model.set_weights(weights)
        reg, cls = model.predict_on_batch(imgs)
        outs.append((reg.clone(), cls.clone()))
        kernels = {op["name"]: op["kernel"] for op in model.ops()}
        assert ("hr_f16x3" in kernels["block_4_project"]) == (mb == 8), kernels["block_4_project"]
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2e-5 and (outs[0][1] - outs[1][1]).abs().max().item() <= 2e-5
    ref = cv.rpn_forward("mobilenet_v2", imgs.cpu().numpy(), weights, dtype=torch.float64)
    for reg, cls in outs:
        assert np.abs(reg.cpu().numpy() - ref[0]).max() <= 1e-4 and np.abs(cls.cpu().numpy() - ref[1]).max() <= 1e-4


@pytest.mark.parametrize("backbone,precision", [("mobilenet_v2", "f16x3"), ("vgg16", "f16x3"), ("vgg16", "bf16x3")])
def test_rpn_conv_k_tree_same_bits_at_every_split_factor(backbone, precision):
    """rpn_conv in the split-precision modes is a K TREE: four fixed leaves of K, value (l0 + l1) + (l2 + l3).  One image
    (64 workgroups of 4 x 32 x 64: four per tile, one leaf each, the head adds four slabs), three images (two per tile: l0 + l1
    and l2 + l3 as slabs) and eight (the persistent 64-wide kernel walks all of K and folds the leaves in its tile loop) must
    give BIT-identical outputs for the same image; a model that keeps every activation never splits (tolerance: its
    upstream layers run on other kernels)."""
    hp = bo.get_hyper_params(backbone)
    weights = synthetic_weights(backbone, hp, seed=3)
    model = RPNModel(backbone, hp, precision=precision, max_batch=8)
    model.set_weights(weights)
    imgs = torch.rand((8, 500, 500, 3), generator=torch.Generator().manual_seed(5)).cuda()
    reg, cls = model.predict_on_batch(imgs)
    reg, cls = reg.clone(), cls.clone()
    for nb in (1, 3, 4, 5):
        r, c = model.predict_on_batch(imgs[:nb].contiguous())
        assert torch.equal(r, reg[:nb]) and torch.equal(c, cls[:nb]), nb
    r, c = model.predict_on_batch(imgs[7:8].contiguous())
    assert torch.equal(r[0], reg[7]) and torch.equal(c[0], cls[7])
    keep = RPNModel(backbone, hp, precision=precision, max_batch=2, keep_activations=True)   # (layer by layer: other kernels
    keep.set_weights(weights)                                                                # upstream, so not the same bits)
    rk, ck = keep.predict_on_batch(imgs[:2].contiguous())
    assert (rk - reg[:2]).abs().max().item() <= 2e-5 and (ck - cls[:2]).abs().max().item() <= 2e-5
    ref = cv.rpn_forward(backbone, imgs[:1].cpu().numpy(), weights, dtype=torch.float64)
    assert np.abs(reg[:1].cpu().numpy() - ref[0]).max() <= 1e-4 and np.abs(cls[:1].cpu().numpy() - ref[1]).max() <= 1e-4


@pytest.mark.gpu
def test_rpn_conv_one_chain_subprocess():
    """RPN_KSPLIT=0 (read once per process): rpn_conv as ONE accumulation chain at every batch size, as before the K tree
    (one workgroup per tile walks all of K at batch 1).  Same tolerance against the oracle, same batch invariance."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from oracle import bbox_oracle as bo
        from oracle import conv_oracle as cv
        from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights
        hp = bo.get_hyper_params("mobilenet_v2")
        w = synthetic_weights("mobilenet_v2", hp, seed=3)
        m = RPNModel("mobilenet_v2", hp, precision="f16x3", max_batch=4); m.set_weights(w)
        imgs = torch.rand((4, 500, 500, 3), generator=torch.Generator().manual_seed(5)).cuda()
        reg, cls = m.predict_on_batch(imgs); reg, cls = reg.clone(), cls.clone()
        r1, c1 = m.predict_on_batch(imgs[:1].contiguous())
        assert torch.equal(r1[0], reg[0]) and torch.equal(c1[0], cls[0])
        ref = cv.rpn_forward("mobilenet_v2", imgs[:1].cpu().numpy(), w, dtype=torch.float64)
        assert np.abs(reg[:1].cpu().numpy() - ref[0]).max() <= 1e-4 and np.abs(cls[:1].cpu().numpy() - ref[1]).max() <= 1e-4
        print("one chain ok")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RPN_KSPLIT="0"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "one chain ok" in r.stdout


def test_forward_requires_all_weights():
    hp = bo.get_hyper_params("vgg16", img_size=64, feature_map_shape=4)
    model = RPNModel("vgg16", hp, max_batch=1)
    with pytest.raises(ValueError, match="never set"):
        model.predict_on_batch(np.zeros((1, 64, 64, 3), np.float32))
    with pytest.raises(ValueError):
        model.predict_on_batch(np.zeros((1, 32, 64, 3), np.float32))


# ---- x3-split 16-bit MFMA path (bf16x3 / f16x3) -------------------------------------------------------
SPLIT_CASES = [
    # B, H, W, Cin, Cout, act
    (1, 8, 32, 16, 64, "relu"),            # one exact 8x32 tile, one K slice, 64-wide N tile
    (2, 21, 45, 64, 64, "relu"),           # ragged tiles (block1_conv2 shape class)
    (1, 37, 70, 64, 128, "relu"),          # 128-wide N tile
    (2, 31, 31, 512, 512, "relu"),         # block5 / rpn_conv: short tiles (TH = 4)
    (8, 62, 62, 256, 512, "relu"),         # block4_conv1: tall tiles, 512 blocks
    (1, 13, 9, 32, 48, "linear"),          # Cout = 48 (cout_pad 64), signed outputs
    (1, 40, 33, 128, 256, "relu6"),
    # persistent LDS-DMA kernel (even number of 32-channel slices)
    (8, 125, 125, 128, 128, "relu"),       # 512 tiles: every workgroup walks two tiles (tap stream crosses a tile boundary)
    (3, 50, 45, 128, 384, "relu"),         # 3 n-tiles on a 2 x 4 XCD ownership grid: padding slots are skipped
    (1, 9, 40, 64, 192, "linear"),         # Cin = 64 (two slices per tile), ragged rows / columns, signed outputs
    (2, 31, 31, 256, 320, "relu"),         # 64-wide tiles (small map), Cout not a multiple of 128
    (1, 16, 64, 192, 128, "relu"),         # exact tiles, 6 slices
    # register-staged 16x16x32 kernels (an odd number of 32-channel slices)
    (2, 20, 40, 160, 128, "relu"),         # 5 slices, 4 x 32 px x 64 ch tiles (small grid)
    (8, 70, 70, 160, 256, "relu"),         # 5 slices, 8 x 32 px x 128 ch tiles (>= 256 workgroups)
    (1, 31, 31, 288, 192, "linear"),       # 9 slices, signed outputs
]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_conv3x3_split_single_layer(case, precision):
    B, H, W, Cin, Cout, act = case
    rng = np.random.RandomState(abs(hash(case[:5])) % (2 ** 31))
    x = rng.uniform(-1, 1, size=(B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, size=(Cout,)).astype(np.float32)
    ref = cv.conv2d_nhwc(x, w, b, pad=(1, 1, 1, 1), act=act, dtype=torch.float64)
    got = _conv_gpu(x, w, b, 1, 1, 1, H, W, act, precision=precision)
    assert not np.isnan(got).any(), "some outputs were never written"
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref).max()
    bound = 1e-4 if precision == "bf16x3" else 1e-5          # observed: ~1e-5 / ~1e-6 (product error 2^-16 / 2^-21)
    assert err <= bound * scale, "%s max abs err %.3e (scale %.2f)" % (precision, err, scale)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("backbone,img,B", [("vgg16", 96, 2), ("vgg16", 150, 1), ("mobilenet_v2", 96, 2)])
def test_model_forward_split_small(backbone, img, B, precision):
    hp = bo.get_hyper_params(backbone, img_size=img, feature_map_shape=None)
    weights = synthetic_weights(backbone, hp, seed=1)
    model = RPNModel(backbone, hp, precision=precision, max_batch=B, keep_activations=True)
    model.set_weights(weights)
    imgs = np.random.RandomState(0).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    reg, cls = model.predict_on_batch(imgs)
    ref = cv.rpn_forward(backbone, imgs, weights, dtype=torch.float64, return_features=True)
    feat = model.get_activation(model.tap_layer, batch=B).cpu().numpy()
    fscale = max(1.0, float(np.abs(ref[2]).max()))
    assert np.abs(feat - ref[2]).max() <= 2e-4 * fscale
    assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4      # the north star's bound


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("img", [150, 200])
def test_model_forward_split_fused_pool(img, precision):
    """Without keep_activations the 2x2 max-pools run inside the conv epilogue; odd sizes (150 -> 75 -> 37 ->
    18 -> 9) exercise the 'valid' floor at every level."""
    hp = bo.get_hyper_params("vgg16", img_size=img)
    weights = synthetic_weights("vgg16", hp, seed=4)
    model = RPNModel("vgg16", hp, precision=precision, max_batch=2)
    model.set_weights(weights)
    imgs = np.random.RandomState(5).uniform(0, 1, size=(2, img, img, 3)).astype(np.float32)
    reg, cls = model.predict_on_batch(imgs)
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    assert reg.shape == ref[0].shape
    assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_model_forward_split_full_size(precision):
    """VGG16 at the reference's 500x500: head outputs within 1e-4 of torch-CPU float32 AND of the exact-f32
    HIP path; the measured errors are printed (-s) and recorded in DESIGN.md."""
    hp = bo.get_hyper_params("vgg16")
    weights = synthetic_weights("vgg16", hp, seed=1)
    imgs = np.random.RandomState(0).uniform(0, 1, size=(1, 500, 500, 3)).astype(np.float32)
    outs = {}
    for prec in ("f32", precision):
        model = RPNModel("vgg16", hp, precision=prec, max_batch=1)
        model.set_weights(weights)
        outs[prec] = model.predict_on_batch(imgs)
        del model
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    for name, i in (("reg", 0), ("cls", 1)):
        e_split = np.abs(outs[precision][i] - ref[i]).max()
        e_f32 = np.abs(outs["f32"][i] - ref[i]).max()
        print("%s %s: max|err| vs float64 oracle: split %.3e, exact-f32 kernel %.3e" % (precision, name, e_split, e_f32))
        assert e_split <= 1e-4
        assert np.abs(outs[precision][i] - outs["f32"][i]).max() <= 1e-4


@pytest.mark.parametrize("img,B", [(96, 2), (150, 1), (203, 2), (500, 1)])
def test_vgg16_block1_one_launch(img, B):
    """f16x3 runs VGG16 block 1 (block1_conv1 -> block1_conv2 -> block1_pool) as ONE launch: the first layer is
    computed into the halo tile of the second, the 64-channel full-resolution tensors never reach HBM.  Same results as
    the layer-by-layer graph (keep_activations=True: one kernel per layer) and as the float64 oracle; image sizes that
    are not multiples of the 8 x 32 tile exercise the zero halo at the right / bottom edge; image 0 alone gives the same
    bits (no cross-image coupling)."""
    hp = bo.get_hyper_params("vgg16", img_size=img, feature_map_shape=None)
    weights = synthetic_weights("vgg16", hp, seed=6)
    for w in weights.values():                                  # non-zero biases: the halo outside the image must be 0,
        if "bias" in w:                                         # not relu(bias)
            w["bias"] = np.random.RandomState(len(w["bias"])).uniform(0.05, 0.2, size=w["bias"].shape).astype(np.float32)
    imgs = np.random.RandomState(7).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
    fused = RPNModel("vgg16", hp, precision="f16x3", max_batch=B)
    fused.set_weights(weights)
    kernels = [op["kernel"] for op in fused.ops()]
    assert kernels[0] == "vgg_block1<f16x3>" and not any(k.startswith("conv_cin3") for k in kernels)
    reg, cls = fused.predict_on_batch(imgs)
    feat = fused.get_activation(fused.tap_layer, batch=B).cpu().numpy()
    layerwise = RPNModel("vgg16", hp, precision="f16x3", max_batch=B, keep_activations=True)
    layerwise.set_weights(weights)
    assert layerwise.ops()[0]["kernel"].startswith("conv_cin3")
    reg_l, cls_l = layerwise.predict_on_batch(imgs)
    feat_l = layerwise.get_activation(layerwise.tap_layer, batch=B).cpu().numpy()
    fscale = max(1.0, float(np.abs(feat_l).max()))
    assert np.abs(feat - feat_l).max() <= 2e-5 * fscale
    assert np.abs(reg - reg_l).max() <= 2e-5 and np.abs(cls - cls_l).max() <= 2e-5
    ref = cv.rpn_forward("vgg16", imgs, weights, dtype=torch.float64)
    assert np.abs(reg - ref[0]).max() <= 1e-4 and np.abs(cls - ref[1]).max() <= 1e-4
    assert not fused.status()["f16_range"]
    r1, c1 = fused.predict_on_batch(imgs[:1])
    assert np.array_equal(r1[0], reg[0]) and np.array_equal(c1[0], cls[0])


def test_vgg16_block1_one_launch_flags_f16_overflow():
    """block1_conv1's activations pushed past 65504 inside the fused launch (they exist only in LDS there): the device
    flag is raised all the same."""
    hp = bo.get_hyper_params("vgg16", img_size=96, feature_map_shape=6)
    base = synthetic_weights("vgg16", hp, seed=1)
    for w in base.values():
        if "bias" in w:
            w["bias"] = np.full_like(w["bias"], 0.1)
    # ReLU is positively homogeneous: conv1 (kernel, bias) x s and conv2's kernel / s change only conv1's output
    weights = {k: dict(v) for k, v in base.items()}
    s = np.float32(2.0 ** 20)
    weights["block1_conv1"]["kernel"] = base["block1_conv1"]["kernel"] * s
    weights["block1_conv1"]["bias"] = base["block1_conv1"]["bias"] * s
    weights["block1_conv2"]["kernel"] = base["block1_conv2"]["kernel"] / s
    x = torch.from_numpy(np.random.RandomState(0).uniform(0, 1, size=(1, 96, 96, 3)).astype(np.float32)).cuda()
    model = RPNModel("vgg16", hp, precision="f16x3", max_batch=1)
    model.set_weights(weights)
    assert model.ops()[0]["kernel"] == "vgg_block1<f16x3>"
    model.predict_on_batch(x)
    assert model.status(reset=True)["f16_range"]
    model.set_weights(base)
    model.predict_on_batch(x)
    assert not model.status()["f16_range"]


def test_split_round_trip_and_pool():
    """SPLIT16 carries hi + lo: float32 -> split -> float32 is within 2^-16 (bf16) / 2^-21 (f16) relative,
    and pooling in split form equals pooling the joined values."""
    hp = bo.get_hyper_params("vgg16", img_size=64)
    weights = synthetic_weights("vgg16", hp, seed=2)
    imgs = np.random.RandomState(3).uniform(0, 1, size=(1, 64, 64, 3)).astype(np.float32)
    for precision, rel in (("bf16x3", 2.0 ** -15), ("f16x3", 2.0 ** -20)):
        model = RPNModel("vgg16", hp, precision=precision, max_batch=1, keep_activations=True)
        model.set_weights(weights)
        model.predict_on_batch(imgs)
        exact = RPNModel("vgg16", hp, precision="f32", max_batch=1, keep_activations=True)
        exact.set_weights(weights)
        exact.predict_on_batch(imgs)
        a = exact.get_activation("block1_conv1", batch=1).cpu().numpy()              # float32 NHWC
        s = model.get_activation("block1_conv1", batch=1).cpu().numpy()              # SPLIT16 written directly, joined
        assert np.abs(s - a).max() <= rel * max(1.0, np.abs(a).max())
        c2 = model.get_activation("block1_conv2", batch=1).cpu().numpy()
        p = model.get_activation("block1_pool", batch=1).cpu().numpy()
        assert np.array_equal(p, co.maxpool2x2(c2))


@pytest.mark.gpu
def test_split_conv_dynamic_tile_schedule_subprocess():
    """RPN_S16_DYN=1 (tile queue of the persistent kernel) and RPN_S16_C64=1 (64 -> 64 layers on it) are read once per
    process: run the persistent-kernel cases in a child process with both set.  Same results required (the schedule
    only changes which workgroup computes which tile); two launches per case check that a launch re-arms the queue."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from oracle import conv_oracle as cv
        from tests.test_gpu_conv import _conv_gpu
        worst = 0.0
        for (B, H, W, Cin, Cout, act) in [(8, 125, 125, 128, 128, "relu"), (3, 50, 45, 128, 384, "relu"),
                                          (2, 31, 31, 256, 320, "relu"), (2, 21, 45, 64, 64, "relu"), (1, 9, 40, 64, 192, "linear")]:
            rng = np.random.RandomState(B * 1000 + H)
            x = rng.uniform(-1, 1, size=(B, H, W, Cin)).astype(np.float32)
            w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
            b = rng.uniform(-0.5, 0.5, size=(Cout,)).astype(np.float32)
            ref = cv.conv2d_nhwc(x, w, b, pad=(1, 1, 1, 1), act=act, dtype=torch.float64)
            for rep in range(2):
                got = _conv_gpu(x, w, b, 1, 1, 1, H, W, act, precision="f16x3")
                assert not np.isnan(got).any(), "some outputs were never written"
                worst = max(worst, float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())))
        assert worst <= 1e-5, worst
        print("dynamic schedule ok, worst scaled error %%.2e" %% worst)
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, RPN_S16_DYN="1", RPN_S16_C64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "dynamic schedule ok" in r.stdout


@pytest.mark.gpu
def test_small_batch_split_k_subprocess():
    """RPN_KSPLIT=1 (read once per process): at small batches rpn_conv is cut 2 or 4 ways along K, its raw partial sums
    land in slabs and the head adds them (+ rpn_conv's bias, ReLU) while loading its operand.  Same 1e-4 bound against
    the float64 oracle for both backbones and both split factors; the default path (one accumulation chain, bit-identical
    across batch sizes) must differ from it by rounding only; a batch that is too large for the split runs unsplit on the
    same handle."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from oracle import bbox_oracle as bo, conv_oracle as cv
        from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights
        for backbone, img, B in [("vgg16", 500, 1), ("vgg16", 500, 3), ("mobilenet_v2", 500, 2), ("mobilenet_v2", 1024, 1),
                                 ("vgg16", 96, 2)]:
            hp = bo.get_hyper_params(backbone, img_size=img, feature_map_shape=None)
            w = synthetic_weights(backbone, hp, seed=3)
            for v in w.values():
                if "bias" in v:
                    v["bias"] = np.random.RandomState(len(v["bias"])).uniform(-0.1, 0.1, size=v["bias"].shape).astype(np.float32)
            imgs = np.random.RandomState(B).uniform(0, 1, size=(B, img, img, 3)).astype(np.float32)
            m = RPNModel(backbone, hp, precision="f16x3", max_batch=max(B, 4))
            m.set_weights(w)
            reg, cls = m.predict_on_batch(imgs)
            ref = cv.rpn_forward(backbone, imgs, w, dtype=torch.float64)
            e = max(np.abs(reg - ref[0]).max(), np.abs(cls - ref[1]).max())
            assert e <= 1e-4, (backbone, img, B, e)
            r1, c1 = m.predict_on_batch(imgs[:1])                 # another split factor (or none) on the same handle
            assert np.abs(r1[0] - reg[0]).max() <= 2e-6 and np.abs(c1[0] - cls[0]).max() <= 2e-6
            print("ksplit ok", backbone, img, B, "%%.2e" %% e)
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, RPN_KSPLIT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ksplit ok") == 5


@pytest.mark.gpu
def test_persistent_conv_repeats_bit_identically():
    """Race screen of the LDS-DMA pipeline (scripts/race_screen.py is the long version): a DMA that lands after its
    reader shows up as a rarely different tile, so the same launch repeated -- beside a second stream that perturbs the
    timing -- must give bit-identical outputs."""
    side = torch.cuda.Stream()
    junk = torch.rand((16, 1024, 1024), device="cuda")
    for (B, H, Cin, Cout, prec) in [(8, 125, 128, 128, "f16x3"), (2, 31, 512, 512, "bf16x3")]:
        torch.manual_seed(H)
        x = (torch.rand((B, H, H, Cin), device="cuda") - 0.5).contiguous()
        w = (torch.randn((3, 3, Cin, Cout), device="cuda") * (2.0 / (9 * Cin)) ** 0.5).contiguous()
        b = torch.rand((Cout,), device="cuda") - 0.5
        out = torch.empty((B, H, H, Cout), device="cuda")
        ref = None
        for i in range(60):
            if i % 3 == 1:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)
            out.fill_(float("nan"))
            L.check(L.lib().rpn_conv2d(L.ptr(x), B, H, H, Cin, L.ptr(w), L.ptr(b), 3, 3, Cout, 1, 1, 1, H, H, L.ACTS["relu"],
                                       L.PRECISIONS[prec], L.ptr(out), L.stream_ptr()), "rpn_conv2d")
            if ref is None:
                ref = out.clone()
                assert not torch.isnan(ref).any()
            else:
                assert torch.equal(out, ref), "run %d differs from run 0" % i
    torch.cuda.synchronize()


H5PY_PYTHON = "/opt/conda/bin/python3.9"          # the image's interpreter that has the real h5py


@pytest.mark.gpu
@pytest.mark.parametrize("backbone,strings", [("vgg16", "fixed"), ("mobilenet_v2", "vlen")])
def test_load_weights_from_keras_h5(tmp_path, backbone, strings):
    """The reference's ``rpn_model.load_weights(path, by_name=True)`` (predictor.py:43-44) on a Keras-layout .h5 file
    written by the real h5py (full-size VGG16 / MobileNetV2 + RPN weight set, BatchNorm and depthwise layers included):
    the model loaded from the file must produce bit-identical outputs to the model given the same arrays directly."""
    import os, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        ok = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except OSError:
        ok = False
    if not ok:
        pytest.skip("no interpreter with h5py on this machine to write the file")
    hp, weights, _layerwise, imgs = _model_case(backbone, 64, 2)
    model = RPNModel(backbone, hp, max_batch=2)            # the same (production) graph as `loaded` below
    model.set_weights(weights)
    npz, h5 = str(tmp_path / "w.npz"), str(tmp_path / "w.h5")
    RPNModel.save_weights(weights, npz)
    r = subprocess.run([H5PY_PYTHON, os.path.join(here, "golden", "npz_to_keras_h5.py"), npz, h5, strings],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref_reg, ref_cls = model.predict_on_batch(imgs)
    loaded = RPNModel(backbone, hp, max_batch=2)
    done = loaded.load_weights(h5, by_name=True)
    assert sorted(done) == sorted(l["name"] for l in loaded.layers)
    reg, cls = loaded.predict_on_batch(imgs)
    np.testing.assert_array_equal(reg, ref_reg)
    np.testing.assert_array_equal(cls, ref_cls)
    # by_name: a file with only some of the layers sets just those
    part = {k: v for k, v in weights.items() if k.startswith("rpn_")}
    RPNModel.save_weights(part, npz)
    assert subprocess.run([H5PY_PYTHON, os.path.join(here, "golden", "npz_to_keras_h5.py"), npz, h5, strings],
                          capture_output=True, timeout=600).returncode == 0
    fresh = RPNModel(backbone, hp, max_batch=2)
    assert sorted(fresh.load_weights(h5)) == ["rpn_cls", "rpn_conv", "rpn_reg"]


@pytest.mark.gpu
def test_profiling_mask_and_rotation():
    """Per-op HIP-event timing: a mask restricts it to some ops; with rotation each forward times one marked op (round
    robin) and every marked op still gets a duration close to the one measured with all of them timed."""
    hp, weights, model, imgs = _model_case("vgg16", 160, 2)
    x = torch.from_numpy(imgs).cuda()
    ops = model.ops()
    marked = [i for i, op in enumerate(ops) if "conv3x3" in op["kernel"]][:4]
    mask = [i in marked for i in range(len(ops))]
    model.set_profiling_mask(mask)
    model.set_profiling(12)
    for _ in range(12):
        model.predict_on_batch(x)
    full, kept = model.profile_ms()
    assert kept == 12
    assert all((full[i] > 0) == mask[i] for i in range(len(ops)))
    model.set_profiling_rotate(True)
    model.set_profiling(12)
    for _ in range(12):
        model.predict_on_batch(x)
    rot, kept = model.profile_ms()
    assert kept == 12
    for i in range(len(ops)):
        if mask[i]:
            assert rot[i] > 0 and abs(rot[i] - full[i]) <= 0.5 * full[i] + 0.01, (ops[i]["name"], rot[i], full[i])
        else:
            assert rot[i] == 0
    model.set_profiling_rotate(False)
    model.set_profiling_mask(None)
    model.set_profiling(0)
