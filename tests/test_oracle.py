"""CPU suite, part 1: the oracle itself.

The reference has no tests or golden vectors, so the oracle is pinned three
ways: (a) the committed restatement-generated fixtures, (b) agreement of two independent
restatements (numpy vs plain C), (c) the self-consistency relations the reference's own
functions offer (encode o decode = id, layout agreement, IoU invariants) and facts derived in
SURVEY.md section 8a (base-anchor values, duplicate counts).
"""
import os

import numpy as np
import pytest

import cases
from oracle import bbox_oracle as bo
from oracle import c_oracle as co


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---- (a) golden fixtures -----------------------------------------------------------------
def test_anchor_fixtures(golden_dir):
    g = _load(golden_dir, "anchors.npz")
    assert np.array_equal(bo.generate_anchors(bo.get_hyper_params("vgg16", feature_map_shape=4)), g["anchors_small_f4"])
    for bb in ("vgg16", "mobilenet_v2"):
        hp = bo.get_hyper_params(bb)
        assert np.array_equal(bo.generate_anchors(hp), g["anchors_" + bb])
        assert np.array_equal(co.generate_anchors(hp), g["anchors_" + bb])          # C restatement, bit-exact
    assert np.array_equal(bo.generate_base_anchors(bo.get_hyper_params("vgg16")), g["base_anchors_default"])


def test_boxmath_fixtures(golden_dir):
    g = _load(golden_dir, "boxmath.npz")
    scaled = bo.scale_deltas(g["deltas"], g["variances"])
    assert np.array_equal(bo.get_bboxes_from_deltas(g["anchors"], scaled), g["boxes"])
    # C uses glibc expf, numpy its own SIMD exp: a few ulp apart, far inside the 1e-4 bound
    np.testing.assert_allclose(co.decode(g["anchors"], g["deltas"], g["variances"]), g["boxes"], rtol=0, atol=2e-6)
    assert np.array_equal(bo.generate_iou_map(g["anchors"], g["gt"]), g["iou_map"])
    assert np.array_equal(co.iou_map(g["anchors"], g["gt"]), g["iou_map"])
    assert np.array_equal(bo.get_deltas_from_bboxes(g["anchors"], g["gt_per_anchor"]), g["encoded"])
    np.testing.assert_allclose(co.encode(g["anchors"], g["gt_per_anchor"]), g["encoded"], rtol=0, atol=2e-6)


_NMS_CASES = {
    "edge_default": dict(max_output_size_per_class=20, max_total_size=20),
    "edge_thr07_noclip": dict(max_output_size_per_class=64, max_total_size=64, iou_threshold=0.7, clip_boxes=False),
    "edge_score_thr": dict(max_output_size_per_class=64, max_total_size=10, score_threshold=0.5),
}


@pytest.mark.parametrize("name", sorted(_NMS_CASES))
def test_nms_edge_fixtures(golden_dir, name):
    g = _load(golden_dir, "nms.npz")
    boxes, scores = g["edge_boxes"][None, :, None, :], g["edge_scores"][None, :, None]
    for impl in (lambda: bo.combined_non_max_suppression(boxes, scores, return_indices=True, **_NMS_CASES[name]),
                 lambda: co.combined_nms(boxes, scores, **_NMS_CASES[name])):
        r = impl()
        for key, arr in zip(("boxes", "scores", "classes", "valid", "idx"), r):
            assert np.array_equal(arr, g["%s_%s" % (name, key)], equal_nan=True), (name, key)


def test_nms_multiclass_fixtures(golden_dir):
    g = _load(golden_dir, "nms.npz")
    for name, bx, kw in (
        ("mc_q1", g["mc_boxes"][:, :, None, :], dict(max_output_size_per_class=5, max_total_size=12)),
        ("mc_qc", g["mc_boxes_q"], dict(max_output_size_per_class=6, max_total_size=40, pad_per_class=True,
                                        iou_threshold=0.3)),
    ):
        for r in (bo.combined_non_max_suppression(bx, g["mc_scores"], return_indices=True, **kw),
                  co.combined_nms(bx, g["mc_scores"], **kw)):
            for key, arr in zip(("boxes", "scores", "classes", "valid", "idx"), r):
                assert np.array_equal(arr, g["%s_%s" % (name, key)]), (name, key)


# ---- (b) numpy vs C on fresh seeded inputs ----------------------------------------------------
def test_numpy_and_c_nms_agree_on_random_images():
    rng = np.random.RandomState(11)
    boxes = cases.clustered_boxes(rng, 3, 400, n_clusters=20)
    scores = cases.permutation_scores(rng, 3, 400)
    for thr in (0.5, 0.7):
        kw = dict(max_output_size_per_class=60, max_total_size=60, iou_threshold=thr)
        a = bo.combined_non_max_suppression(boxes[:, :, None, :], scores[:, :, None], return_indices=True, **kw)
        b = co.combined_nms(boxes[:, :, None, :], scores[:, :, None], **kw)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert (a[3] > 0).all()


def test_c_conv_matches_torch_conv():
    from oracle import conv_oracle as cv
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 1, size=(2, 9, 11, 5)).astype(np.float32)
    w = rng.standard_normal((3, 3, 5, 7)).astype(np.float32)
    b = rng.standard_normal((7,)).astype(np.float32)
    ref = cv.conv2d_nhwc(x, w, b, pad=(1, 1, 1, 1), act="relu")
    got = co.conv2d(x, w, b, pad_t=1, pad_l=1, out_hw=(9, 11), act="relu")
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    # stride 2 with the keras correct_pad rule for an odd size: pad (1,1)
    ref = cv.conv2d_nhwc(x, w, None, stride=2, pad=(1, 1, 1, 1))
    got = co.conv2d(x, w, None, stride=2, pad_t=1, pad_l=1, out_hw=(5, 6))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(co.maxpool2x2(x[..., :4]),
                                  x[:, :8, :10, :4].reshape(2, 4, 2, 5, 2, 4).max(axis=(2, 4)))


# ---- (c) properties and derived facts ----------------------------------------------------------
def test_base_anchor_values_match_survey():
    base = bo.generate_base_anchors(bo.get_hyper_params("vgg16"))
    assert base.shape == (9, 4)
    np.testing.assert_array_equal(base[0], np.float32([-0.128, -0.128, 0.128, 0.128]))
    np.testing.assert_allclose(base[1], [-0.18101934, -0.09050967, 0.18101934, 0.09050967], rtol=0, atol=1e-8)
    np.testing.assert_allclose(base[2], [-0.09050967, -0.18101934, 0.09050967, 0.18101934], rtol=0, atol=1e-8)
    np.testing.assert_array_equal(base[3:6], base[0:3] * 2)          # scale 256, 512: exact doubling
    np.testing.assert_array_equal(base[6:9], base[0:3] * 4)


@pytest.mark.parametrize("backbone,count,dups", [("vgg16", 8649, 744), ("mobilenet_v2", 9216, 832)])
def test_anchor_counts_and_duplicates(backbone, count, dups):
    a = bo.generate_anchors(bo.get_hyper_params(backbone))
    assert a.shape == (count, 4) and a.min() >= 0 and a.max() <= 1
    assert count - len(np.unique(a, axis=0)) == dups


def test_anchor_layout_matches_head_layout():
    """flat anchor index (y*F + x)*K + k == flat index of head output (B,F,F,K) -> (B,-1) (predictor.py:52-53)."""
    hp = bo.get_hyper_params("vgg16", feature_map_shape=5)
    F, K = 5, hp["anchor_count"]
    a = bo.generate_anchors(hp).reshape(F, F, K, 4)
    base = bo.generate_base_anchors(hp)
    centres = ((np.arange(F) / F) + 0.5 / F).astype(np.float32)
    y, x, k = 2, 3, 4                                                  # an interior cell: no clipping
    np.testing.assert_array_equal(a[y, x, k], base[k] + np.float32([centres[y], centres[x], centres[y], centres[x]]))


def test_encode_decode_round_trip():
    rng = np.random.RandomState(5)
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16", feature_map_shape=6))
    gt = cases.random_boxes(rng, (2, len(anchors)))
    deltas = bo.get_deltas_from_bboxes(anchors, gt)
    np.testing.assert_allclose(bo.get_bboxes_from_deltas(anchors, deltas), gt, rtol=0, atol=2e-6)


def test_normalize_denormalize_numpy_vs_c():
    rng = np.random.RandomState(8)
    px = rng.uniform(-20, 520, size=(3, 17, 4)).astype(np.float32)
    assert np.array_equal(bo.normalize_bboxes(px, 375, 500), co.scale_boxes(px, 375, 500, False))
    nb = rng.uniform(-0.1, 1.1, size=(3, 17, 4)).astype(np.float32)
    nb[0, 0] = [0.5 / 375, 1.5 / 500, 2.5 / 375, 0.0]             # exact .5 cases: half to even
    assert np.array_equal(bo.denormalize_bboxes(nb, 375, 500), co.scale_boxes(nb, 375, 500, True))
    assert bo.denormalize_bboxes(np.float32([[0.5 / 4, 1.5 / 4, 2.5 / 4, 3.5 / 4]]), 4, 4).tolist() == [[0, 2, 2, 4]]


def test_iou_map_invariants():
    rng = np.random.RandomState(6)
    b = cases.random_boxes(rng, (1, 40))
    iou = bo.generate_iou_map(b[0], b)[0]
    np.testing.assert_array_equal(np.diag(iou), np.ones(40, np.float32))
    np.testing.assert_array_equal(iou, iou.T)
    assert iou.min() >= 0 and iou.max() <= 1
    padded = np.zeros((1, 3, 4), np.float32)
    assert np.array_equal(bo.generate_iou_map(b[0], padded), np.zeros((1, 40, 3), np.float32))   # 0/area = 0


def test_nms_iou_differs_from_map_iou_on_flipped_corners():
    box, flipped = np.float32([0.1, 0.1, 0.5, 0.6]), np.float32([0.5, 0.6, 0.1, 0.1])
    assert bo.nms_iou(box, flipped) == np.float32(1.0)                 # TF canonicalises corners
    assert bo.generate_iou_map(box[None], flipped[None, None])[0, 0, 0] != np.float32(1.0)
    assert bo.nms_iou(box, np.float32([0.3, 0.3, 0.3, 0.9])) == 0       # zero area -> 0
    assert co.lib().orc_nms_iou(box.ctypes.data_as(co._f32p), flipped.ctypes.data_as(co._f32p)) == 1.0


def test_nms_tie_order_and_padding():
    boxes = np.float32([[0, 0, .1, .1], [.5, .5, .6, .6], [.2, .2, .3, .3], [.8, .8, .9, .9]])
    scores = np.float32([0.5, 0.5, 0.5, 0.7])
    b, s, c, v, idx = bo.combined_non_max_suppression(boxes[None, :, None], scores[None, :, None], 10, 6,
                                                      return_indices=True)
    assert v[0] == 4 and list(idx[0]) == [3, 0, 1, 2, -1, -1]           # ties -> lower index first
    assert (b[0, 4:] == 0).all() and (s[0, 4:] == 0).all() and (c == 0).all()


def test_top_k_ties_lower_index():
    assert list(bo.top_k_indices(np.float32([[0.2, 0.9, 0.9, 0.1]]), 3)[0]) == [1, 2, 0]


def test_preprocess_oracle_properties():
    """convert_image_dtype + TF2 bilinear resize restatement: identity at equal size, exact on constants,
    half-pixel sampling on a 2x upscale, flip = mirrored columns."""
    rng = np.random.RandomState(9)
    img = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    f = bo.convert_image_dtype_uint8(img)
    assert f.dtype == np.float32 and f.max() <= 1.0 and f[0, 0, 0] == np.float32(img[0, 0, 0]) * np.float32(1 / 255)
    assert np.array_equal(bo.resize_bilinear(f, 37, 53), f)
    const = np.full((5, 7, 3), 0.25, np.float32)
    assert np.array_equal(bo.resize_bilinear(const, 11, 3), np.full((11, 3, 3), 0.25, np.float32))
    ramp = np.arange(4, dtype=np.float32)[None, :, None].repeat(3, 2)          # 1x4 row: 0 1 2 3
    up = bo.resize_bilinear(ramp, 1, 8)[0, :, 0]
    np.testing.assert_allclose(up, [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0], atol=1e-6)   # edges clamp
    out = bo.preprocess_image(img, 50, 50)
    assert np.array_equal(bo.preprocess_image(img, 50, 50, flip=True), out[:, ::-1])
    b = np.float32([[0.1, 0.2, 0.5, 0.6]])
    assert np.allclose(bo.flip_boxes_horizontally(b), [[0.1, 0.4, 0.5, 0.8]])


def _target_case(backbone="vgg16", B=3, G=42, n_valid=6, seed=0):
    hp = bo.get_hyper_params(backbone)
    anchors = bo.generate_anchors(hp)
    rng = np.random.RandomState(seed)
    gt = cases.gt_boxes(rng, B, G=G, n_valid=n_valid)
    labels = np.full((B, G), -1, np.int32)
    labels[:, :n_valid] = rng.randint(1, 21, size=(B, n_valid))
    A = len(anchors)
    rp = rng.randint(1, 1280, size=(B, A)).astype(np.int32)
    rn = rng.randint(1, 2560, size=(B, A)).astype(np.int32)
    return hp, anchors, gt, labels, rp, rn


def test_rpn_targets_oracle_properties():
    """calculate_rpn_actual_outputs restatement: 256 sampled anchors per image, positives are IoU > 0.7 or the best
    anchor of a valid gt, negatives IoU < 0.3, deltas of positives decode back to their gt box, the rest are zero."""
    hp, anchors, gt, labels, rp, rn = _target_case()
    deltas, lab = bo.calculate_rpn_actual_outputs(anchors, gt, labels, hp, rp, rn)
    B, A = deltas.shape[0], deltas.shape[1]
    lab = lab.reshape(B, A)
    assert set(np.unique(lab)) <= {-1.0, 0.0, 1.0}
    iou = bo.generate_iou_map(anchors, gt)
    merged = iou.max(axis=2)
    for b in range(B):
        pos, neg = lab[b] == 1, lab[b] == 0
        assert pos.sum() + neg.sum() == 256 and 0 < pos.sum() <= 128
        best = {int(np.argmax(iou[b, :, g])) for g in range(6)}
        assert all(merged[b, i] > np.float32(0.7) or i in best for i in np.nonzero(pos)[0])
        assert (merged[b, neg] < np.float32(0.3)).all()
        assert (deltas[b, ~pos] == 0).all()
        back = bo.get_bboxes_from_deltas(anchors[pos], (deltas[b, pos] * np.float32(hp["variances"]))[None])[0]
        np.testing.assert_allclose(back, gt[b, np.argmax(iou[b, pos], axis=1)], atol=3e-6)


def test_randomly_select_ties_and_short_masks():
    mask = np.array([[1, 0, 1, 1, 1, 0, 1]], bool)
    rnd = np.array([[5, 9, 5, 7, 5, 9, 1]], np.int32)
    got = bo.randomly_select_xyz_mask(mask, np.array([3]), rnd)
    assert got.tolist() == [[True, False, True, True, False, False, False]]      # 7, then the two lowest-index 5s
    assert np.array_equal(bo.randomly_select_xyz_mask(mask, np.array([10]), rnd), mask)          # fewer than asked
    assert not bo.randomly_select_xyz_mask(mask, np.array([0]), rnd).any()


def test_resize_restatement_matches_scikit_image():
    """Independent pin of the bilinear-resize restatement (data_utils.py:26): scikit-image's order-1 resize without
    anti-aliasing samples the same half-pixel-centre grid as TF 2.x (tests/golden/make_resize_golden.py, generated with
    the image's scikit-image 0.18.3; up- and down-scaling, identity, non-square)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_skimage.npz"))
    for i in range(5):
        src, ref = d["in%d" % i], d["out%d" % i]
        got = bo.preprocess_image(src, ref.shape[0], ref.shape[1])
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-5          # float32 rounding of a [0,1] image (observed 7e-6)


# ---- (d) TensorFlow's own published known-answer tests for the NMS kernels (recalled; tests/golden/tf_nms_kat.py) ----
def _kat():
    import importlib.util
    spec = importlib.util.spec_from_file_location("tf_nms_kat", os.path.join(os.path.dirname(__file__), "golden", "tf_nms_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_nms_restatements_against_recalled_tf_unit_test_vectors(impl):
    kat = _kat()

    def run(boxes, scores, per_class, total, iou, thr, clip):
        b, s = boxes[None, :, None, :], scores[None, :, None]
        if impl == "c":
            return co.combined_nms(b, s, per_class, total, iou_threshold=iou, score_threshold=thr, clip_boxes=clip)
        ob, osc, oc, ov, oi = bo.combined_non_max_suppression(b, s, per_class, total, iou_threshold=iou, score_threshold=thr,
                                                              clip_boxes=clip, return_indices=True)
        return ob, osc, oc, ov, oi

    for name, boxes, scores, max_out, iou, thr, want in kat.NMS_CASES:
        _ob, _os, _oc, ov, oi = run(boxes, scores, max_out, max_out, iou, thr, False)
        assert int(ov[0]) == len(want) and oi[0, :len(want)].tolist() == want, name
        assert (oi[0, len(want):] == -1).all(), name
    for name, boxes, scores, per_class, total, iou, thr, clip, wb, ws, wv in kat.COMBINED_CASES:
        ob, osc, oc, ov, _oi = run(boxes, scores, per_class, total, iou, thr, clip)
        assert int(ov[0]) == wv, name
        assert np.array_equal(ob[0], wb) and np.array_equal(osc[0], ws) and (oc[0] == 0).all(), name


def test_c_oracle_under_address_and_ub_sanitizers():
    """The plain-C restatement is the checker of every GPU parity test: run this module's tests again in a child process on an
    AddressSanitizer + UBSan build of oracle/rpn_oracle.c (``make -C oracle san``; libasan preloaded into the child's python), so
    that an out-of-bounds read or signed overflow in the CHECKER cannot pass as agreement.  (Sanitizers run on the CPU build only;
    the GPU pool has no ASan.)"""
    import subprocess
    import sys
    if os.environ.get("RPN_ORACLE_SO"):
        pytest.skip("already inside the sanitizer child")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan for this gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "san"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               RPN_ORACLE_SO=os.path.join(root, "oracle", "_build", "librpn_oracle_san.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "1 skipped" in r.stdout, r.stdout[-500:]
