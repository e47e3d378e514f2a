"""GPU parity, box math: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact for anchors, IoU map and every NMS integer output (indices,
valid counts); <= 1e-4 absolute on float32 boxes / scores (the north star's tolerance), with the
much tighter bound actually observed asserted where the arithmetic is identical."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import bbox_oracle as bo
from oracle import c_oracle as co
from tf_rpn_amd.utils import bbox_utils

pytestmark = pytest.mark.gpu
TOL = 1e-4                       # north-star tolerance for float32 coordinates / scores
VAR = np.float32([0.1, 0.1, 0.2, 0.2])


def _np(t):
    return t.cpu().numpy() if isinstance(t, torch.Tensor) else t


# ---- anchors: bit-exact -------------------------------------------------------------------
@pytest.mark.parametrize("backbone,kw", [
    ("vgg16", {}), ("mobilenet_v2", {}),
    ("vgg16", dict(feature_map_shape=4)),
    ("mobilenet_v2", dict(img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1 / 3.])),   # C5
    ("vgg16", dict(img_size=333, feature_map_shape=7, anchor_scales=[37, 290], anchor_ratios=[0.7, 1.9, 3.3])),
])
def test_anchors_bit_exact(backbone, kw):
    hp = bo.get_hyper_params(backbone, **kw)
    got = bbox_utils.generate_anchors(hp, as_numpy=True)
    assert np.array_equal(got, bo.generate_anchors(hp))


def test_anchor_golden_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "anchors.npz"))
    assert np.array_equal(bbox_utils.generate_anchors(bo.get_hyper_params("vgg16"), as_numpy=True), g["anchors_vgg16"])
    hp_c5 = bo.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                anchor_ratios=[1., 2., .5, 3., 1 / 3.])
    a5 = bbox_utils.generate_anchors(hp_c5, as_numpy=True)
    assert np.array_equal(a5[:600], g["anchors_c5_head"])
    assert a5.astype(np.float64).sum() == g["anchors_c5_sum"][0] and len(a5) == g["anchors_c5_sum"][1]


# ---- decode / encode ------------------------------------------------------------------------
@pytest.mark.parametrize("B,backbone", [(1, "vgg16"), (8, "vgg16"), (64, "mobilenet_v2")])
def test_decode_matches_oracle(B, backbone):
    anchors = bo.generate_anchors(bo.get_hyper_params(backbone))
    deltas = np.random.RandomState(2).standard_normal((B, len(anchors), 4)).astype(np.float32)
    ref = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))
    got = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)          # fused x variances
    assert np.abs(got - ref).max() <= 2e-6 < TOL
    got2 = bbox_utils.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))  # the reference's two steps
    assert np.array_equal(got, got2)
    batched = np.broadcast_to(anchors, (B,) + anchors.shape).copy()                 # (B,A,4) anchors
    assert np.array_equal(bbox_utils.get_bboxes_from_deltas(batched, deltas, variances=VAR), got)


def test_decode_golden_and_edge_shapes(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxmath.npz"))
    got = bbox_utils.get_bboxes_from_deltas(g["anchors"], g["deltas"], variances=g["variances"])
    assert np.abs(got - g["boxes"]).max() <= 2e-6
    empty = bbox_utils.get_bboxes_from_deltas(g["anchors"], np.zeros((0, len(g["anchors"]), 4), np.float32))
    assert empty.shape == (0, len(g["anchors"]), 4)
    with pytest.raises(ValueError):
        bbox_utils.get_bboxes_from_deltas(g["anchors"][:-1], g["deltas"])
    # unbounded exp (no clamp in the reference, bbox_utils.py:86-87): inf/NaN must match the oracle's
    big = np.float32([[[0, 0, 100, -100], [0, 0, np.inf, 1]]])
    an = g["anchors"][:2]
    with np.errstate(all="ignore"):
        ref = bo.get_bboxes_from_deltas(an, big)
    got = bbox_utils.get_bboxes_from_deltas(an, big)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref)) and np.array_equal(np.isnan(got), np.isnan(ref))


def test_encode_matches_oracle_and_round_trips(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxmath.npz"))
    got = bbox_utils.get_deltas_from_bboxes(g["anchors"], g["gt_per_anchor"])
    assert np.abs(got - g["encoded"]).max() <= 2e-6
    assert (got[0, 5] == 0).all()                                  # zero-size gt row -> zero deltas (:119-122)
    back = bbox_utils.get_bboxes_from_deltas(g["anchors"], got)
    keep = np.ones(got.shape[:2], bool)
    keep[0, 5] = False
    assert np.abs(back - g["gt_per_anchor"])[keep].max() <= 4e-6


def test_normalize_denormalize_bit_exact():
    rng = np.random.RandomState(8)
    px = rng.uniform(-20, 520, size=(3, 1000, 4)).astype(np.float32)
    assert np.array_equal(bbox_utils.normalize_bboxes(px, 375, 500), bo.normalize_bboxes(px, 375, 500))
    nb = rng.uniform(-0.1, 1.1, size=(3, 1000, 4)).astype(np.float32)
    nb[0, 0] = [0.5 / 4, 1.5 / 4, 2.5 / 4, 3.5 / 4]
    assert np.array_equal(bbox_utils.denormalize_bboxes(nb, 375, 500), bo.denormalize_bboxes(nb, 375, 500))
    assert bbox_utils.denormalize_bboxes(nb[:1, :1], 4, 4).tolist() == [[[0, 2, 2, 4]]]     # tf.round: half to even


# ---- IoU map: bit-exact ---------------------------------------------------------------------
@pytest.mark.parametrize("B,G,backbone", [(1, 1, "vgg16"), (3, 7, "vgg16"), (64, 42, "vgg16"), (8, 42, "mobilenet_v2")])
def test_iou_map_bit_exact(B, G, backbone):
    anchors = bo.generate_anchors(bo.get_hyper_params(backbone))
    gt = cases.gt_boxes(np.random.RandomState(4), B, G=G, n_valid=min(10, G))
    got = bbox_utils.generate_iou_map(anchors, gt)
    assert got.shape == (B, len(anchors), G)
    assert np.array_equal(got, co.iou_map(anchors, gt))


def test_iou_map_golden_batched_and_invariants(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxmath.npz"))
    assert np.array_equal(bbox_utils.generate_iou_map(g["anchors"], g["gt"]), g["iou_map"])
    b = cases.random_boxes(np.random.RandomState(6), (2, 50))
    iou = bbox_utils.generate_iou_map(b, b)                        # batched bboxes (B,A,4)
    assert np.array_equal(iou, bo.generate_iou_map(b, b))
    assert (np.diagonal(iou, axis1=1, axis2=2) == 1).all() and np.array_equal(iou, iou.transpose(0, 2, 1))
    assert bbox_utils.generate_iou_map(b[0], np.zeros((2, 0, 4), np.float32)).shape == (2, 50, 0)


@pytest.mark.parametrize("B,A,G,batched", [(3, 1001, 7, True), (5, 1027, 5, False), (2, 4099, 4, True), (7, 600, 9, True),
                                              (64, 8649, 42, False), (1, 61440, 42, False)])
def test_iou_map_chunked_kernel_alignment_cases(B, A, G, batched):
    """The chunked IoU kernel cuts the flat (B, A, G) map into 16-byte-aligned runs of 4096 floats: image slabs that are
    not multiples of 4 floats (runs straddle two images), G = 4 (every vector crosses an anchor), a total that is not a
    multiple of 4 (scalar tail), batched and shared bboxes -- all bit-exact against the C restatement."""
    rng = np.random.RandomState(B * 1000 + A)
    boxes = cases.random_boxes(rng, (B, A) if batched else (A,))
    gt = cases.gt_boxes(rng, B, G=G, n_valid=max(1, G // 2))
    got = bbox_utils.generate_iou_map(boxes, gt)
    want = co.iou_map(boxes, gt) if not batched else np.stack([co.iou_map(boxes[b], gt[b:b + 1])[0] for b in range(B)])
    assert got.shape == (B, A, G) and np.array_equal(got, want)


@pytest.mark.parametrize("G", [1, 2, 6, 9, 21, 42])
def test_iou_map_rows_kernel_divide_is_exact_everywhere(G):
    """The row kernel forms the quotient with the bare arithmetic of the IEEE divide (reciprocal + fma chain, packed two
    pairs per instruction) and falls back to the plain divide when an operand leaves [2^-60, 2^60].  Boxes that sit on
    and beyond those limits -- tiny, huge, zero-area, flipped (negative areas), denormal coordinates, NaN and inf rows,
    exact duplicates (IoU == 1) -- mixed into ordinary ones, anchors A not a multiple of 64, odd G: every float equal to
    the C restatement's, NaN for NaN."""
    rng = np.random.RandomState(100 + G)
    B, A = 5, 64 * 7 + 13
    boxes = cases.random_boxes(rng, (A,)).astype(np.float32)
    gt = cases.gt_boxes(rng, B, G=G, n_valid=G).astype(np.float32)
    def tiny(n, scale):                                      # boxes of side `scale` at random places
        p = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
        return np.concatenate([p, p + np.float32(scale)], axis=1).astype(np.float32)
    specials = [tiny(8, 1e-9), tiny(8, 1e-19), tiny(4, 1e-30) * np.float32(1e-8), tiny(6, 1.0) * np.float32(1e12),
                tiny(4, 1.0) * np.float32(3e18), np.zeros((3, 4), np.float32),
                np.float32([[0.6, 0.6, 0.2, 0.2], [0.1, 0.9, 0.5, 0.3]]),            # flipped corners
                np.float32([[np.nan, 0.1, 0.5, 0.5], [0.1, 0.1, np.inf, 0.5], [-np.inf, 0, 1, 1]]),
                np.float32([[1e-42, 1e-42, 3e-42, 5e-42], [0, 0, 1e-45, 1]])]          # denormal coordinates
    sp = np.concatenate(specials, axis=0)
    at = rng.choice(A, size=len(sp), replace=False)
    boxes[at] = sp
    for b in range(B):                                       # the same specials among the gt rows of some images
        if G >= 2 and b % 2 == 0:
            k = min(G // 2, len(sp))
            gt[b, rng.choice(G, size=k, replace=False)] = sp[rng.choice(len(sp), size=k, replace=False)]
    gt[1, 0] = boxes[5]                                      # an exact duplicate: IoU == 1
    got = bbox_utils.generate_iou_map(boxes, gt)
    want = co.iou_map(boxes, gt)
    assert got.shape == want.shape == (B, A, G)
    assert np.array_equal(got.view(np.uint32) & 0x7fc00000 == 0x7fc00000, np.isnan(want))        # NaN exactly where the oracle has one
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    # and the ordinary rows alone (no special anywhere): the fast path's own results, bit for bit
    plain = cases.random_boxes(rng, (A,)).astype(np.float32)
    gtp = cases.gt_boxes(rng, B, G=G, n_valid=G).astype(np.float32)
    assert np.array_equal(bbox_utils.generate_iou_map(plain, gtp).view(np.uint32), co.iou_map(plain, gtp).view(np.uint32))


# ---- NMS: integer outputs bit-exact -------------------------------------------------------------
def _check_nms(boxes, scores, **kw):
    ref = co.combined_nms(boxes, scores, **kw)
    got = bbox_utils.non_max_suppression(boxes, scores, return_indices=True, **kw)
    names = ("boxes", "scores", "classes", "valid", "idx")
    for n, r, x in zip(names, ref, got):
        assert r.shape == x.shape, n
    assert np.array_equal(got[3], ref[3]), "valid_detections differ: %s vs %s" % (got[3], ref[3])
    assert np.array_equal(got[4], ref[4]), "selected indices differ"
    assert np.array_equal(got[2], ref[2])
    assert np.array_equal(got[0], ref[0], equal_nan=True) and np.array_equal(got[1], ref[1], equal_nan=True)
    return got


def test_nms_golden_edge_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "nms.npz"))
    boxes, scores = g["edge_boxes"][None, :, None, :], g["edge_scores"][None, :, None]
    for name, kw in {
        "edge_default": dict(max_output_size_per_class=20, max_total_size=20),
        "edge_thr07_noclip": dict(max_output_size_per_class=64, max_total_size=64, iou_threshold=0.7, clip_boxes=False),
        "edge_score_thr": dict(max_output_size_per_class=64, max_total_size=10, score_threshold=0.5),
    }.items():
        got = _check_nms(boxes, scores, **kw)
        for key, arr in zip(("boxes", "scores", "classes", "valid", "idx"), got):
            assert np.array_equal(arr, g["%s_%s" % (name, key)], equal_nan=True), (name, key)


def test_nms_golden_multiclass(golden_dir):
    g = np.load(os.path.join(golden_dir, "nms.npz"))
    got = _check_nms(g["mc_boxes"][:, :, None, :], g["mc_scores"], max_output_size_per_class=5, max_total_size=12)
    assert np.array_equal(got[4], g["mc_q1_idx"]) and np.array_equal(got[2], g["mc_q1_classes"])
    got = _check_nms(g["mc_boxes_q"], g["mc_scores"], max_output_size_per_class=6, max_total_size=40,
                     pad_per_class=True, iou_threshold=0.3)
    assert got[0].shape == (2, 18, 4) and np.array_equal(got[4], g["mc_qc_idx"])


@pytest.mark.parametrize("iou_thr", [0.5, 0.7])
@pytest.mark.parametrize("backbone,B", [("vgg16", 8), ("mobilenet_v2", 4)])
def test_nms_full_size_decoded_anchors(backbone, B, iou_thr):
    """C3-shaped input: decoded anchors (N(0,1) deltas x variances), tie-free permutation scores, top 300."""
    anchors = bo.generate_anchors(bo.get_hyper_params(backbone))
    A = len(anchors)
    deltas = np.random.RandomState(2).standard_normal((B, A, 4)).astype(np.float32)
    boxes = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))
    scores = cases.permutation_scores(np.random.RandomState(3), B, A)
    got = _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                     iou_threshold=iou_thr)
    assert (got[3] == 300).all()


@pytest.mark.parametrize("iou_thr", [0.5, 0.7])
def test_nms_deep_walk_on_a_smooth_score_field(iou_thr):
    """What a trained head produces rather than C3's random draw: boxes close to their anchors (small deltas) and a smooth
    objectness field, so that neighbours overlap heavily AND have near-equal scores -- thousands of candidates are visited,
    across several bands of adaptive size, for 300 outputs."""
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
    A = len(anchors)
    rng = np.random.RandomState(31)
    deltas = (0.3 * rng.standard_normal((3, A, 4))).astype(np.float32)
    f = rng.standard_normal((3, 31, 31, 9)).astype(np.float32)
    for _ in range(3):
        f = (f + np.roll(f, 1, 1) + np.roll(f, 1, 2) + np.roll(f, -1, 1) + np.roll(f, -1, 2)) / 5
    scores = (1 / (1 + np.exp(-8 * f))).reshape(3, A).astype(np.float32)
    boxes = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))
    got = _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                     iou_threshold=iou_thr)
    assert (got[3] > 200).all()          # (at 0.5 every candidate is visited and fewer than 300 survive)
    fb, fs, fi, fv = bbox_utils.decode_and_nms(anchors, deltas, scores, VAR, 300, iou_threshold=iou_thr)
    gpu_boxes = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)
    rb, rs, _rc, rv, ri = co.combined_nms(gpu_boxes[:, :, None, :], scores[:, :, None], 300, 300, iou_threshold=iou_thr)
    assert np.array_equal(fv, rv) and np.array_equal(fi, ri) and np.array_equal(fb, rb)


def test_nms_ties_and_duplicates_on_raw_anchors():
    """Clipped anchors contain 744 exact duplicate rows; with constant / coarse scores every tie rule fires."""
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
    A = len(anchors)
    rng = np.random.RandomState(9)
    scores = np.stack([np.full(A, 0.5, np.float32), (rng.randint(0, 8, size=A) / 8).astype(np.float32)])
    boxes = np.stack([anchors, anchors])
    for thr in (0.3, 0.7, 1.0):
        _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                   iou_threshold=thr)


def test_nms_heavy_suppression_walks_many_chunks():
    """Tight clusters: far more than 256 candidates are visited to collect the outputs."""
    rng = np.random.RandomState(12)
    boxes = cases.clustered_boxes(rng, 3, 6000, n_clusters=40, jitter=0.004)
    scores = cases.permutation_scores(rng, 3, 6000)
    got = _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                     iou_threshold=0.5)
    assert (got[3] < 300).all() and (got[3] >= 20).all()      # ~40 overlapping clusters survive as 30-40 boxes


def test_nms_ratios_on_and_next_to_the_threshold():
    """IoUs that are exactly the threshold, or one float above / below it: dyadic boxes whose intersection / union is exactly
    1/4, 1/2, 3/4 (the kernel's fused pre-test is undecided there and must fall back to the IEEE quotient), with the
    threshold at that value and at its two neighbours; and ratios a few 2^-20 away from the threshold."""
    rng = np.random.RandomState(21)
    rows = []
    for k in range(400):                                     # pairs (a, b): b shares a's corner, IoU = 1/4, 1/2 or 3/4
        s = 2.0 ** -rng.randint(3, 6)
        y, x = rng.randint(0, 8, size=2) * 0.125
        f = (0.25, 0.5, 0.75)[k % 3]
        rows += [[y, x, y + s, x + s], [y, x, y + s, x + s * f]]
    boxes = np.float32(rows)
    boxes = boxes[rng.permutation(len(boxes))][None]
    scores = cases.permutation_scores(rng, 1, boxes.shape[1])
    for centre in (0.25, 0.5, 0.75):
        c = np.float32(centre)
        for thr in (np.nextafter(c, np.float32(0)), c, np.nextafter(c, np.float32(1))):
            _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=800, max_total_size=800,
                       iou_threshold=float(thr), clip_boxes=False)
    # near misses: b = a narrowed by (1 - f) with f = thr * (1 + k 2^-20), k = -8 .. 8
    rows = []
    for k in range(-8, 9):
        for rep in range(12):
            s = np.float32(rng.uniform(0.05, 0.3))
            y, x = np.float32(rng.uniform(0, 0.6, size=2))
            f = np.float32(0.7) * np.float32(1 + k * 2.0 ** -20)
            rows += [[y, x, y + s, x + s], [y, x, y + s, x + s * f]]
    boxes = np.float32(rows)[None]
    scores = cases.permutation_scores(rng, 1, boxes.shape[1])
    _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
               iou_threshold=0.7, clip_boxes=False)


def _nested_families(rng, thr, n_fam=640):
    """Families (outer box, inner box nested in it): the IoU of a nested pair IS its area ratio, set to thr * (1 + k 2^-20)
    for k over a range that straddles both the threshold and the area-pruning bound thr * (1 - 2^-19); families sit at
    magnitudes 2^-9 .. 2^2 (areas 2^-22 .. 2^4: below the first area bin, inside the table, above its last bin)."""
    rows = []
    for f in range(n_fam):
        mag = np.float32(2.0 ** rng.randint(-9, 3))
        y, x = (np.float32(rng.uniform(0, 40, size=2)) * mag).astype(np.float32)
        h, w = (np.float32(rng.uniform(0.5, 1.0, size=2)) * mag).astype(np.float32)
        k = rng.randint(-48, 49)
        r = np.float32(thr) * np.float32(1 + k * 2.0 ** -20)
        if f % 3 == 0:                                   # ratio split over both sides
            fy = np.float32(np.sqrt(r)); fx = np.float32(r / fy)
        else:
            fy = np.float32(1); fx = r
        rows.append(([y, x, y + h, x + w], [y, x, y + h * fy, x + w * fx]))
    return rows


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7, 0.95])
def test_nms_area_pruning_bounds_on_nested_boxes(thr):
    """The tests against the boxes selected in earlier chunks skip pairs whose areas differ by more than the threshold allows
    (nms_kernels.hip, prune_bin).  Sharpest case: nested boxes, whose IoU equals the area ratio -- ratios on both sides of the
    threshold and of the pruning bound, at every magnitude of the area-bin table, with the partner selected several chunks
    (256 candidates) earlier: first the outer boxes lead (a candidate smaller than the selected box), then the inner ones."""
    rng = np.random.RandomState(41)
    fam = _nested_families(rng, thr)
    n = len(fam)
    for inner_first in (False, True):
        first = np.float32([f[1 if inner_first else 0] for f in fam])
        second = np.float32([f[0 if inner_first else 1] for f in fam])
        boxes = np.concatenate([first, second])[None]
        base = cases.permutation_scores(rng, 1, 2 * n)[0]
        order = np.sort(base)[::-1]
        scores = np.empty(2 * n, np.float32)
        scores[rng.permutation(n)] = order[:n]                     # the leaders take the n best scores
        scores[n + rng.permutation(n)] = order[n:]
        got = _check_nms(boxes[:, :, None, :], scores[None, :, None], max_output_size_per_class=1200, max_total_size=1200,
                         iou_threshold=thr, clip_boxes=False)
        assert n // 2 < got[3][0] < 2 * n                        # (some partners are suppressed, some are not)


@pytest.mark.gpu
def test_nms_without_area_pruning_subprocess():
    """RPN_NMS_PRUNE=0 (read once per process): every candidate is tested against the whole selected list, as before the area
    bins existed.  Same outputs as the oracle on C3-shaped input and on the nested-box families."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import cases
        from oracle import bbox_oracle as bo
        from tests.test_gpu_bbox import _check_nms, VAR, test_nms_area_pruning_bounds_on_nested_boxes
        anchors = bo.generate_anchors(bo.get_hyper_params("vgg16")); A = len(anchors)
        deltas = np.random.RandomState(2).standard_normal((4, A, 4)).astype(np.float32)
        boxes = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))
        scores = cases.permutation_scores(np.random.RandomState(3), 4, A)
        for thr in (0.5, 0.7):
            _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                       iou_threshold=thr)
        test_nms_area_pruning_bounds_on_nested_boxes(0.7)
        print("nms unpruned ok")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RPN_NMS_PRUNE="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "nms unpruned ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "2"])
def test_nms_without_the_linear_histogram_subprocess(mode):
    """RPN_NMS_LINEAR (read once per process) = 0: radix select + compaction + bitonic sort only; = 2: histogram select,
    band sorted the old way.  Same outputs, bit for bit, as the oracle on C3-shaped, clustered and multi-band inputs."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import cases
        from oracle import bbox_oracle as bo
        from tests.test_gpu_bbox import _check_nms, VAR
        anchors = bo.generate_anchors(bo.get_hyper_params("vgg16")); A = len(anchors)
        deltas = np.random.RandomState(2).standard_normal((4, A, 4)).astype(np.float32)
        boxes = bo.get_bboxes_from_deltas(anchors, bo.scale_deltas(deltas, VAR))
        scores = cases.permutation_scores(np.random.RandomState(3), 4, A)
        for thr in (0.5, 0.7):
            got = _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                             iou_threshold=thr)
            assert (got[3] == 300).all()
        rng = np.random.RandomState(15)
        cl = cases.clustered_boxes(rng, 2, 20000, n_clusters=60, jitter=0.002)
        _check_nms(cl[:, :, None, :], cases.permutation_scores(rng, 2, 20000)[:, :, None], max_output_size_per_class=300,
                   max_total_size=300, iou_threshold=0.4)
        print("nms fallback ok")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RPN_NMS_LINEAR=mode), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "nms fallback ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("size", ["4", "16"])
def test_nms_cluster_mode_forced_subprocess(size):
    """RPN_NMS_CLUSTER (read once per process) = k: k workgroups per (image, class) share the first band's passes over the
    scores wherever they fit -- automatic only for few images with >= 16 384 candidates, forced here so that the small and
    special problems (ties, duplicates, thresholds on a ratio, several classes, empty slices) go through the cluster
    barriers too.  Same outputs, bit for bit, as the oracle."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import tests.test_gpu_bbox as t
        t.test_nms_randomised_shapes_thresholds_and_box_kinds()
        t.test_nms_ties_and_duplicates_on_raw_anchors()
        t.test_nms_ratios_on_and_next_to_the_threshold()
        t.test_nms_multiclass_random()
        t.test_nms_thresholds_sizes_and_limits()
        t.test_nms_bunched_and_saturated_scores()
        t.test_nms_full_size_decoded_anchors("vgg16", 8, 0.7)
        t.test_decode_nms_fused_equals_two_step("mobilenet_v2", 3)
        t.test_nms_kernel_against_recalled_tf_unit_test_vectors()
        print("nms cluster ok")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RPN_NMS_CLUSTER=size), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "nms cluster ok" in r.stdout


def test_nms_randomised_shapes_thresholds_and_box_kinds():
    """240 seeded random problems against the oracle, bit for bit: N from 1 to 3000 (partial groups, partial chunks, several
    chunks), output sizes from 1 to 400, thresholds over (0, 1) including dyadic ones that the boxes' ratios hit exactly,
    random / tightly clustered / grid-aligned / duplicated / flipped / degenerate boxes, tie-free and heavily tied scores,
    with and without a score threshold."""
    rng = np.random.RandomState(77)
    for it in range(240):
        B = int(rng.randint(1, 4))
        N = int(rng.choice([rng.randint(1, 70), rng.randint(60, 330), rng.randint(300, 3000)]))
        kind = it % 4
        if kind == 0:
            boxes = cases.random_boxes(rng, (B, N))
        elif kind == 1:
            boxes = cases.clustered_boxes(rng, B, N, n_clusters=int(rng.randint(1, 30)), jitter=float(rng.choice([0.0, 0.002, 0.02])))
        elif kind == 2:       # grid-aligned boxes: intersections / unions in exact dyadic ratios
            y, x = rng.randint(0, 12, size=(2, B, N)) / 16.0
            h, w = rng.randint(1, 5, size=(2, B, N)) / 16.0
            boxes = np.stack([y, x, y + h, x + w], axis=-1).astype(np.float32)
        else:                 # a mixture with duplicates, flipped corners and degenerate boxes
            boxes = cases.clustered_boxes(rng, B, N, n_clusters=8)
            k = max(1, N // 5)
            boxes[:, rng.randint(0, N, size=k)] = boxes[:, rng.randint(0, N, size=k)]
            fl = rng.randint(0, N, size=k)
            boxes[:, fl] = boxes[:, fl][..., [2, 3, 0, 1]]
            boxes[:, rng.randint(0, N, size=max(1, N // 20)), 2] = boxes[:, rng.randint(0, N, size=max(1, N // 20)), 0]
        if it % 3 == 0:
            scores = cases.permutation_scores(rng, B, N)
        elif it % 3 == 1:
            scores = (rng.randint(0, 6, size=(B, N)) / 8.0).astype(np.float32)           # heavy ties: the index order decides
        else:
            scores = rng.uniform(-0.5, 1.5, size=(B, N)).astype(np.float32)              # outside [0, 1): the radix path
        thr = float(rng.choice([rng.uniform(0.05, 0.95), 0.25, 0.5, 0.75, 1.0 / 3.0]))
        m = int(rng.choice([1, 7, 64, 100, 300, 400]))
        kw = dict(max_output_size_per_class=m, max_total_size=int(rng.choice([m, max(1, m // 2)])), iou_threshold=thr,
                  clip_boxes=bool(it & 1))
        if it % 5 == 0:
            kw["score_threshold"] = float(rng.uniform(0.0, 0.6))
        try:
            _check_nms(boxes[:, :, None, :], scores[:, :, None], **kw)
        except AssertionError as e:
            raise AssertionError("case %d (B=%d N=%d kind=%d %r): %s" % (it, B, N, kind, kw, e))


def test_nms_bunched_and_saturated_scores():
    """Scores bunched into a few of the selection histogram's 2048 linear bins (so the kernel refines inside the crossing
    bin), including the top bin of a saturated sigmoid with scores exactly 1.0, ties inside a sub-bin and a bunch whose
    crossing bin is bin 0 / holds more ties than a band may take (the radix select takes over)."""
    rng = np.random.RandomState(41)
    N = 30000
    boxes = cases.random_boxes(rng, (2, N), smin=0.02, smax=0.08)
    b4 = boxes[:, :, None, :]
    sat = 1.0 / (1.0 + np.exp(-rng.normal(11.0, 4.0, size=(2, N))))                  # most of them in [0.9995, 1]
    sat = sat.astype(np.float32)
    assert (sat == 1.0).sum() > 10 and (sat > 2047 / 2048).mean() > 0.5
    mid = (0.5 + 1e-4 * rng.standard_normal((2, N))).astype(np.float32)               # one or two bins around 0.5
    coarse = (np.float32(0.75) + rng.randint(0, 40, size=(2, N)).astype(np.float32) * np.float32(2.0 ** -23))   # 40 distinct values
    low = (1e-5 * rng.uniform(size=(2, N))).astype(np.float32)                       # everything in bin 0
    for scores in (sat, mid, coarse, low):
        for thr in (0.5, 0.7):
            _check_nms(b4, scores[:, :, None], max_output_size_per_class=300, max_total_size=300, iou_threshold=thr)


def test_nms_thresholds_sizes_and_limits():
    rng = np.random.RandomState(13)
    boxes = cases.clustered_boxes(rng, 2, 1000, n_clusters=30)
    scores = rng.uniform(-1, 1, size=(2, 1000)).astype(np.float32)
    b4, s3 = boxes[:, :, None, :], scores[:, :, None]
    _check_nms(b4, s3, max_output_size_per_class=50, max_total_size=80, score_threshold=0.25)
    _check_nms(b4, s3, max_output_size_per_class=100, max_total_size=30, iou_threshold=0.6)
    _check_nms(b4, s3, max_output_size_per_class=1000, max_total_size=1000, iou_threshold=0.9)    # more slots than survivors
    _check_nms(b4, s3, max_output_size_per_class=40, max_total_size=40, score_threshold=2.0)       # nothing qualifies
    _check_nms(b4, s3, max_output_size_per_class=7, max_total_size=7, iou_threshold=0.0)
    _check_nms(b4, s3, max_output_size_per_class=7, max_total_size=7, iou_threshold=-1.0)          # IoU 0 > thr: only the top box
    _check_nms(b4, s3, max_output_size_per_class=0, max_total_size=5)
    # N = 1, N = 2, and N beyond one band of the radix select (4096 sorted candidates per band)
    _check_nms(b4[:, :1], s3[:, :1], max_output_size_per_class=3, max_total_size=3)
    _check_nms(b4[:, :2], s3[:, :2], max_output_size_per_class=3, max_total_size=3)
    big = cases.clustered_boxes(rng, 1, 16384, n_clusters=200)
    _check_nms(big[:, :, None, :], cases.permutation_scores(rng, 1, 16384)[:, :, None],
               max_output_size_per_class=300, max_total_size=300)


def test_nms_walks_several_bands():
    """Selections that need more candidates than one 4096-key band: tight clusters (deep walk), huge output
    sizes, and all-equal scores (the radix select must descend into the index bits of the key)."""
    rng = np.random.RandomState(15)
    boxes = cases.clustered_boxes(rng, 2, 20000, n_clusters=60, jitter=0.002)
    scores = cases.permutation_scores(rng, 2, 20000)
    got = _check_nms(boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                     iou_threshold=0.4)
    assert (got[3] < 300).all()                                    # every band was consumed
    spread = cases.random_boxes(rng, (1, 12000), smin=0.01, smax=0.03)
    _check_nms(spread[:, :, None, :], cases.permutation_scores(rng, 1, 12000)[:, :, None],
               max_output_size_per_class=2000, max_total_size=2000, iou_threshold=0.5)
    const = np.full((1, 9000, 1), 0.25, np.float32)
    _check_nms(spread[:, :9000, None, :], const, max_output_size_per_class=1500, max_total_size=1500)


def test_nms_c5_anchor_count():
    """BASELINE config C5: 64x64 feature map x 15 anchors = 61 440 candidates per image."""
    hp = bo.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64, anchor_ratios=[1., 2., .5, 3., 1 / 3.])
    anchors = bo.generate_anchors(hp)
    A = len(anchors)
    assert A == 61440
    deltas = np.random.RandomState(2).standard_normal((2, A, 4)).astype(np.float32)
    scores = cases.permutation_scores(np.random.RandomState(3), 2, A)
    fb, fs, fi, fv = bbox_utils.decode_and_nms(anchors, deltas, scores, VAR, 300, iou_threshold=0.7)
    gpu_boxes = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)
    rb, rs, _rc, rv, ri = co.combined_nms(gpu_boxes[:, :, None, :], scores[:, :, None], 300, 300, iou_threshold=0.7)
    assert np.array_equal(fv, rv) and np.array_equal(fi, ri) and np.array_equal(fb, rb)


def test_nms_empty_inputs():
    z = lambda *s: np.zeros(s, np.float32)
    out = bbox_utils.non_max_suppression(z(0, 10, 1, 4), z(0, 10, 1), max_output_size_per_class=5, max_total_size=5)
    assert out[0].shape == (0, 5, 4) and out[3].shape == (0,)
    out = bbox_utils.non_max_suppression(z(2, 0, 1, 4), z(2, 0, 1), max_output_size_per_class=5, max_total_size=5)
    assert out[0].shape == (2, 5, 4) and (out[3] == 0).all() and (out[0] == 0).all()
    out = bbox_utils.non_max_suppression(z(2, 9, 1, 4), z(2, 9, 1), max_output_size_per_class=5, max_total_size=0)
    assert out[0].shape == (2, 0, 4)
    with pytest.raises(TypeError):
        bbox_utils.non_max_suppression(z(1, 4, 1, 4), z(1, 4, 1), max_total_size=3)
    with pytest.raises(ValueError):
        bbox_utils.non_max_suppression(z(1, 4, 2, 4), z(1, 4, 3), max_output_size_per_class=1, max_total_size=1)


def test_nms_multiclass_random():
    rng = np.random.RandomState(14)
    boxes = cases.clustered_boxes(rng, 3, 700, n_clusters=25)
    scores = rng.uniform(0, 1, size=(3, 700, 21)).astype(np.float32)          # VOC: 21 labels
    _check_nms(boxes[:, :, None, :], scores, max_output_size_per_class=30, max_total_size=100, score_threshold=0.3)
    per_class = np.repeat(boxes[:, :, None, :], 21, axis=2) + rng.normal(0, 0.01, size=(3, 700, 21, 4)).astype(np.float32)
    _check_nms(per_class.astype(np.float32), scores, max_output_size_per_class=10, max_total_size=500, pad_per_class=True)


def test_nms_selection_properties_at_full_size():
    """Size-independent properties at C3 scale (B=64): sorted scores, pairwise IoU <= thr among the kept
    boxes, idempotence (NMS of its own output keeps everything)."""
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
    B, A = 64, len(anchors)
    deltas = torch.randn((B, A, 4), generator=torch.Generator().manual_seed(2)).numpy()
    boxes = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)
    scores = cases.permutation_scores(np.random.RandomState(3), B, A)
    nb, ns, _nc, nv, ni = bbox_utils.non_max_suppression(boxes[:, :, None, :], scores[:, :, None], return_indices=True,
                                                         max_output_size_per_class=300, max_total_size=300,
                                                         iou_threshold=0.7, clip_boxes=False)
    assert (nv == 300).all() and (np.diff(ns, axis=1) <= 0).all()
    assert np.array_equal(np.take_along_axis(scores, ni.astype(np.int64), axis=1), ns)
    for b in (0, 17, 63):
        kept = nb[b]
        iou = np.array([[bo.nms_iou(kept[i], kept[j]) for j in range(0, 300, 7)] for i in range(0, 300, 7)])
        np.fill_diagonal(iou, 0)
        assert iou.max() <= np.float32(0.7)
    again = bbox_utils.non_max_suppression(nb[:, :, None, :], ns[:, :, None], max_output_size_per_class=300,
                                           max_total_size=300, iou_threshold=0.7, clip_boxes=False)
    assert np.array_equal(again[0], nb) and (again[3] == 300).all()


@pytest.mark.parametrize("iou_thr", [0.5, 0.7])
def test_nms_c3_baseline_size_against_oracle(iou_thr):
    """BASELINE.json configs[2] at its full size -- B = 64 images, A = 8 649 decoded anchors, top 300 -- against the C
    oracle (not only through properties): ``rpn_combined_nms`` on the GPU-decoded boxes and the fused ``rpn_decode_nms``
    on the raw deltas, valid counts and selected indices bit-exact, at TF's default threshold 0.5 (what the reference's
    bare ``**kwargs`` pass-through gives, utils/bbox_utils.py:66-70) and at 0.7.  Inputs as bench.py's `c3` leg."""
    anchors = bo.generate_anchors(bo.get_hyper_params("vgg16"))
    B, A = 64, len(anchors)
    assert A == 8649
    deltas = np.random.RandomState(2).standard_normal((B, A, 4)).astype(np.float32)
    scores = cases.permutation_scores(np.random.RandomState(3), B, A)
    gpu_boxes = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)
    got = _check_nms(gpu_boxes[:, :, None, :], scores[:, :, None], max_output_size_per_class=300, max_total_size=300,
                     iou_threshold=iou_thr)
    assert (got[3] == 300).all()
    fb, fs, fi, fv = bbox_utils.decode_and_nms(anchors, deltas, scores, VAR, 300, iou_threshold=iou_thr)
    assert np.array_equal(fv, got[3]) and np.array_equal(fi, got[4])
    assert np.array_equal(fs, got[1]) and np.array_equal(fb, got[0])


# ---- fused decode + NMS ---------------------------------------------------------------------------
@pytest.mark.parametrize("backbone,B", [("vgg16", 8), ("mobilenet_v2", 3)])
def test_decode_nms_fused_equals_two_step(backbone, B):
    anchors = bo.generate_anchors(bo.get_hyper_params(backbone))
    A = len(anchors)
    deltas = np.random.RandomState(21).standard_normal((B, A, 4)).astype(np.float32)
    scores = cases.permutation_scores(np.random.RandomState(22), B, A)
    fb, fs, fi, fv = bbox_utils.decode_and_nms(anchors, deltas, scores, VAR, 300, iou_threshold=0.7)
    gpu_boxes = bbox_utils.get_bboxes_from_deltas(anchors, deltas, variances=VAR)
    # bit-exact against the oracle fed the SAME decoded boxes (decode itself is checked above to 2e-6)
    rb, rs, _rc, rv, ri = co.combined_nms(gpu_boxes[:, :, None, :], scores[:, :, None], 300, 300, iou_threshold=0.7)
    assert np.array_equal(fv, rv) and np.array_equal(fi, ri)
    assert np.array_equal(fb, rb) and np.array_equal(fs, rs)


# ---- preprocessing (data_utils.py:25-28): bit-exact -----------------------------------------------------------
@pytest.mark.parametrize("shape,out", [((375, 500), 500), ((500, 333), 500), ((64, 64), 64), ((1, 1), 9),
                                        ((720, 1280), 1024), ((37, 53), 32)])
def test_preprocess_image_bit_exact(shape, out):
    from tf_rpn_amd.utils import data_utils
    img = np.random.RandomState(shape[0]).randint(0, 256, size=shape + (3,)).astype(np.uint8)
    for flip in (False, True):
        got = data_utils.preprocess_image(img, out, out, flip=flip)
        assert got.dtype == np.float32 and got.shape == (out, out, 3)
        assert np.array_equal(got, bo.preprocess_image(img, out, out, flip=flip))


def test_preprocess_batch_feeds_the_model_layout():
    from tf_rpn_amd.utils import data_utils
    rng = np.random.RandomState(1)
    imgs = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in ((375, 500), (500, 375), (281, 500))]
    batch = data_utils.preprocess_batch(imgs, 500, 500, flips=[False, True, False])
    assert batch.shape == (3, 500, 500, 3) and batch.is_contiguous() and batch.dtype == torch.float32
    for i, im in enumerate(imgs):
        assert np.array_equal(batch[i].cpu().numpy(), bo.preprocess_image(im, 500, 500, flip=(i == 1)))
    g = torch.tensor([[[0.1, 0.2, 0.5, 0.6]]])
    assert np.allclose(data_utils.flip_boxes_horizontally(g).numpy(), bo.flip_boxes_horizontally(g.numpy()))


# ---- training targets (train_utils.py:84-144): labels bit-exact, deltas <= 1e-6 ----------------------------------
@pytest.mark.parametrize("backbone,B,G,n_valid", [("vgg16", 3, 42, 6), ("mobilenet_v2", 8, 42, 10), ("vgg16", 1, 1, 1),
                                                   ("vgg16", 2, 5, 0)])
def test_rpn_targets_match_oracle(backbone, B, G, n_valid):
    from tf_rpn_amd.utils import train_utils
    hp = bo.get_hyper_params(backbone)
    anchors = bo.generate_anchors(hp)
    rng = np.random.RandomState(B * 100 + G)
    gt = cases.gt_boxes(rng, B, G=G, n_valid=n_valid)
    labels = np.full((B, G), -1, np.int32)
    labels[:, :n_valid] = rng.randint(1, 21, size=(B, n_valid))
    A = len(anchors)
    rp = rng.randint(1, 1280, size=(B, A)).astype(np.int32)
    rn = rng.randint(1, 40, size=(B, A)).astype(np.int32)               # narrow range: many priority ties
    ref_d, ref_l = bo.calculate_rpn_actual_outputs(anchors, gt, labels, hp, rp, rn)
    got_d, got_l = train_utils.calculate_rpn_actual_outputs(anchors, gt, labels, hp, random_pos=rp, random_neg=rn)
    assert got_l.shape == ref_l.shape and got_d.shape == ref_d.shape
    assert np.array_equal(got_l, ref_l)
    assert np.abs(got_d - ref_d).max() <= 2e-6 * max(1.0, np.abs(ref_d).max())


def test_rpn_targets_many_positives_and_default_rng():
    """More than 128 raw positives (the random subsample is exercised) and the internal RNG path."""
    from tf_rpn_amd.utils import train_utils
    hp = bo.get_hyper_params("vgg16")
    anchors = bo.generate_anchors(hp)
    A = len(anchors)
    gt = np.zeros((2, 42, 4), np.float32)
    idx = np.random.RandomState(3).choice(A, size=40, replace=False)
    gt[:, :40] = anchors[idx]                                            # gt boxes ARE anchors: IoU 1 + duplicates
    labels = np.full((2, 42), -1, np.int32)
    labels[:, :40] = 1
    rng = np.random.RandomState(4)
    rp, rn = rng.randint(1, 1280, size=(2, A)).astype(np.int32), rng.randint(1, 2560, size=(2, A)).astype(np.int32)
    ref_d, ref_l = bo.calculate_rpn_actual_outputs(anchors, gt, labels, hp, rp, rn)
    got_d, got_l = train_utils.calculate_rpn_actual_outputs(anchors, gt, labels, hp, random_pos=rp, random_neg=rn)
    assert (ref_l.reshape(2, -1) == 1).sum(axis=1).tolist() == [128, 128]
    assert np.array_equal(got_l, ref_l) and np.abs(got_d - ref_d).max() <= 1e-5
    d2, l2 = train_utils.calculate_rpn_actual_outputs(anchors, gt, labels, hp)          # torch RNG
    l2 = l2.reshape(2, -1)
    assert ((l2 == 1).sum(axis=1) == 128).all() and ((l2 == 0).sum(axis=1) == 128).all()


@pytest.mark.gpu
def test_rpn_generator_cycles_the_dataset():
    """train_utils.py:67-82: endless (img, (deltas, labels)) stream; targets per batch as calculate_rpn_actual_outputs gives."""
    hp = bo.get_hyper_params("vgg16")
    anchors = bo.generate_anchors(hp)
    rng = np.random.RandomState(5)
    batches = []
    for b in range(2):
        gt = cases.gt_boxes(rng, 2, G=6, n_valid=4)
        labels = np.full((2, 6), -1, np.int32)
        labels[:, :4] = rng.randint(1, 21, size=(2, 4))
        batches.append((np.full((2, 4, 4, 3), float(b), np.float32), gt, labels))
    from tf_rpn_amd.utils import train_utils
    gen = train_utils.rpn_generator(batches, anchors, hp)
    out = [next(gen) for _ in range(5)]                      # 2.5 epochs
    for i, (img, (deltas, lab)) in enumerate(out):
        assert float(np.asarray(img).flat[0]) == float(i % 2)
        assert np.asarray(deltas).shape == (2, len(anchors), 4)
        assert np.asarray(lab).shape == (2, 31, 31, 9)
        assert set(np.unique(np.asarray(lab))) <= {-1.0, 0.0, 1.0}
        lab2 = np.asarray(lab).reshape(2, -1)               # per image: <= 128 positives, negatives fill up to 256
        assert ((lab2 == 1).sum(axis=1) <= 128).all() and ((lab2 == 1).sum(axis=1) + (lab2 == 0).sum(axis=1) == 256).all()


def test_nms_kernel_against_recalled_tf_unit_test_vectors():
    """The HIP CombinedNMS through the C ABI on the known-answer vectors of TensorFlow's own unit tests (recalled from
    non_max_suppression_op_test.cc, see tests/golden/tf_nms_kat.py for the provenance caveat)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("tf_nms_kat", os.path.join(os.path.dirname(__file__), "golden", "tf_nms_kat.py"))
    kat = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kat)
    from tf_rpn_amd.utils import bbox_utils
    for name, boxes, scores, max_out, iou, thr, want in kat.NMS_CASES:
        _b, _s, _c, v, idx = bbox_utils.non_max_suppression(boxes[None, :, None, :], scores[None, :, None],
                                                            max_output_size_per_class=max_out, max_total_size=max_out,
                                                            iou_threshold=iou, score_threshold=thr, clip_boxes=False,
                                                            return_indices=True)
        assert int(v[0]) == len(want) and idx[0, :len(want)].tolist() == want, name
    for name, boxes, scores, per_class, total, iou, thr, clip, wb, ws, wv in kat.COMBINED_CASES:
        b, s, c, v = bbox_utils.non_max_suppression(boxes[None, :, None, :], scores[None, :, None],
                                                    max_output_size_per_class=per_class, max_total_size=total,
                                                    iou_threshold=iou, score_threshold=thr, clip_boxes=clip)
        assert int(v[0]) == wv and np.array_equal(b[0], wb) and np.array_equal(s[0], ws) and (c[0] == 0).all(), name
