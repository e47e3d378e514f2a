"""Seeded input builders shared by the golden-fixture generator and the parity tests.
Synthetic inputs follow SURVEY.md section 8(d)."""
import numpy as np

F32 = np.float32


def random_boxes(rng, shape, lo=0.0, hi=0.7, smin=0.05, smax=0.3):
    """[y1,x1,y2,x2] with y1,x1 ~ U[lo,hi), h,w ~ U[smin,smax)."""
    y1 = rng.uniform(lo, hi, size=shape)
    x1 = rng.uniform(lo, hi, size=shape)
    h = rng.uniform(smin, smax, size=shape)
    w = rng.uniform(smin, smax, size=shape)
    return np.stack([y1, x1, y1 + h, x1 + w], axis=-1).astype(F32)


def permutation_scores(rng, B, A):
    """Tie-free scores: a seeded random permutation of (i + 0.5) / A per image."""
    base = ((np.arange(A) + 0.5) / A).astype(F32)
    return np.stack([base[rng.permutation(A)] for _ in range(B)], axis=0)


def gt_boxes(rng, B, G=42, n_valid=10):
    """VOC-like padded ground truth: first n_valid rows real, the rest zero padding
    (utils/data_utils.py:152-157)."""
    gt = np.zeros((B, G, 4), F32)
    gt[:, :n_valid] = random_boxes(rng, (B, n_valid))
    return gt


def clustered_boxes(rng, B, N, n_clusters=12, jitter=0.02):
    """Boxes in tight clusters so that NMS actually suppresses."""
    centres = random_boxes(rng, (B, n_clusters), smin=0.1, smax=0.35)
    pick = rng.randint(0, n_clusters, size=(B, N))
    boxes = np.take_along_axis(centres, pick[..., None].repeat(4, -1), axis=1)
    return (boxes + rng.normal(0, jitter, size=boxes.shape)).astype(F32)


def nms_edge_case_boxes(rng, N=64):
    """One image with every awkward box kind TF's kernel has an opinion on."""
    boxes = clustered_boxes(rng, 1, N, n_clusters=6)[0]
    boxes[3] = boxes[2]                                   # exact duplicate
    boxes[5] = boxes[4][[2, 3, 0, 1]]                     # flipped corners of box 4 (same canonical box)
    boxes[7] = [0.3, 0.3, 0.3, 0.6]                       # zero height -> area 0
    boxes[9] = [0.5, 0.5, 0.5, 0.5]                       # a point
    boxes[11] = [np.nan, 0.1, 0.4, 0.5]                   # NaN corner
    boxes[13] = [0.1, 0.1, np.inf, 0.5]                   # infinite box
    boxes[15] = [-0.2, -0.1, 0.3, 0.4]                    # sticks out (clip_boxes)
    boxes[17] = [0.8, 0.7, 1.4, 1.2]
    scores = rng.uniform(0.05, 1.0, size=(N,)).astype(F32)
    scores[20:26] = F32(0.5)                              # ties -> lower index first
    scores[2] = scores[3] = F32(0.9)                      # tied duplicates
    scores[30] = np.nan                                   # never a candidate
    scores[31] = -np.inf
    scores[32] = np.inf
    scores[33] = F32(-0.0)
    scores[34] = F32(0.0)
    return boxes.astype(F32), scores.astype(F32)
