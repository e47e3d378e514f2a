"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/bbox_oracle.py, oracle/conv_oracle.py).

The fixtures are RESTATEMENT-GENERATED, not TF-generated: the reference cannot be imported here
(TensorFlow is absent) and ships no golden vectors of its own.  They pin
the oracle against silent drift and travel to the GPU box, where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import bbox_oracle as bo  # noqa: E402


def main():
    out = {}
    # ---- anchors: the two default configs, the 1024^2 / 15-anchor config (C5), and a tiny grid
    hp_small = bo.get_hyper_params("vgg16", feature_map_shape=4)
    out["anchors_small_f4"] = bo.generate_anchors(hp_small)
    out["base_anchors_default"] = bo.generate_base_anchors(bo.get_hyper_params("vgg16"))
    out["anchors_vgg16"] = bo.generate_anchors(bo.get_hyper_params("vgg16"))
    out["anchors_mobilenet_v2"] = bo.generate_anchors(bo.get_hyper_params("mobilenet_v2"))
    hp_c5 = bo.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                anchor_ratios=[1., 2., 1. / 2., 3., 1. / 3.])
    a5 = bo.generate_anchors(hp_c5)
    out["anchors_c5_head"] = a5[:600]                       # first 40 cells
    out["anchors_c5_sum"] = np.array([a5.astype(np.float64).sum(), float(len(a5))])
    np.savez_compressed(os.path.join(HERE, "anchors.npz"), **out)

    # ---- decode / encode / IoU map
    rng = np.random.RandomState(2)
    anchors = out["anchors_small_f4"]
    deltas = rng.standard_normal((3, len(anchors), 4)).astype(np.float32)
    variances = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    scaled = bo.scale_deltas(deltas, variances)
    boxes = bo.get_bboxes_from_deltas(anchors, scaled)
    gt = cases.gt_boxes(np.random.RandomState(4), 3, G=7, n_valid=4)
    gt_per_anchor = cases.random_boxes(np.random.RandomState(5), (3, len(anchors)))
    gt_per_anchor[0, 5] = 0.0                               # zero-size gt row (padding) -> zero deltas
    np.savez_compressed(os.path.join(HERE, "boxmath.npz"), anchors=anchors, deltas=deltas, variances=variances,
                        boxes=boxes, gt=gt, iou_map=bo.generate_iou_map(anchors, gt),
                        gt_per_anchor=gt_per_anchor, encoded=bo.get_deltas_from_bboxes(anchors, gt_per_anchor))

    # ---- NMS
    nms = {}
    rng = np.random.RandomState(7)
    eb, es = cases.nms_edge_case_boxes(rng)
    nms["edge_boxes"], nms["edge_scores"] = eb, es
    for name, kw in {
        "edge_default": dict(max_output_size_per_class=20, max_total_size=20),
        "edge_thr07_noclip": dict(max_output_size_per_class=64, max_total_size=64, iou_threshold=0.7, clip_boxes=False),
        "edge_score_thr": dict(max_output_size_per_class=64, max_total_size=10, score_threshold=0.5),
    }.items():
        r = bo.combined_non_max_suppression(eb[None, :, None, :], es[None, :, None], return_indices=True, **kw)
        for key, arr in zip(("boxes", "scores", "classes", "valid", "idx"), r):
            nms["%s_%s" % (name, key)] = arr
    # multi-class, shared boxes (q=1) and per-class boxes (q=C)
    mb = cases.clustered_boxes(rng, 2, 48, n_clusters=5)
    ms = rng.uniform(0, 1, size=(2, 48, 3)).astype(np.float32)
    mbq = np.stack([mb, mb + np.float32(0.01), cases.clustered_boxes(rng, 2, 48, n_clusters=4)], axis=2)
    nms["mc_boxes"], nms["mc_scores"], nms["mc_boxes_q"] = mb, ms, mbq
    for name, bx, kw in (
        ("mc_q1", mb[:, :, None, :], dict(max_output_size_per_class=5, max_total_size=12)),
        ("mc_qc", mbq, dict(max_output_size_per_class=6, max_total_size=40, pad_per_class=True, iou_threshold=0.3)),
    ):
        r = bo.combined_non_max_suppression(bx, ms, return_indices=True, **kw)
        for key, arr in zip(("boxes", "scores", "classes", "valid", "idx"), r):
            nms["%s_%s" % (name, key)] = arr
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **nms)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
