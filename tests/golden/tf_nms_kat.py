"""Known-answer vectors of TensorFlow's NMS kernels, RECALLED from TensorFlow's own published unit tests
(tensorflow/core/kernels/non_max_suppression_op_test.cc: NonMaxSuppressionOpTest / CombinedNonMaxSuppressionOpTest) --
the third-party dependency the reference calls at utils/bbox_utils.py:66 (tensorflow==2.0.0, environment.yml:49-52).

Provenance, stated plainly: TensorFlow is not in this image and there is no network, so these values were written down
from memory of that test file, not generated or verified here.  They are data (inputs and expected outputs), used as
an additional, independent pin of the restatement's SEMANTICS: descending-score visiting order, strict `>` on the IoU
threshold, corner canonicalisation (flipped boxes), the first of several identical equal-score boxes wins, the score
threshold, zero padding past valid_detections and clip_boxes.  They do not lift the "parity unpinned" label: a vector
recalled wrongly would only fail the tests, a vector recalled rightly still is not an output of the reference run here.
"""
import numpy as np

F = np.float32

# ---- NonMaxSuppressionOpTest (single image, single class; expected = selected indices) --------------------------------
THREE_CLUSTERS = F([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]])
THREE_CLUSTERS_FLIPPED = F([[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, 0.9, 1, -0.1], [0, 10, 1, 11], [1, 10.1, 0, 11.1],
                            [1, 101, 0, 100]])
SCORES = F([0.9, 0.75, 0.6, 0.95, 0.5, 0.3])

NMS_CASES = [
    # name, boxes, scores, max_output_size, iou_threshold, score_threshold, expected indices
    ("TestSelectFromThreeClusters", THREE_CLUSTERS, SCORES, 3, 0.5, float("-inf"), [3, 0, 5]),
    ("TestSelectFromThreeClustersFlippedCoordinates", THREE_CLUSTERS_FLIPPED, SCORES, 3, 0.5, float("-inf"), [3, 0, 5]),
    ("TestSelectAtMostTwoBoxesFromThreeClusters", THREE_CLUSTERS, SCORES, 2, 0.5, float("-inf"), [3, 0]),
    ("TestSelectWithNegativeScores", THREE_CLUSTERS, SCORES - F(10.0), 6, 0.5, float("-inf"), [3, 0, 5]),
    ("TestSelectAtMostThirtyBoxesFromThreeClusters", THREE_CLUSTERS, SCORES, 30, 0.5, float("-inf"), [3, 0, 5]),
    ("TestSelectSingleBox", F([[0, 0, 1, 1]]), F([0.9]), 3, 0.5, float("-inf"), [0]),
    ("TestSelectFromTenIdenticalBoxes", np.tile(F([[0, 0, 1, 1]]), (10, 1)), np.full((10,), 0.9, F), 3, 0.5, float("-inf"), [0]),
    ("TestSelectFromThreeClustersWithScoreThreshold (V3)", THREE_CLUSTERS, SCORES, 3, 0.5, 0.4, [3, 0]),
]

# ---- CombinedNonMaxSuppressionOpTest (boxes (1,6,1,4), scores (1,6,1)) -------------------------------------------------
COMBINED_BOXES = F([[0, 0, 0.1, 0.1], [0, 0.01, 0.1, 0.11], [0, -0.01, 0.1, 0.09], [0, 0.11, 0.1, 0.2], [0, 0.12, 0.1, 0.21],
                    [0, 0.3, 1, 0.4]])
COMBINED_BOXES_BIG = F([[0, 0, 10, 10], [0, 1, 10, 11], [0, 1, 10, 9], [0, 11, 10, 20], [0, 12, 10, 21], [0, 30, 100, 40]])

COMBINED_CASES = [
    # name, boxes, scores, max_per_class, max_total, iou, score_threshold, clip_boxes, expected boxes / scores / valid
    ("TestSelectFromThreeClusters", COMBINED_BOXES, SCORES, 3, 3, 0.5, 0.0, True,
     F([[0, 0.11, 0.1, 0.2], [0, 0, 0.1, 0.1], [0, 0.3, 1, 0.4]]), F([0.95, 0.9, 0.3]), 3),
    ("TestSelectFromThreeClustersNoBoxClipping", COMBINED_BOXES_BIG, SCORES, 3, 3, 0.5, 0.0, False,
     F([[0, 11, 10, 20], [0, 0, 10, 10], [0, 30, 100, 40]]), F([0.95, 0.9, 0.3]), 3),
    ("TestSelectFromThreeClustersWithScoreThreshold", COMBINED_BOXES, SCORES, 3, 3, 0.5, 0.4, True,
     F([[0, 0.11, 0.1, 0.2], [0, 0, 0.1, 0.1], [0, 0, 0, 0]]), F([0.95, 0.9, 0.0]), 2),
]
