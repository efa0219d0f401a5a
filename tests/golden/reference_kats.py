"""Known-answer vectors transcribed (as data) from the reference's own tests.

Each entry cites the reference test it comes from.  Inputs/expected outputs only -- no
reference source text.  Stale tests (written against earlier signatures)
are transcribed in the form that still describes the current code.
"""
import numpy as np

# levels_test.py:5-9 -- compute_box_size(32, (1, 2), 1): len 2, area 32^2, w/h == 2
BOX_SIZE_CASE = dict(base=32, aspect=(1, 2), scale=1, area=32 ** 2, ratio=2)
# levels_test.py:12-14 -- Level(32, [(1, 4)], [1, 2]).anchor_sizes
LEVEL_CASE = dict(base=32, aspects=[(1, 4)], scales=[2 ** 0, 2 ** 1],
                  expected=np.array([[16, 64], [32, 128]], dtype=np.float64))

# utils_test.py:7-42 -- boxmap_anchor_relative_to_image_relative on a 3x4 grid, one anchor
_c = [0.5, 1.0, 0.25, 0.75]
ANCHOR_REL_INPUT = np.tile(np.array(_c, dtype=np.float32), (1, 3, 4, 1, 1))
ANCHOR_REL_EXPECTED = np.array(
    [[[[(2 * i + 1) / 6 + 0.5, (2 * j + 1) / 8 + 1.0, 0.25, 0.75]] for j in range(4)] for i in range(3)],
    dtype=np.float64)[None]

# utils_test.py:44-74 -- anchor_boxmap(grid (3,4), anchors [[0.2, 0.4]])
ANCHOR_BOXMAP_GRID = (3, 4)
ANCHOR_BOXMAP_ANCHORS = np.array([[0.2, 0.4]], dtype=np.float32)
ANCHOR_BOXMAP_EXPECTED = np.array(
    [[[[(2 * i + 1) / 6 - 0.1, (2 * j + 1) / 8 - 0.2, (2 * i + 1) / 6 + 0.1, (2 * j + 1) / 8 + 0.2]]
      for j in range(4)] for i in range(3)], dtype=np.float64)[None]

# utils_test.py:76-97 -- centre -> corner
CENTER_CORNER_INPUT = np.tile(np.array([0.5, 1.0, 0.2, 0.4], dtype=np.float32), (1, 3, 4, 1, 1))
CENTER_CORNER_EXPECTED = np.tile(np.array([0.4, 0.8, 0.6, 1.2], dtype=np.float32), (1, 3, 4, 1, 1))

# utils_test.py:99-118 -- iou, incl. disjoint and degenerate pairs
IOU_A = np.array([[0.1, 0.1, 0.2, 0.2], [100, 100, 200, 200], [0.1, 0.1, 0.2, 0.2], [1., 1., 1., 1.]],
                 dtype=np.float32)
IOU_B = np.array([[0.1, 0.1, 0.3, 0.3], [100, 100, 300, 300], [100, 100, 300, 300], [0., 0., 0., 0.]],
                 dtype=np.float32)
IOU_EXPECTED = np.array([0.25, 0.25, 0, 0], dtype=np.float32)

# utils_test.py:120-138 -- classmap_decode; the test predates the ClassmapDecoded(fg_mask)
# return type (utils.py:179): [1, 1, -1, -1] there == fg [T, T, F, F] now.
CLASSMAP = np.array([[0.1, 0.9, 0.3, 0.8], [0, 1, 0, 0], [0.1, 0.2, 0.4, 0.3], [0, 0, 0, 0]],
                    dtype=np.float32)
CLASSMAP_FG_EXPECTED = np.array([True, True, False, False])

# retinanet_old_test.py:15-37 -- scale_regression
SCALE_REGR_INPUT = np.array([[0.5, 1.0, 0.5, 1.0], [0.5, 0.5, 0.5, 0.5]], dtype=np.float32).reshape(1, 1, 1, 2, 4)
SCALE_REGR_ANCHORS = np.array([[0.2, 0.4], [0.4, 0.2]], dtype=np.float32)
SCALE_REGR_EXPECTED = np.array([[0.1, 0.4, 0.1, 0.4], [0.2, 0.1, 0.2, 0.1]], dtype=np.float32).reshape(1, 1, 1, 2, 4)

# losses_test.py:17-27 -- regression_loss == 2.0
HUBER_LOGITS = np.array([[1.], [2.], [3.]], dtype=np.float32)
HUBER_LABELS = np.array([[3.], [4.], [6.]], dtype=np.float32)
HUBER_FG = np.array([True, False, True])
HUBER_EXPECTED = 2.0

# losses_test.py:7-15 -- "mask then concat over dict order" (now utils.py:270-278)
MERGE_OUTPUTS = {"a": np.array([1, 2, 3]), "b": np.array([4, 5, 6])}
MERGE_MASKS = {"a": np.array([False, True, True]), "b": np.array([True, True, False])}
MERGE_EXPECTED = np.array([2, 3, 4, 5])

# dataset_test.py:8-45 -- level_labels on a 32x32 image, Level(16, [(1,1)], [1, 1.5]),
# factor 16.  Class part only (the test predates one-hot classes and log-space sizes):
# expected class id per (cell, anchor), 0 == background (all-zero one-hot row).
ASSIGN_IMAGE_SIZE = (32, 32)
ASSIGN_CLASS_IDS = np.array([100, 200, 300, 400])
ASSIGN_BOXES = np.array([[0, 0, 16, 16], [8, 8, 24, 24], [16, 16, 32, 32], [-4, -4, 20, 20]],
                        dtype=np.float64) / 32.0
ASSIGN_LEVEL = dict(base=16, aspects=[(1, 1)], scales=[1, 1.5])
ASSIGN_FACTOR = 16
ASSIGN_CLASSMAP_EXPECTED = np.array([[[100, 400], [0, 0]], [[0, 0], [300, 0]]])
# same test, regression part re-expressed in the CURRENT format (shift/anchor, log(size/anchor)):
# the three assigned anchors coincide with their object => all four targets are 0.
ASSIGN_MATCHED_REGRESSION = np.zeros(4, dtype=np.float32)

# augmentation_test.py:7-45 -- flip reverses W and negates the x shift (component 1)
FLIP_REGR_INPUT = np.arange(2 * 3 * 1 * 4, dtype=np.float32).reshape(2, 3, 1, 4)

# retinanet_test.py:7-69 / mobilenet_v2.py:226-233 -- output shape contracts
def pyramid_sizes(s):
    out, c = [], s
    for i in range(7):
        c = -(-c // 2)
        if i >= 2:
            out.append(c)
    return out          # P3..P7


# ------------------------------------------------------------------------------------------------------------------
# Known-answer vectors from TENSORFLOW's own published unit tests, for the ops the reference calls but holds no test
# of (SURVEY section 8c "[TF-sem]" rows).  Transcribed from memory of the TF 1.x test sources (TF is
# not installable here); each vector is also self-evidently consistent with the op's documented definition, which the
# comments spell out.  Data only.
#
# tensorflow/core/kernels/non_max_suppression_op_test.cc, NonMaxSuppressionOpTest (and, first case, the same data in
# tensorflow/python/ops/image_ops_test.py NonMaxSuppressionTest.testSelectFromThreeClusters) -- the op behind
# reference utils.py:213-220.  Boxes [y1, x1, y2, x2]; three clusters at x ~ 0, ~ 10, ~ 100; IoU threshold 0.5.
TF_NMS_THREE_CLUSTERS_BOXES = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9],
                                        [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], dtype=np.float32)
TF_NMS_THREE_CLUSTERS_SCORES = np.array([0.9, 0.75, 0.6, 0.95, 0.5, 0.3], dtype=np.float32)
# TestSelectFromThreeClustersFlippedCoordinates: the same boxes with some corner pairs swapped (the kernel min/max-normalises)
TF_NMS_FLIPPED_BOXES = np.array([[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, 0.9, 1, -0.1],
                                 [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]], dtype=np.float32)
TF_NMS_CASES = [
    # (TF test name, boxes, scores, max_output_size, expected selected indices)
    ("TestSelectFromThreeClusters", TF_NMS_THREE_CLUSTERS_BOXES, TF_NMS_THREE_CLUSTERS_SCORES, 3, [3, 0, 5]),
    ("TestSelectFromThreeClustersFlippedCoordinates", TF_NMS_FLIPPED_BOXES, TF_NMS_THREE_CLUSTERS_SCORES, 3, [3, 0, 5]),
    ("TestSelectAtMostTwoBoxesFromThreeClusters", TF_NMS_THREE_CLUSTERS_BOXES, TF_NMS_THREE_CLUSTERS_SCORES, 2, [3, 0]),
    ("TestSelectAtMostThirtyBoxesFromThreeClusters", TF_NMS_THREE_CLUSTERS_BOXES, TF_NMS_THREE_CLUSTERS_SCORES, 30, [3, 0, 5]),
    ("TestSelectWithNegativeScores", TF_NMS_THREE_CLUSTERS_BOXES, TF_NMS_THREE_CLUSTERS_SCORES - np.float32(10.0), 6, [3, 0, 5]),
    ("TestSelectSingleBox", np.array([[0, 0, 1, 1]], dtype=np.float32), np.array([0.9], dtype=np.float32), 3, [0]),
    ("TestSelectFromTenIdenticalBoxes", np.tile(np.array([[0, 0, 1, 1]], dtype=np.float32), (10, 1)),
     np.full(10, 0.9, dtype=np.float32), 3, [0]),
    ("TestEmptyInput", np.zeros((0, 4), dtype=np.float32), np.zeros((0,), dtype=np.float32), 30, []),
]

# tensorflow/python/ops/image_ops_test.py, ResizeImagesTest.testResizeUpAlignCornersTrue -- tf.image.resize_images(...,
# align_corners=True), the op behind reference retinanet.py:153-155 (NEAREST_NEIGHBOR, Q12) and dataset.py:145-151 (BILINEAR):
# a [1, 3, 2, 1] image resized to 5 x 4.  Source rows 0, 0.5, 1, 1.5, 2: the nearest-neighbour case pins the half-way
# rounding (0.5 -> 1, 1.5 -> 2: roundf, not half-to-even); source columns 0, 1/3, 2/3, 1.
TF_RESIZE_ALIGN_CORNERS_INPUT = np.array([6, 3, 3, 6, 6, 9], dtype=np.float32).reshape(1, 3, 2, 1)
TF_RESIZE_ALIGN_CORNERS_SIZE = (5, 4)
TF_RESIZE_ALIGN_CORNERS_NEAREST = np.array([6.0, 6.0, 3.0, 3.0, 3.0, 3.0, 6.0, 6.0, 3.0, 3.0, 6.0, 6.0, 6.0, 6.0,
                                            9.0, 9.0, 6.0, 6.0, 9.0, 9.0], dtype=np.float32).reshape(1, 5, 4, 1)
TF_RESIZE_ALIGN_CORNERS_BILINEAR = np.array([6.0, 5.0, 4.0, 3.0, 4.5, 4.5, 4.5, 4.5, 3.0, 4.0, 5.0, 6.0, 4.5, 5.5,
                                             6.5, 7.5, 6.0, 7.0, 8.0, 9.0], dtype=np.float32).reshape(1, 5, 4, 1)

# tensorflow/python/kernel_tests/losses_test.py, HuberLossTest (delta = 1, the reference's losses.py:146): mean over the elements
# of 0.5 e^2 for |e| <= 1, |e| - 0.5 above.
TF_HUBER_PREDICTIONS = np.array([1.5, -1.4, -1.0, 0.0], dtype=np.float32)
TF_HUBER_CASES = [
    # (TF test name, labels, predictions, expected loss)
    ("testAllQuadratic", np.array([1.0, -1.0, 0.0, 0.5], dtype=np.float32), TF_HUBER_PREDICTIONS,
     0.5 * (0.25 + 0.16 + 1.0 + 0.25) / 4.0),
    ("testAllLinear", np.array([0.0, 1.0, 0.0, 1.5], dtype=np.float32), TF_HUBER_PREDICTIONS,
     (1.5 + 2.4 + 1.0 + 1.5) / 4.0 - 0.5),
    ("testMixedQuadraticLinear", np.array([[1.0, -1.0, 0.0, 0.5], [0.0, 1.0, 0.0, 1.5]], dtype=np.float32),
     np.stack([TF_HUBER_PREDICTIONS, TF_HUBER_PREDICTIONS]),
     (0.5 * (0.25 + 0.16 + 1.0 + 0.25) / 4.0 + (1.5 + 2.4 + 1.0 + 1.5) / 4.0 - 0.5) / 2.0),
]

# tensorflow/python/kernel_tests/losses_test.py, SigmoidCrossEntropyLossTest (mean of tf.nn.sigmoid_cross_entropy_with_logits,
# the op of reference losses.py:124): saturated logits of +-100 -- a naive log(sigmoid) overflows, the stable form
# max(x, 0) - x z + log(1 + exp(-|x|)) gives 0 per right entry and 100 per wrong one.
TF_BCE_LOGITS = np.array([[100.0, -100.0, -100.0], [-100.0, 100.0, -100.0], [-100.0, -100.0, 100.0]], dtype=np.float32)
TF_BCE_CASES = [
    ("testAllCorrectSigmoid", np.eye(3, dtype=np.float32), 0.0),
    ("testAllWrongSigmoid", np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.float32), 600.0 / 9.0),
]
