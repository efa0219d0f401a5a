"""Oracle: IoU, anchor decode, candidate extraction and class-wise greedy NMS (numpy f32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every arithmetic step is a separate float32 numpy operation in the reference's order (no
fused multiply-add), so the HIP kernels -- compiled with ``-ffp-contract=off`` for these
paths -- can be compared bit for bit.

Follows reference ``utils.py``:
  * boxmap_anchor_relative_to_image_relative  utils.py:22-36
  * boxmap_center_relative_to_corner_relative utils.py:39-44
  * anchor_boxmap                             utils.py:47-58
  * iou                                       utils.py:62-97
  * scale_regression / regression_postprocess utils.py:100-117
  * classmap_decode                           utils.py:171-179
  * boxes_decode                              utils.py:183-195
  * nms_classwise / nms / merge_boxes_decoded utils.py:198-227
  * postprocess_and_mask (compaction order)   utils.py:258-284
"""
from collections import namedtuple

import numpy as np

from . import levels_ref, tf_ops_ref

NMS_MAX_OUTPUT_SIZE = 1000          # utils.py:9
BoxesDecoded = namedtuple("BoxesDecoded", ["boxes", "scores", "class_ids"])

f32 = np.float32


def grid_positions(h, w):
    """[H, W, 2] (y, x) cell centres, float32 (utils.py:24-31, dataset.py:16-25)."""
    ys = tf_ops_ref.cell_centers(h)
    xs = tf_ops_ref.cell_centers(w)
    return np.stack(np.meshgrid(ys, xs, indexing="ij"), -1).astype(f32)


def anchor_relative_to_image_relative(regression):
    """utils.py:22-36: add the cell centre to the (y, x) shift; sizes untouched."""
    regression = np.asarray(regression, dtype=f32)
    h, w = regression.shape[1:3]
    grid = grid_positions(h, w)[None, :, :, None, :]
    pos, size = regression[..., :2], regression[..., 2:]
    return np.concatenate([(pos + grid).astype(f32), size], -1)


def center_to_corner(regression):
    """utils.py:39-44: [cy, cx, h, w] -> [cy-h/2, cx-w/2, cy+h/2, cx+w/2]."""
    regression = np.asarray(regression, dtype=f32)
    pos = regression[..., :2]
    half = (regression[..., 2:] / f32(2)).astype(f32)
    return np.concatenate([(pos - half).astype(f32), (pos + half).astype(f32)], -1)


def anchor_boxmap(grid_size, anchor_boxes):
    """utils.py:47-58."""
    anchor_boxes = np.asarray(anchor_boxes, dtype=f32)
    a = anchor_boxes.shape[0]
    boxes = np.concatenate([np.zeros_like(anchor_boxes), anchor_boxes], -1).reshape(1, 1, 1, a, 4)
    boxes = np.tile(boxes, (1, grid_size[0], grid_size[1], 1, 1))
    return center_to_corner(anchor_relative_to_image_relative(boxes))


def iou(a, b):
    """utils.py:62-97 with numpy broadcasting; degenerate / disjoint pairs -> 0."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    assert np.all(a[..., :2] <= a[..., 2:]) and np.all(b[..., :2] <= b[..., 2:])
    y_top = np.maximum(a[..., 0], b[..., 0])
    x_left = np.maximum(a[..., 1], b[..., 1])
    y_bottom = np.minimum(a[..., 2], b[..., 2])
    x_right = np.minimum(a[..., 3], b[..., 3])
    invalid = (y_bottom < y_top) | (x_right < x_left)
    inter = ((y_bottom - y_top).astype(f32) * (x_right - x_left).astype(f32)).astype(f32)
    area_a = ((a[..., 2] - a[..., 0]).astype(f32) * (a[..., 3] - a[..., 1]).astype(f32)).astype(f32)
    area_b = ((b[..., 2] - b[..., 0]).astype(f32) * (b[..., 3] - b[..., 1]).astype(f32)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = ((area_a + area_b).astype(f32) - inter).astype(f32)
        val = (inter / denom).astype(f32)
    return np.where(invalid, f32(0), val).astype(f32)


def scale_regression(regression, anchor_boxes):
    """utils.py:100-105: multiply [.., A, 4] by [ah, aw, ah, aw]."""
    anchor_boxes = np.asarray(anchor_boxes, dtype=f32)
    tiled = np.tile(anchor_boxes, (1, 2)).reshape(1, 1, 1, anchor_boxes.shape[0], 4)
    return (np.asarray(regression, dtype=f32) * tiled).astype(f32)


def regression_postprocess(regression, anchor_boxes):
    """utils.py:108-117: exp the log-sizes, scale by the anchor, add the
    cell centre, convert to normalised corners [y1, x1, y2, x2]."""
    regression = np.asarray(regression, dtype=f32)
    shifts, scales = regression[..., :2], regression[..., 2:]
    regression = np.concatenate([shifts, np.exp(scales).astype(f32)], -1)
    regression = scale_regression(regression, anchor_boxes)
    regression = anchor_relative_to_image_relative(regression)
    return center_to_corner(regression)


def classmap_decode(classmap):
    """utils.py:171-179 -> fg mask."""
    return np.asarray(classmap).max(-1) > 0.5


def boxes_decode(classifications, regressions):
    """utils.py:183-195: max/argmax over classes (first index on ties [TF-sem]),
    keep max > 0.5, compaction in row-major order."""
    classifications = np.asarray(classifications, dtype=f32)
    cmax = classifications.max(-1)
    ids = classifications.argmax(-1).astype(np.int64)
    fg = cmax > f32(0.5)
    return BoxesDecoded(np.asarray(regressions, dtype=f32)[fg], cmax[fg], ids[fg])


def merge_boxes_decoded(items):
    """utils.py:223-227."""
    return BoxesDecoded(np.concatenate([d.boxes for d in items], 0),
                        np.concatenate([d.scores for d in items], 0),
                        np.concatenate([d.class_ids for d in items], 0))


def _nms_iou(bi, bj):
    """[TF-sem] IoU as TF's NonMaxSuppression kernel computes it: corners are
    min/max-normalised, non-positive areas give 0, intersection extents clamp at 0."""
    ymin_i, ymax_i = min(bi[0], bi[2]), max(bi[0], bi[2])
    xmin_i, xmax_i = min(bi[1], bi[3]), max(bi[1], bi[3])
    ymin_j, ymax_j = min(bj[0], bj[2]), max(bj[0], bj[2])
    xmin_j, xmax_j = min(bj[1], bj[3]), max(bj[1], bj[3])
    area_i = f32(f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i))
    area_j = f32(f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return f32(0)
    iy = max(f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f32(0))
    ix = max(f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f32(0))
    inter = f32(iy * ix)
    return f32(inter / f32(f32(area_i + area_j) - inter))


def nms_indices(boxes, scores, max_output_size=NMS_MAX_OUTPUT_SIZE, iou_threshold=0.5):
    """[TF-sem] tf.image.non_max_suppression: visit candidates by descending score (ties:
    lower index first), keep one unless its IoU with an already kept box is > threshold,
    stop at max_output_size."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    thr = f32(iou_threshold)
    keep = []
    for idx in order:
        if len(keep) >= max_output_size:
            break
        ok = True
        for k in reversed(keep):
            if _nms_iou(boxes[idx], boxes[k]) > thr:
                ok = False
                break
        if ok:
            keep.append(int(idx))
    return np.asarray(keep, dtype=np.int64)


def nms_indices_vectorised(boxes, scores, max_output_size=NMS_MAX_OUTPUT_SIZE, iou_threshold=0.5):
    """Same result as nms_indices, with the inner loop vectorised (used for the larger
    parity cases and as the timed CPU baseline)."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    b = boxes[order]
    ymin = np.minimum(b[:, 0], b[:, 2]); ymax = np.maximum(b[:, 0], b[:, 2])
    xmin = np.minimum(b[:, 1], b[:, 3]); xmax = np.maximum(b[:, 1], b[:, 3])
    area = ((ymax - ymin).astype(f32) * (xmax - xmin).astype(f32)).astype(f32)
    thr = f32(iou_threshold)
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        if len(keep) >= max_output_size:
            break
        r = slice(i + 1, n)
        iy = np.maximum((np.minimum(ymax[i], ymax[r]) - np.maximum(ymin[i], ymin[r])).astype(f32), f32(0))
        ix = np.maximum((np.minimum(xmax[i], xmax[r]) - np.maximum(xmin[i], xmin[r])).astype(f32), f32(0))
        inter = (iy * ix).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            val = (inter / ((area[i] + area[r]).astype(f32) - inter).astype(f32)).astype(f32)
        val = np.where((area[i] <= 0) | (area[r] <= 0), f32(0), val)
        alive[r] &= ~(val > thr)
    return order[np.asarray(keep, dtype=np.int64)]


def nms(decoded, max_output_size=NMS_MAX_OUTPUT_SIZE, fast=True):
    """utils.py:213-220."""
    fn = nms_indices_vectorised if fast else nms_indices
    idx = fn(decoded.boxes, decoded.scores, max_output_size)
    return BoxesDecoded(decoded.boxes[idx], decoded.scores[idx], decoded.class_ids[idx])


def nms_classwise(decoded, num_classes, fast=True):
    """utils.py:198-210: per class mask -> nms -> concat class-major."""
    parts = []
    for c in range(num_classes):
        m = decoded.class_ids == c
        parts.append(nms(BoxesDecoded(decoded.boxes[m], decoded.scores[m], decoded.class_ids[m]), fast=fast))
    return merge_boxes_decoded(parts)


def detect_image(class_probs, regressions, image_size, num_classes, anchor_mode="trunc_int"):
    """train.py:68-85 composition for ONE image: per level boxes_decode of the decoded
    regressions, merge P3..P7, class-wise NMS.  `class_probs` / `regressions` are dicts
    P3..P7 of [H, W, A, C] / [H, W, A, 4] (raw regressions)."""
    pyr = levels_ref.pyramid()
    parts = []
    for k in pyr:
        anchors = levels_ref.normalized_anchor_sizes(pyr[k], image_size, anchor_mode)
        dec = regression_postprocess(np.asarray(regressions[k], dtype=f32)[None], anchors)[0]
        parts.append(boxes_decode(class_probs[k], dec))
    return nms_classwise(merge_boxes_decoded(parts), num_classes)


def compact_trainable(per_level, masks):
    """utils.py:270-278: boolean_mask each level by its trainable mask, concat P3..P7."""
    return np.concatenate([np.asarray(per_level[k])[np.asarray(masks[k], dtype=bool)] for k in per_level], 0)
