"""Evaluation the reference only sketches: COCO-style mean average precision over the detections
of `utils.detect`, and the two IoU metrics of the reference's `build_metrics` (train.py:137-161): `class_iou`
(tf.metrics.mean_iou, 2 classes, of the thresholded class probabilities) and `regr_iou` (mean IoU between the
decoded ground-truth and predicted boxes at the foreground anchors).  Host-side numpy on copies of the device
results: evaluation is not on the training hot path."""
import numpy as np


def iou_matrix(a, b):
    """IoU of corner boxes a [K,4] x b [O,4] ([y1,x1,y2,x2]); empty / inverted intersections count 0 (utils.py:62-97)."""
    a = np.asarray(a, np.float64).reshape(-1, 1, 4)
    b = np.asarray(b, np.float64).reshape(1, -1, 4)
    tl = np.maximum(a[..., :2], b[..., :2])
    br = np.minimum(a[..., 2:], b[..., 2:])
    inter = np.prod(np.maximum(br - tl, 0.0), -1)
    area_a = np.prod(np.maximum(a[..., 2:] - a[..., :2], 0.0), -1)
    area_b = np.prod(np.maximum(b[..., 2:] - b[..., :2], 0.0), -1)
    union = area_a + area_b - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)


def average_precision(tp, scores, num_gt):
    """COCO 101-point interpolated AP of one (class, IoU threshold): tp flags of the detections, their scores."""
    if num_gt == 0:
        return np.nan
    order = np.argsort(-np.asarray(scores, np.float64), kind='mergesort')
    tp = np.asarray(tp, np.float64)[order]
    ctp, cfp = np.cumsum(tp), np.cumsum(1.0 - tp)
    recall = ctp / num_gt
    precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
    for i in range(len(precision) - 1, 0, -1):                  # precision envelope
        precision[i - 1] = max(precision[i - 1], precision[i])
    points = np.linspace(0.0, 1.0, 101)
    idx = np.searchsorted(recall, points, side='left')
    return float(np.mean([precision[i] if i < len(precision) else 0.0 for i in idx]))


def mean_average_precision(detections, ground_truth, num_classes, iou_thresholds=None):
    """detections: per image (boxes [K,4], scores [K], class_ids [K]); ground_truth: per image (boxes [O,4], class_ids [O]).
    Returns {'mAP' (IoU 0.50:0.95), 'AP50', 'AP75', 'per_class' [C] (mean over thresholds; nan = class absent)}."""
    thr = np.arange(0.5, 0.96, 0.05) if iou_thresholds is None else np.asarray(iou_thresholds, np.float64)
    ap = np.full((num_classes, len(thr)), np.nan)
    for c in range(num_classes):
        per_image, num_gt = [], 0
        for (db, ds, dc), (gb, gc) in zip(detections, ground_truth):
            dsel = np.asarray(dc) == c
            gsel = np.asarray(gc) == c
            num_gt += int(gsel.sum())
            per_image.append((np.asarray(db, np.float64).reshape(-1, 4)[dsel], np.asarray(ds, np.float64)[dsel],
                              np.asarray(gb, np.float64).reshape(-1, 4)[gsel]))
        if num_gt == 0:
            continue
        for t, th in enumerate(thr):
            tps, scs = [], []
            for db, ds, gb in per_image:
                order = np.argsort(-ds, kind='mergesort')
                ious = iou_matrix(db[order], gb) if len(gb) and len(db) else np.zeros((len(db), len(gb)))
                taken = np.zeros(len(gb), bool)
                for i in range(len(order)):
                    best, bj = th, -1
                    for j in range(len(gb)):                     # best still-free ground truth at or above the threshold
                        if not taken[j] and ious[i, j] >= best:
                            best, bj = ious[i, j], j
                    if bj >= 0:
                        taken[bj] = True
                    tps.append(1.0 if bj >= 0 else 0.0)
                    scs.append(ds[order][i])
            ap[c, t] = average_precision(tps, scs, num_gt) if len(tps) else 0.0
    def col(v):
        k = int(np.argmin(np.abs(thr - v)))
        return float(np.nanmean(ap[:, k])) if abs(thr[k] - v) < 1e-9 and np.any(~np.isnan(ap[:, k])) else float('nan')
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)      # classes without ground truth stay nan
        per_class = np.nanmean(ap, 1)
    return {'mAP': float(np.nanmean(ap)) if np.any(~np.isnan(ap)) else float('nan'), 'AP50': col(0.5), 'AP75': col(0.75),
            'per_class': per_class}


def detections_from_detect(out, num_images):
    """(boxes, scores, class_ids, image_ids, ...) of utils.detect -> per-image tuples for mean_average_precision."""
    boxes, scores, cls, img = (np.asarray(t.detach().cpu().numpy() if hasattr(t, 'detach') else t) for t in out[:4])
    return [(boxes[img == i], scores[img == i], cls[img == i]) for i in range(num_images)]


def class_iou(labels, probs, threshold=0.5):
    """tf.metrics.mean_iou(labels, sigmoid(logits) > 0.5, num_classes=2) over the trainable anchors (train.py:149-152):
    mean over the classes {0, 1} present of TP / (TP + FP + FN)."""
    l = np.asarray(labels).reshape(-1) > 0.5
    p = np.asarray(probs).reshape(-1) > threshold
    ious = []
    for v in (False, True):
        inter = np.sum((l == v) & (p == v))
        union = np.sum((l == v) | (p == v))
        if union > 0:
            ious.append(inter / union)
    return float(np.mean(ious)) if ious else float('nan')


def regr_iou(true_boxes, pred_boxes):
    """Mean IoU of matching rows (train.py:138-145,153-154: both decoded at the ground-truth foreground anchors)."""
    a = np.asarray(true_boxes, np.float64).reshape(-1, 4)
    b = np.asarray(pred_boxes, np.float64).reshape(-1, 4)
    if len(a) == 0:
        return float('nan')
    tl = np.maximum(a[:, :2], b[:, :2])
    br = np.minimum(a[:, 2:], b[:, 2:])
    inter = np.prod(np.maximum(br - tl, 0.0), -1)
    union = np.prod(np.maximum(a[:, 2:] - a[:, :2], 0), -1) + np.prod(np.maximum(b[:, 2:] - b[:, :2], 0), -1) - inter
    return float(np.mean(np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)))
