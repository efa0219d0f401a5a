"""Oracle for metrics.py: an independent, deliberately naive restatement of COCO-style AP (sort all detections of a
class, greedy best-IoU matching per image, 101-point interpolation) and of tf.metrics.mean_iou for two classes.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference has no mAP code and its
build_metrics (train.py:137-161) is never exercised by its tests; the known answers in tests/test_host_cpu.py
(perfect / half-missed / duplicate detections) pin the definition instead."""
import numpy as np


def iou(a, b):
    ih = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    iw = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = ih * iw
    ua = max(0.0, a[2] - a[0]) * max(0.0, a[3] - a[1]) + max(0.0, b[2] - b[0]) * max(0.0, b[3] - b[1]) - inter
    return inter / ua if ua > 0 else 0.0


def ap_one(dets, gts_per_image, thr):
    """dets: list of (image, score, box); gts_per_image: {image: [boxes]}."""
    ngt = sum(len(v) for v in gts_per_image.values())
    if ngt == 0:
        return float('nan')
    used = {k: [False] * len(v) for k, v in gts_per_image.items()}
    flags = []
    for img, score, box in sorted(dets, key=lambda d: -d[1]):
        best, bj = thr, -1
        for j, g in enumerate(gts_per_image.get(img, [])):
            v = iou(box, g)
            if not used[img][j] and v >= best:
                best, bj = v, j
        if bj >= 0:
            used[img][bj] = True
        flags.append(bj >= 0)
    tp = fp = 0
    rec, prec = [], []
    for f in flags:
        tp += f
        fp += not f
        rec.append(tp / ngt)
        prec.append(tp / (tp + fp))
    for i in range(len(prec) - 2, -1, -1):
        prec[i] = max(prec[i], prec[i + 1])
    total = 0.0
    for r in np.linspace(0, 1, 101):
        p = 0.0
        for rr, pp in zip(rec, prec):
            if rr >= r:
                p = pp
                break
        total += p
    return total / 101


def mean_ap(detections, ground_truth, num_classes, thresholds):
    vals = []
    for c in range(num_classes):
        gts = {i: [list(b) for b, cc in zip(g[0], g[1]) if cc == c] for i, g in enumerate(ground_truth)}
        dets = [(i, float(s), list(b)) for i, d in enumerate(detections) for b, s, cc in zip(d[0], d[1], d[2]) if cc == c]
        for t in thresholds:
            v = ap_one(dets, gts, t)
            if not np.isnan(v):
                vals.append(v)
    return float(np.mean(vals)) if vals else float('nan')
