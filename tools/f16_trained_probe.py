"""fp16 inference judged on a TRAINED net: ResNeXt-50-FPN trained on the shapes stream at 256^2 with the
product's loop, then -- on held-out images -- the fp16 product's detections against the fp32 product's: survivors shared, score
and box differences, mAP of both.  (The test adds the CPU oracle: tests/test_gpu_fullsize.py.)
    python tools/f16_trained_probe.py [steps] [lr]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch


def detections(net, image, lv, classes, f16):
    import layers, ops, utils
    size = (int(image.shape[1]), int(image.shape[2]))
    anchors = {k: lv[k].normalized_anchor_sizes(size) for k in lv}
    layers.set_inference_dtype('f16' if f16 else 'f32')
    try:
        with torch.no_grad():
            out = net(image, training=False)
            return utils.detect_raw(out["classifications"], out["regressions"], anchors, classes, logits=True)[0], out
    finally:
        layers.set_inference_dtype('f32')


def agreement(a, b):
    """(shared / len(a), shared / len(b), max |score diff| over shared, max box corner diff / box extent over shared); survivors are
    matched by class and IoU > 0.9"""
    import metrics
    if len(a.scores) == 0 or len(b.scores) == 0:
        return 1.0 if len(a.scores) == len(b.scores) else 0.0, 1.0 if len(a.scores) == len(b.scores) else 0.0, 0.0, 0.0
    ab, bb = a.boxes.cpu().numpy(), b.boxes.cpu().numpy()
    ac, bc = a.class_ids.cpu().numpy(), b.class_ids.cpu().numpy()
    as_, bs = a.scores.cpu().numpy(), b.scores.cpu().numpy()
    used, shared, ds, db = set(), 0, 0.0, 0.0
    for i in range(len(as_)):
        best, bj = 0.9, -1
        for j in range(len(bs)):
            if j in used or bc[j] != ac[i]:
                continue
            y1, x1 = max(ab[i, 0], bb[j, 0]), max(ab[i, 1], bb[j, 1])
            y2, x2 = min(ab[i, 2], bb[j, 2]), min(ab[i, 3], bb[j, 3])
            inter = max(y2 - y1, 0) * max(x2 - x1, 0)
            ua = (ab[i, 2] - ab[i, 0]) * (ab[i, 3] - ab[i, 1]) + (bb[j, 2] - bb[j, 0]) * (bb[j, 3] - bb[j, 1]) - inter
            iou = inter / ua if ua > 0 else 0.0
            if iou > best:
                best, bj = iou, j
        if bj >= 0:
            used.add(bj); shared += 1
            ds = max(ds, abs(float(as_[i]) - float(bs[bj])))
            ext = max(ab[i, 2] - ab[i, 0], ab[i, 3] - ab[i, 1], 1e-6)
            db = max(db, float(np.abs(ab[i] - bb[bj]).max()) / ext)
    return shared / len(as_), shared / len(bs), ds, db


def main(steps=2500, lr=1e-2, images=32):
    import dataset, metrics, train
    from data_loaders.shapes import Shapes
    from test_gpu_train_cli import _shapes_trainer
    dev = torch.device("cuda:0")
    net, tr, feed, lv = _shapes_trainer(dev, True, True, dropout=0.0, seed=0, scale=256, backbone='resnet_50')
    tr.opt.lr = lr
    t0 = time.perf_counter()
    for i in range(steps):
        out = tr.step()
        if (i + 1) % 500 == 0:
            print("step %d class_loss %.4f regr_loss %.4f (%.1f s)" % (i + 1, out["class_loss"].item(), out["regr_loss"].item(), time.perf_counter() - t0), flush=True)
    feed.close()
    loader = Shapes(None, image_size=(320, 256), seed=12345)
    it = dataset.build_dataset(loader, lv, scale=256, device=dev)
    d32, d16, gts, ag = [], [], [], []
    for _ in range(images):
        b = next(it)
        image = b['image'][:1]
        a, _ = detections(net, image, lv, loader.num_classes, False)
        h, _ = detections(net, image, lv, loader.num_classes, True)
        d32.append((a.boxes.cpu().numpy(), a.scores.cpu().numpy(), a.class_ids.cpu().numpy()))
        d16.append((h.boxes.cpu().numpy(), h.scores.float().cpu().numpy(), h.class_ids.cpu().numpy()))
        gts.append((np.asarray(b['boxes'], np.float32), np.asarray(b['class_ids'])))
        ag.append(agreement(a, h))
    m32 = metrics.mean_average_precision(d32, gts, loader.num_classes)
    m16 = metrics.mean_average_precision(d16, gts, loader.num_classes)
    ag = np.array(ag)
    print(json.dumps({"steps": steps, "lr": lr, "mAP_f32": round(m32["mAP"], 4), "mAP_f16": round(m16["mAP"], 4), "AP50_f32": round(m32["AP50"], 4),
                      "AP50_f16": round(m16["AP50"], 4), "survivors_f32": int(sum(len(d[1]) for d in d32)), "survivors_f16": int(sum(len(d[1]) for d in d16)),
                      "shared_of_f32_min": round(float(ag[:, 0].min()), 4), "shared_of_f32_mean": round(float(ag[:, 0].mean()), 4),
                      "shared_of_f16_mean": round(float(ag[:, 1].mean()), 4), "max_score_diff": round(float(ag[:, 2].max()), 5),
                      "max_box_diff_of_extent": round(float(ag[:, 3].max()), 5)}))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2500, float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2)
