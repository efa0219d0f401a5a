"""Writes the fixtures under tests/golden/.  Run in the BUILD container only:

    python tests/golden/make_golden.py

1. ``levels_reference.npz`` -- produced by importing the REFERENCE's own ``levels.py``
   (numpy-only, the one reference module that imports here) from /root/reference and
   recording ``build_levels()``'s anchor sizes / num_anchors / key order.
2. ``oracle_e2e_tiny.npz`` -- a seeded end-to-end case from the CPU oracle (S=64, C=3,
   N=2) that freezes the oracle's own outputs so a later edit of the oracle is noticed.
The reference never travels to the GPU box; only these arrays do.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def levels_fixture():
    sys.path.insert(0, "/root/reference")
    import levels as ref_levels                      # the reference module itself
    lv = ref_levels.build_levels()
    data = {"keys": np.array(list(lv.keys())), "num_anchors": np.array(lv.num_anchors)}
    for k in lv:
        data["anchor_sizes_" + k] = np.asarray(lv[k].anchor_sizes, dtype=np.float64)
    data["box_size_32_1x2_1"] = ref_levels.compute_box_size(32, (1, 2), 1)
    data["level_32_1x4"] = ref_levels.Level(32, [(1, 4)], [2 ** 0, 2 ** 1]).anchor_sizes
    np.savez(os.path.join(HERE, "levels_reference.npz"), **data)
    sys.path.pop(0)
    del sys.modules["levels"]


def oracle_fixture():
    sys.path.insert(0, ROOT)
    import torch
    from oracle import dataset_ref, model_ref, train_ref
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    s, c, n = 64, 3, 2
    params = model_ref.init_params("mobilenet_v2", num_classes=c, seed=0)
    image = torch.from_numpy(rng.standard_normal((n, s, s, 3)).astype(np.float32))
    boxes = np.array([[0.1, 0.15, 0.6, 0.7], [0.5, 0.4, 0.95, 0.9]], dtype=np.float32)
    cids = np.array([0, 2])
    cls, reg, msk = dataset_ref.build_labels((s, s), cids, boxes, c)
    fc, fr, fm, _ = dataset_ref.flip(cls, reg, msk)
    labels = {"classifications": {}, "regressions": {}, "trainable_masks": {}}
    for k in cls:
        labels["classifications"][k] = torch.from_numpy(np.stack([cls[k], fc[k]]))
        labels["regressions"][k] = torch.from_numpy(np.stack([reg[k], fr[k]]))
        labels["trainable_masks"][k] = torch.from_numpy(np.stack([msk[k], fm[k]]))
    out = {}
    for mode in ("bce_dice", "focal"):
        with torch.no_grad():
            tot = train_ref.total_loss(params, image, labels, c, mode)
        out["loss_" + mode] = np.array([float(x) for x in tot], dtype=np.float64)
    with torch.no_grad():
        fwd = model_ref.retinanet_forward(params, image, c)
    out["cls_P5"] = fwd["classifications"]["P5"].numpy()
    out["reg_P7"] = fwd["regressions"]["P7"].numpy()
    out["image"] = image.numpy()
    out["boxes"] = boxes
    out["class_ids"] = cids
    np.savez_compressed(os.path.join(HERE, "oracle_e2e_tiny.npz"), **out)


if __name__ == "__main__":
    levels_fixture()
    oracle_fixture()
    print("fixtures written to", HERE)
