"""Oracle: anchor assignment (label construction) and the h-flip augmentation, numpy f32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference ``dataset.py``:
  * position_grid / to_center_box / from_center_box  dataset.py:16-39
  * level_labels                                     dataset.py:43-123
  * build_labels                                     dataset.py:126-142
and ``augmentation.flip`` augmentation.py:5-22.
Each float32 operation is its own numpy op in the reference's order so the masks and
class maps can be compared bit for bit with the HIP kernel.
"""
import numpy as np

from . import levels_ref, utils_ref

NEG_IOU_THRESHOLD = np.float32(0.4)     # dataset.py:10
POS_IOU_THRESHOLD = np.float32(0.5)     # dataset.py:11
f32 = np.float32


def to_center_box(box):
    """dataset.py:28-32: [a, b] -> [a + (b-a)/2, b-a]."""
    box = np.asarray(box, dtype=f32)
    a, b = box[..., :2], box[..., 2:]
    size = (b - a).astype(f32)
    return np.concatenate([(a + (size / f32(2)).astype(f32)).astype(f32), size], -1)


def from_center_box(box):
    """dataset.py:35-39."""
    box = np.asarray(box, dtype=f32)
    pos, size = box[..., :2], box[..., 2:]
    half = (size / f32(2)).astype(f32)
    return np.concatenate([(pos - half).astype(f32), (pos + half).astype(f32)], -1)


def level_labels(image_size, class_id, true_box, anchor_sizes_px, factor, num_classes,
                 anchor_mode="trunc_int"):
    """dataset.py:43-123 for one pyramid level.

    image_size: (H, W) ints; class_id [O] ints; true_box [O,4] normalised corners;
    anchor_sizes_px [A,2] fp64.  Returns (classification [H,W,A,C] f32 one-hot zeroed
    where IoU<0.5, regression [H,W,A,4] f32 of the arg-max object (NOT zeroed on bg),
    trainable [H,W,A] bool = IoU<0.4 or IoU>=0.5) plus the arg-max index [H,W,A].
    """
    class_id = np.asarray(class_id, dtype=np.int64)
    true_box = np.asarray(true_box, dtype=f32)
    o = true_box.shape[0]
    true_c = to_center_box(true_box).reshape(o, 1, 1, 1, 4)
    anchor_size = levels_ref.normalized_anchor_sizes(anchor_sizes_px, image_size, anchor_mode)
    a = anchor_size.shape[0]
    gh = int(np.ceil(image_size[0] / factor))
    gw = int(np.ceil(image_size[1] / factor))
    pos = utils_ref.grid_positions(gh, gw).reshape(1, gh, gw, 1, 2)
    pos = np.tile(pos, (1, 1, 1, a, 1))
    size = np.tile(anchor_size.reshape(1, 1, 1, a, 2), (1, gh, gw, 1, 1))
    anchor = np.concatenate([pos, size], -1)

    iou = utils_ref.iou(from_center_box(anchor), from_center_box(true_c))      # [O,H,W,A]
    iou_index = iou.argmax(0)
    iou_value = iou.max(0)
    bg_mask = iou_value < POS_IOU_THRESHOLD                                    # :83 (Q6)
    trainable = (iou_value < NEG_IOU_THRESHOLD) | (iou_value >= POS_IOU_THRESHOLD)  # :87

    cls = class_id[iou_index]
    # [TF-sem] tf.one_hot: indices outside [0, C) give an all-zero row
    onehot = (cls[..., None] == np.arange(num_classes)).astype(f32)
    onehot = np.where(bg_mask[..., None], f32(0), onehot).astype(f32)

    t_pos, t_size = true_c[..., :2], true_c[..., 2:]
    shifts = ((t_pos - pos).astype(f32) / size).astype(f32)                    # [O,H,W,A,2]
    with np.errstate(divide="ignore"):
        scales = np.log((t_size / size).astype(f32)).astype(f32)
    regr_all = np.concatenate([shifts, scales], -1)
    regression = np.take_along_axis(regr_all, iou_index[None, ..., None], 0)[0]
    return onehot, regression.astype(f32), trainable, iou_index


def build_labels(image_size, class_ids, boxes, num_classes, anchor_mode="trunc_int"):
    """dataset.py:126-142 -> three dicts P3..P7."""
    pyr = levels_ref.pyramid()
    cls, reg, msk = {}, {}, {}
    for name, sizes in pyr.items():
        c, r, m, _ = level_labels(image_size, class_ids, boxes, sizes, 2 ** int(name[-1]),
                                  num_classes, anchor_mode)
        cls[name], reg[name], msk[name] = c, r, m
    return cls, reg, msk


def flip(classifications, regressions, trainable_masks, image=None):
    """augmentation.py:5-22: reverse the W axis of every map and negate the x shift."""
    out_c = {k: v[:, ::-1].copy() for k, v in classifications.items()}
    out_m = {k: v[:, ::-1].copy() for k, v in trainable_masks.items()}
    out_r = {}
    for k, v in regressions.items():
        r = v[:, ::-1].copy()
        r[..., 1] = -r[..., 1]
        out_r[k] = r
    img = None if image is None else image[:, ::-1].copy()
    return out_c, out_r, out_m, img
