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
    # dataset.py:118-121: reduce_sum(regression * one_hot(iou_index, num_objects, axis=0), 0) -- a product with the
    # one-hot, not a gather: a non-finite entry (log of a zero / negative extent) of an object that is NOT the arg-max
    # still poisons the sum (-inf * 0 = NaN).  With finite entries the sum is the arg-max object's row exactly.
    sel = (np.arange(o).reshape(o, 1, 1, 1) == iou_index[None]).astype(f32)[..., None]
    with np.errstate(invalid="ignore"):
        regression = (regr_all * sel).sum(0, dtype=f32)
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


# --------------------------------------------------------------------------------------------------------------
# Input pipeline: rescale_image dataset.py:145-151, preprocess_image train.py:48-49,
# tf.image.convert_image_dtype (uint8 -> float32 * 1/255).  PARITY UNPINNED against TensorFlow itself (not
# installable here): this restates the published ResizeBilinear kernel (align_corners=True, half_pixel_centers=False:
# scale = (in-1)/(out-1); in = dst*scale; lo = floor(in); hi = min(ceil(in), size-1); lerp = in - lo;
# top = tl + (tr-tl)*xl; bot = bl + (br-bl)*xl; out = top + (bot-top)*yl), one float32 numpy op per TF operation.
MEAN = np.asarray([0.46618041, 0.44669811, 0.40252436], np.float32)   # dataset.py:12
STD = np.asarray([0.27940595, 0.27489075, 0.28920765], np.float32)    # dataset.py:13


def rescale_size(size, scale):
    size = np.asarray(size, np.float32)
    ratio = f32(scale) / size[int(np.argmin(size))]
    new = np.rint(size * ratio).astype(np.int32)      # tf.round: half to even
    return int(new[0]), int(new[1])


def _interp(out_size, in_size):
    scale = f32(in_size - 1) / f32(out_size - 1) if out_size > 1 else f32(0)
    src = np.arange(out_size, dtype=np.float32) * scale
    lo_f = np.floor(src)
    lo = np.maximum(lo_f.astype(np.int64), 0)
    hi = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
    return lo, hi, (src - lo_f).astype(np.float32)


def resize_bilinear_align_corners(image, out_h, out_w):
    """image [N,H,W,C] uint8 or float32 -> float32 [N,out_h,out_w,C]."""
    x = image.astype(np.float32) * f32(1.0 / 255.0) if image.dtype == np.uint8 else image.astype(np.float32)
    y0, y1, yl = _interp(out_h, x.shape[1])
    x0, x1, xl = _interp(out_w, x.shape[2])
    xl = xl[None, None, :, None]
    yl = yl[None, :, None, None]
    tl, tr = x[:, y0][:, :, x0], x[:, y0][:, :, x1]
    bl, br = x[:, y1][:, :, x0], x[:, y1][:, :, x1]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).astype(np.float32)


def rescale_image(image, scale):
    oh, ow = rescale_size(image.shape[-3:-1], scale)
    return resize_bilinear_align_corners(image if image.ndim == 4 else image[None], oh, ow)[slice(None) if image.ndim == 4 else 0]


def preprocess_image(image):
    return ((image.astype(np.float32) - MEAN) / STD).astype(np.float32)
