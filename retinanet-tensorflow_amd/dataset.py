"""Anchor assignment on the device (drop-in for reference dataset.py:43-142 ``level_labels`` /
``build_labels``), batched over images.

    cls, reg, masks = build_labels(image_size, class_ids, boxes, num_obj, levels, num_classes)
    # dicts P3..P7: [N,H,W,A,C] f32 one-hot (zero where IoU<0.5), [N,H,W,A,4] f32, [N,H,W,A] u8

The tf.data input pipeline of the reference (file reading, JPEG decode, shuffle,
dataset.py:145-233) is out of scope; what remains of it here is the
label construction and the ``[image, hflip(image)]`` batch convention (dataset.py:182-204).
"""
import ctypes as C
import math

import numpy as np
import torch

import _rn
import utils

NEG_IOU_THRESHOLD = 0.4
POS_IOU_THRESHOLD = 0.5
MEAN = [0.46618041, 0.44669811, 0.40252436]
STD = [0.27940595, 0.27489075, 0.28920765]
ANCHOR_SIZE_MODE = 'trunc_int'


def level_labels(image_size, class_id, true_box, level, factor, num_classes, num_obj=None, return_argmax=False):
    """class_id [N, O] int32, true_box [N, O, 4] normalised corners, num_obj [N] (valid objects per
    image, default O) -> (classification [N,H,W,A,C], regression [N,H,W,A,4], trainable [N,H,W,A])."""
    dev = true_box.device
    n, o = true_box.shape[0], true_box.shape[1]
    true_box = true_box.contiguous().float()
    class_id = class_id.to(torch.int32).contiguous()
    if num_obj is None:
        num_obj = torch.full((n,), o, dtype=torch.int32, device=dev)
    num_obj = num_obj.to(torch.int32).contiguous()
    anchors = utils._anchor_tensor(level.normalized_anchor_sizes(image_size, ANCHOR_SIZE_MODE), dev)
    a = anchors.shape[0]
    gh, gw = int(math.ceil(image_size[0] / factor)), int(math.ceil(image_size[1] / factor))
    cls = torch.empty((n, gh, gw, a, num_classes), dtype=torch.float32, device=dev)
    reg = torch.empty((n, gh, gw, a, 4), dtype=torch.float32, device=dev)
    msk = torch.empty((n, gh, gw, a), dtype=torch.uint8, device=dev)
    arg = torch.empty((n, gh, gw, a), dtype=torch.int32, device=dev) if return_argmax else None
    _rn.check(_rn.lib().rn_anchor_assign(_rn.f32(true_box), _rn.ptr(class_id), _rn.ptr(num_obj), n, o,
                                         _rn.f32(anchors), a, gh, gw, num_classes, _rn.f32(cls), _rn.f32(reg),
                                         _rn.ptr(msk), _rn.ptr(arg), _rn.stream()), 'rn_anchor_assign')
    if return_argmax:
        return cls, reg, msk, arg
    return cls, reg, msk


def build_labels(image_size, class_ids, boxes, levels, num_classes, num_obj=None, flip_pair=False):
    """dataset.py:126-142 for a batch: every level's maps from one launch (rn_anchor_assign_levels); the per-level
    results are the ones `level_labels` gives.
    flip_pair=True: the reference's batch of two per sample (dataset.py:182-204) from the same launch -- image i is assigned
    once and fills batch slots 2i (as is) and 2i+1 (= augmentation.flip of its maps: W reversed, x shift negated), so the
    mirror image's labels are bit for bit the flipped maps, and the assignment runs once per sample."""
    dev = boxes.device
    n_src, o = boxes.shape[0], boxes.shape[1]
    n = 2 * n_src if flip_pair else n_src
    boxes = boxes.contiguous().float()
    class_ids = class_ids.to(torch.int32).contiguous()
    if num_obj is None:
        num_obj = torch.full((n_src,), o, dtype=torch.int32, device=dev)
    num_obj = num_obj.to(torch.int32).contiguous()
    names = list(levels)
    lv = (_rn.AssignLevel * len(names))()
    classifications, regressions, trainable_masks, keep = {}, {}, {}, []
    a = None
    for i, pn in enumerate(names):
        factor = 2 ** int(pn[-1])
        anchors = utils._anchor_tensor(levels[pn].normalized_anchor_sizes(image_size, ANCHOR_SIZE_MODE), dev)
        assert a in (None, anchors.shape[0]), "every level carries the same number of anchors (levels.py:32-44)"
        a = anchors.shape[0]
        gh, gw = int(math.ceil(image_size[0] / factor)), int(math.ceil(image_size[1] / factor))
        classifications[pn] = torch.empty((n, gh, gw, a, num_classes), dtype=torch.float32, device=dev)
        regressions[pn] = torch.empty((n, gh, gw, a, 4), dtype=torch.float32, device=dev)
        trainable_masks[pn] = torch.empty((n, gh, gw, a), dtype=torch.uint8, device=dev)
        keep.append(anchors)
        lv[i] = _rn.AssignLevel(anchors.data_ptr(), gh, gw, classifications[pn].data_ptr(), regressions[pn].data_ptr(),
                                trainable_masks[pn].data_ptr(), None)
    fn = _rn.lib().rn_anchor_assign_levels_pair if flip_pair else _rn.lib().rn_anchor_assign_levels
    _rn.check(fn(_rn.f32(boxes), _rn.ptr(class_ids), _rn.ptr(num_obj), n_src, o, lv, len(names), a, num_classes, _rn.stream()),
              'rn_anchor_assign_levels')
    return classifications, regressions, trainable_masks


def flip_boxes(boxes):
    """h-flip of normalised corner boxes [.., 4] = [y1, x1, y2, x2] (labels of the flipped image
    of the reference's [image, hflip] batch, dataset.py:182-204, are built from these)."""
    return torch.stack([boxes[..., 0], 1.0 - boxes[..., 3], boxes[..., 2], 1.0 - boxes[..., 1]], -1)


def rescale_size(size, scale):
    """New (h, w) of dataset.py:145-151: shorter side -> `scale`, tf.round (half to even) of size * ratio, the
    arithmetic in float32 as tf.to_float / tf.round do it."""
    size = np.asarray(size, np.float32)
    ratio = np.float32(scale) / size[int(np.argmin(size))]
    new = np.rint(size * ratio).astype(np.int32)                 # np.rint == tf.round: half to even
    return int(new[0]), int(new[1])


def rescale_image(image, scale=None, size=None, normalize=False):
    """tf.image.resize_images(image, new_size, BILINEAR, align_corners=True) (dataset.py:145-151) on the device.
    image: [H,W,C] or [N,H,W,C], uint8 (converted like tf.image.convert_image_dtype: * 1/255) or fp32.
    normalize=True also applies train.py:48-49 preprocess_image ((v - MEAN) / STD) in the same pass."""
    batched = image.dim() == 4
    x = (image if batched else image[None]).contiguous()
    n, h, w, c = x.shape
    oh, ow = size if size is not None else rescale_size((h, w), scale)
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    assert x.dtype in (torch.uint8, torch.float32)
    mean = std = None
    if normalize:
        assert c == 3
        mean = (C.c_float * 3)(*MEAN)
        std = (C.c_float * 3)(*STD)
    _rn.check(_rn.lib().rn_resize_bilinear_normalize(_rn.ptr(x), 1 if x.dtype == torch.uint8 else 0, _rn.f32(y), n, h, w, c,
                                                     oh, ow, mean, std, _rn.stream()), 'rn_resize_bilinear_normalize')
    return y if batched else y[0]


def preprocess_image(image):
    """(image - MEAN) / STD (train.py:48-49) for a float image already at its final size."""
    return rescale_image(image, size=tuple(image.shape[-3:-1]), normalize=True)


def build_dataset(data_loader, levels, scale=None, shuffle=None, augment=False, device='cuda', normalize=True):
    """Generator form of dataset.py:154-215: per sample  decode -> boxes / image_size -> rescale_image ->
    build_labels -> [sample, hflip(sample)] batch (augmentation.make_pair) -> preprocess_image.  Everything after the
    host loader runs on the device.  `shuffle` / `augment` are accepted for signature parity (the reference's
    augment_sample is a TODO stub; shuffling belongs to the loader here)."""
    import augmentation
    dev = torch.device(device)
    for sample in data_loader:
        image = torch.from_numpy(np.ascontiguousarray(sample['image'])).to(dev)                # uint8 or float [H,W,3]
        h, w = int(image.shape[0]), int(image.shape[1])
        boxes = np.asarray(sample['boxes'], np.float32) / np.asarray([h, w, h, w], np.float32)   # dataset.py:163
        if scale is not None:
            image = rescale_image(image, scale)
        elif image.dtype == torch.uint8:
            image = rescale_image(image, size=(h, w))
        size = (int(image.shape[0]), int(image.shape[1]))
        ids = torch.from_numpy(np.asarray(sample['class_ids'], np.int32)).to(dev)[None]
        # the labels of [sample, hflip(sample)] from ONE assignment launch (flip_pair: no stack / flip kernels for the maps)
        c, r, m = build_labels(size, ids, torch.from_numpy(boxes).to(dev)[None], levels, data_loader.num_classes, flip_pair=True)
        batch = {'image': torch.stack([image, augmentation._flip(image, 1)], 0), 'image_size': size, 'boxes': boxes,
                 'class_ids': sample['class_ids'], 'detection': {'classifications': c, 'regressions': r}, 'trainable_masks': m}
        if normalize:
            batch['image'] = preprocess_image(batch['image'])
        yield batch
