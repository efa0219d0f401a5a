"""Anchor assignment on the device (drop-in for reference dataset.py:43-142 ``level_labels`` /
``build_labels``), batched over images.

    cls, reg, masks = build_labels(image_size, class_ids, boxes, num_obj, levels, num_classes)
    # dicts P3..P7: [N,H,W,A,C] f32 one-hot (zero where IoU<0.5), [N,H,W,A,4] f32, [N,H,W,A] u8

The tf.data input pipeline of the reference (file reading, JPEG decode, shuffle,
dataset.py:145-233) is out of scope; what remains of it here is the
label construction and the ``[image, hflip(image)]`` batch convention (dataset.py:182-204).
"""
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


def build_labels(image_size, class_ids, boxes, levels, num_classes, num_obj=None):
    labels = {pn: level_labels(image_size, class_ids, boxes, level=levels[pn], factor=2 ** int(pn[-1]),
                               num_classes=num_classes, num_obj=num_obj) for pn in levels}
    classifications = {pn: labels[pn][0] for pn in labels}
    regressions = {pn: labels[pn][1] for pn in labels}
    trainable_masks = {pn: labels[pn][2] for pn in labels}
    return classifications, regressions, trainable_masks


def flip_boxes(boxes):
    """h-flip of normalised corner boxes [.., 4] = [y1, x1, y2, x2] (labels of the flipped image
    of the reference's [image, hflip] batch, dataset.py:182-204, are built from these)."""
    return torch.stack([boxes[..., 0], 1.0 - boxes[..., 3], boxes[..., 2], 1.0 - boxes[..., 1]], -1)
