"""Horizontal-flip augmentation on the device (drop-in for reference augmentation.py:5-22).

    flipped = augmentation.flip(input)
`input` is the reference's sample dict: 'image' [H,W,3] (or a batch [N,H,W,3]), 'detection':
{'classifications': {Pk: [H,W,A,C]}, 'regressions': {Pk: [H,W,A,4]}}, 'trainable_masks': {Pk: [H,W,A]}.
Every map is reversed along its W axis and the x component (index 1) of the regression targets changes
sign.  `make_pair` builds the reference's batch of two = [sample, flip(sample)] (dataset.py:182-204).
"""
import torch

import _rn
import utils


def _flip(t, w_axis, neg_mod=0, neg_idx=0, out=None):
    t = t.contiguous()
    outer = 1
    for d in t.shape[:w_axis]:
        outer *= d
    inner = 1
    for d in t.shape[w_axis + 1:]:
        inner *= d
    esz = t.element_size()
    assert esz in (1, 4), "flip: fp32 maps or 1-byte masks"
    y = torch.empty_like(t) if out is None else out
    assert y.is_contiguous() and y.shape == t.shape and y.dtype == t.dtype
    _rn.check(_rn.lib().rn_flip_width(_rn.ptr(t), _rn.ptr(y), outer, t.shape[w_axis], inner, esz, neg_mod, neg_idx,
                                      _rn.stream()), "rn_flip_width")
    return y


def flip(input):
    batched = input['image'].dim() == 4
    ax = 2 if batched else 1
    image = _flip(input['image'], ax)
    classifications = utils.dict_map(lambda x: _flip(x, ax), input['detection']['classifications'])
    regressions = utils.dict_map(lambda x: _flip(x, ax, 4, 1), input['detection']['regressions'])
    trainable_masks = utils.dict_map(lambda x: _flip(x, ax), input['trainable_masks'])
    return {
        'image': image,
        'detection': {'classifications': classifications, 'regressions': regressions},
        'trainable_masks': trainable_masks,
    }


def make_pair(input):
    """[sample, hflip(sample)] stacked on a new leading axis (dataset.py:182-204)."""
    f = flip(input)
    stack = lambda a, b: torch.stack([a, b], 0)
    return {
        **input,
        'image': stack(input['image'], f['image']),
        'detection': {
            'classifications': utils.dict_starmap(stack, [input['detection']['classifications'], f['detection']['classifications']]),
            'regressions': utils.dict_starmap(stack, [input['detection']['regressions'], f['detection']['regressions']]),
        },
        'trainable_masks': utils.dict_starmap(stack, [input['trainable_masks'], f['trainable_masks']]),
    }
