"""Anchor pyramid API (drop-in for reference levels.py:5-59), pure numpy fp64.

    levels = build_levels(); levels.num_anchors; levels.keys(); levels['P3'].anchor_sizes
Same names, argument meaning and iteration protocol (``for k in levels``) as the reference;
``anchor_sizes`` is the [A, 2] float64 (h, w) pixel table, aspect-major / scale-minor.
The pyramid is fixed at P3..P7 with base sizes 32 * 2**i (levels.py:10-16).
"""
from collections import OrderedDict
from itertools import product

import numpy as np

_BASE_SIZES = OrderedDict((('P%d' % i, 2 ** (i + 2)) for i in range(3, 8)))


def compute_box_size(base_size, aspect_ratio, scale_ratio):
    """(h, w) of a box of area (base_size * scale_ratio)**2 with h:w = aspect_ratio."""
    ratio = np.array(aspect_ratio)
    unit = np.sqrt(base_size ** 2 / ratio.prod())
    return unit * ratio * scale_ratio


class Level(object):
    def __init__(self, anchor_size, anchor_aspect_ratios, anchor_scale_ratios):
        self._base = anchor_size
        self._aspects = anchor_aspect_ratios
        self._scales = anchor_scale_ratios

    @property
    def anchor_sizes(self):
        pairs = product(self._aspects, self._scales)
        return np.stack([compute_box_size(self._base, a, s) for a, s in pairs], 0)

    def normalized_anchor_sizes(self, image_size, mode='trunc_int'):
        """float32 [A, 2] = anchor_sizes / image_size as the reference GRAPH evaluates
        ``tf.to_float(level.anchor_sizes / image_size)`` (dataset.py:53, utils.py:264) with an
        int32 image_size tensor: [TF-sem] the float64 table is first cast to int32 (truncation,
        SURVEY Q1) -> mode 'trunc_int'; mode 'float' divides the un-truncated table."""
        table = self.anchor_sizes
        if mode == 'trunc_int':
            table = np.trunc(table)
        elif mode != 'float':
            raise ValueError(mode)
        size = np.asarray(image_size, dtype=np.int64).astype(np.float64)
        return (table / size).astype(np.float32)


class Levels(object):
    def __init__(self, anchor_aspect_ratios, anchor_scale_ratios):
        self._aspects = anchor_aspect_ratios
        self._scales = anchor_scale_ratios
        self._levels = OrderedDict(
            (name, Level(base, anchor_aspect_ratios, anchor_scale_ratios)) for name, base in _BASE_SIZES.items())

    @property
    def num_anchors(self):
        return len(self._aspects) * len(self._scales)

    def keys(self):
        return self._levels.keys()

    def __getitem__(self, item):
        return self._levels[item]

    def __iter__(self):
        return iter(self.keys())


def build_levels():
    aspects = [(1, 2), (1, 1), (2, 1)]
    scales = [2 ** (i / 3) for i in range(3)]
    return Levels(aspects, scales)
