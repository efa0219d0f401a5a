"""Oracle: anchor pyramid description (fp64 numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference ``levels.py``:
  * ``compute_box_size``  levels.py:49-52  -> box_size()
  * ``Level.anchor_sizes`` levels.py:32-44 -> anchor_table()
  * ``Levels`` / ``build_levels`` levels.py:5-29,55-59 -> pyramid()
"""
import itertools

import numpy as np

PYRAMID_BASE = (("P3", 32), ("P4", 64), ("P5", 128), ("P6", 256), ("P7", 512))
DEFAULT_ASPECTS = ((1, 2), (1, 1), (2, 1))
DEFAULT_SCALES = (2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3))


def box_size(base, aspect, scale):
    """(h, w) in pixels of an anchor of area base^2*scale^2 and given aspect (levels.py:49-52)."""
    aspect = np.asarray(aspect)
    return np.sqrt(base ** 2 / aspect.prod()) * aspect * scale


def anchor_table(base, aspects=DEFAULT_ASPECTS, scales=DEFAULT_SCALES):
    """[A, 2] fp64 (h, w); aspect-major, scale-minor order (levels.py:38-44)."""
    rows = [box_size(base, a, s) for a, s in itertools.product(aspects, scales)]
    return np.stack(rows, 0)


def pyramid(aspects=DEFAULT_ASPECTS, scales=DEFAULT_SCALES):
    """Ordered dict name -> [A,2] anchor sizes for P3..P7 (levels.py:10-16,55-59)."""
    return {name: anchor_table(base, aspects, scales) for name, base in PYRAMID_BASE}


def num_anchors(aspects=DEFAULT_ASPECTS, scales=DEFAULT_SCALES):
    return len(aspects) * len(scales)


def normalized_anchor_sizes(anchor_px, image_size, mode="trunc_int"):
    """float32 [A,2] anchor size / image size as the reference graph evaluates it.

    Reference expression: ``tf.to_float(level.anchor_sizes / image_size)`` at
    dataset.py:53 and utils.py:264, with ``level.anchor_sizes`` a float64 ndarray and
    ``image_size`` an int32 Tensor.

    [TF-sem] numpy defers to ``Tensor.__rtruediv__`` which converts the
    ndarray to the tensor's dtype (int32, ``ndarray.astype`` => truncation toward zero),
    then int32 true-division is evaluated in float64 and ``to_float`` rounds to float32.
    mode="trunc_int" reproduces that; mode="float" is the arithmetic a reader would
    expect (float64 divide, then round to float32).
    """
    anchor_px = np.asarray(anchor_px, dtype=np.float64)
    size = np.asarray(image_size, dtype=np.int64).astype(np.float64)
    if mode == "trunc_int":
        num = np.trunc(anchor_px)
    elif mode == "float":
        num = anchor_px
    else:
        raise ValueError(mode)
    return (num / size).astype(np.float32)
