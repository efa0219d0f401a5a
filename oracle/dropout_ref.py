"""Oracle: tf.layers.Dropout with INJECTED masks ("rate 0 or injected masks").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's dropout sites (mobilenet_v2.py:62,71,79,117,184; densenet.py:23,44,67,77,143) draw their masks from
TensorFlow's RNG stream, which cannot be reproduced.  What CAN be checked is everything around the draw: which tensors
are masked, the inverted scaling, the gradient through the same mask.  The product's mask is a pure function of
(seed of the site + step counter, flat NHWC element index) -- documented at ``rn_dropout`` in include/rn_hip.h -- so this
file restates that function in numpy and hands the oracle's forward functions the masks the kernels will draw:

  * ``uniform01(seed, idx)``          the counter-based uniform in [0, 1) (restated from the header's description:
                                      32-bit murmur3-style avalanche of the 64-bit seed and the 64-bit element index)
  * ``keep_mask(seed, shape, rate)``  bool mask, True = kept  (u >= rate, compared in fp32)
  * ``apply(x, keep, rate)``          [TF-sem] tf.nn.dropout (TF 1.x nn_ops.py): ``x / keep_prob * floor(keep_prob + u)``:
                                      kept elements are DIVIDED by keep_prob = 1 - rate, dropped ones are 0
  * ``Sites(seeds, rate, counter)``   the hook ``model_ref`` / ``backbones_ref`` call at every reference dropout site:
                                      ``hook(site_name, x) -> dropout(x)``

The product multiplies by fp32(1 / (1 - rate)) where TF divides by fp32(1 - rate): at most one rounding apart (6e-8).
"""
import numpy as np
import torch

_M32 = 0xFFFFFFFF


def uniform01(seed, idx):
    """Vectorised over idx (any integer array, taken as uint64).  All arithmetic modulo 2^32 as the kernels do it."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    idx = np.asarray(idx, dtype=np.uint64)
    lo = (idx & np.uint64(_M32)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = lo * np.uint32(0x9E3779B1) + np.uint32(seed & _M32)
        h = h ^ (hi * np.uint32(0x85EBCA77) + np.uint32(seed >> 32))
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13))
        h = h * np.uint32(0xC2B2AE35)
        h = h ^ (h >> np.uint32(16))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def uniform01_scalar(seed, idx):
    """The same function on Python integers, one element (the independent cross-check of the vectorised form)."""
    seed, idx = int(seed) & 0xFFFFFFFFFFFFFFFF, int(idx) & 0xFFFFFFFFFFFFFFFF
    h = ((idx & _M32) * 0x9E3779B1 + (seed & _M32)) & _M32
    h ^= ((idx >> 32) * 0x85EBCA77 + (seed >> 32)) & _M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & _M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & _M32
    h ^= h >> 16
    return float(h >> 8) / 16777216.0


def keep_mask(seed, shape, rate):
    """True where the element (flat C-order index into `shape`, i.e. NHWC) is kept."""
    n = int(np.prod(shape))
    u = uniform01(seed, np.arange(n, dtype=np.uint64))
    return (u >= np.float32(rate)).reshape(shape)


def apply(x, keep, rate):
    """[TF-sem] tf.nn.dropout(x, keep_prob = 1 - rate) with the binary tensor given: x / keep_prob * keep."""
    keep_prob = torch.tensor(1.0, dtype=torch.float32) - torch.tensor(float(rate), dtype=torch.float32)
    return x / keep_prob * torch.from_numpy(np.ascontiguousarray(keep)).to(x.dtype)


class Sites:
    """hook(site, x): dropout of tensor x at the named site with the mask of seed ``seeds[site] + counter``.

    ``seeds``: site name -> the seed the product's Dropout layer at that site carries; ``counter``: the value of the
    trainer's step counter (0 before the first step, +1 per optimizer step).  Every site asked for is recorded in
    ``.seen`` so a test can assert that the oracle visited exactly the product's sites."""

    def __init__(self, seeds, rate, counter=0):
        self.seeds, self.rate, self.counter = dict(seeds), float(rate), int(counter)
        self.seen = []

    def __call__(self, site, x):
        self.seen.append(site)
        if self.rate == 0.0:
            return x
        keep = keep_mask(self.seeds[site] + self.counter, tuple(x.shape), self.rate)
        return apply(x, keep, self.rate)
