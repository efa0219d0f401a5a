"""Oracle: TensorFlow-1.x op semantics the reference graph relies on, on torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  TensorFlow itself is an un-vendored,
un-pinned dependency of the reference (API use implies 1.8-1.12) and cannot be installed
here, so every rule below is a restatement of TF's published behaviour, tagged [TF-sem].
All tensors are NHWC; conv kernels are HWIO; depthwise kernels are [kh, kw, C, 1].

Reference call sites: conv ``retinanet.py:39,55,87,100,127,138,170,183,195``,
``mobilenet_v2.py:35,57,75,112,179``; moments ``normalization.py:30``; resize
``retinanet.py:154``; pools ``resnet.py:200``, ``densenet.py:144,180``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- padding
def same_pad_1d(n, k, s):
    """[TF-sem] SAME padding: out=ceil(n/s), total=max((out-1)s+k-n,0),
    before=total//2, after=total-before."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    before = total // 2
    return out, before, total - before


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- convs
def conv2d_same(x, w, stride=1, bias=None):
    """tf.layers.Conv2D(padding='same') on NHWC x, HWIO w."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pad_1d(x.shape[1], kh, stride)
    _, pl, pr = same_pad_1d(x.shape[2], kw, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), bias=bias, stride=stride)
    return _nhwc(y)


def depthwise_conv2d_same(x, w, stride=1):
    """tf.nn.depthwise_conv2d(padding='SAME'), kernel [kh,kw,C,1] (mobilenet_v2.py:35-36)."""
    kh, kw, c, mult = w.shape
    assert mult == 1
    _, pt, pb = same_pad_1d(x.shape[1], kh, stride)
    _, pl, pr = same_pad_1d(x.shape[2], kw, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    y = F.conv2d(xp, w.permute(2, 3, 0, 1), stride=stride, groups=c)
    return _nhwc(y)


def conv2d_same_naive(x, w, stride=1, bias=None):
    """Independent numpy loop nest (float64 accumulate) used to cross-check conv2d_same
    on small shapes."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    oh, pt, _ = same_pad_1d(h, kh, stride)
    ow, pl, _ = same_pad_1d(wd, kw, stride)
    y = np.zeros((n, oh, ow, co))
    for i in range(oh):
        for j in range(ow):
            for a in range(kh):
                for b in range(kw):
                    ih = i * stride + a - pt
                    iw = j * stride + b - pl
                    if 0 <= ih < h and 0 <= iw < wd:
                        y[:, i, j, :] += x[:, ih, iw, :] @ w[a, b]
    if bias is not None:
        y += np.asarray(bias, dtype=np.float64)
    return y


# --------------------------------------------------------------------------- norm/act
def gn_groups(c, groups=32):
    """Group count used for C channels.

    Reference: ``groups = min(self.groups, c)`` then reshape to [.., groups, c // groups]
    (normalization.py:24-27).  For C=144 (MobileNetV2 bottleneck_2_2 / 3_1) 32 does not
    divide C and the reference's reshape fails, so the build DEFINES the rule
    as the largest divisor of C that is <= 32 -- identical to the reference for every C
    the reference can execute.
    """
    g = min(groups, c)
    while c % g:
        g -= 1
    return g


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
    """normalization.py:20-35: moments over (H, W, C/G) per sample and group, biased
    variance [TF-sem: tf.nn.moments], eps inside the sqrt, per-channel gamma/beta."""
    n, h, w, c = x.shape
    g = gn_groups(c, groups)
    xg = x.reshape(n, h, w, g, c // g)
    mean = xg.mean(dim=(1, 2, 4), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 2, 4), keepdim=True)
    y = (xg - mean) / torch.sqrt(var + eps)
    return y.reshape(n, h, w, c) * gamma.reshape(1, 1, 1, c) + beta.reshape(1, 1, 1, c)


def activation(x, kind):
    """tf.nn.elu (train.py:214) / tf.nn.relu (resnet.py:85) / tf.nn.relu6 (mobilenet_v2.py:102)."""
    if kind in (None, "none", "linear"):
        return x
    if kind == "elu":
        return F.elu(x)
    if kind == "relu":
        return F.relu(x)
    if kind == "relu6":
        return torch.clamp(x, 0.0, 6.0)
    raise ValueError(kind)


# --------------------------------------------------------------------------- resize / pool
def nn_resize_index(out_size, in_size):
    """[TF-sem] ResizeNearestNeighbor(align_corners=True) source index:
    scale=(in-1)/(out-1) (0 if out==1), src=min(round(dst*scale), in-1), round = half away
    from zero (roundf) on float32."""
    if out_size > 1:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1)
    else:
        scale = np.float32(0)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale + np.float32(0.5)).astype(np.int64)
    return np.minimum(src, in_size - 1)


def upsample_nearest_align_corners(x, out_h, out_w):
    """tf.image.resize_images(NEAREST_NEIGHBOR, align_corners=True) (retinanet.py:153-155)."""
    ih = torch.from_numpy(nn_resize_index(out_h, x.shape[1]))
    iw = torch.from_numpy(nn_resize_index(out_w, x.shape[2]))
    return x[:, ih][:, :, iw]


def max_pool_same(x, k=3, stride=2):
    """[TF-sem] tf.layers.MaxPooling2D(padding='same'): padded cells never win (-inf)."""
    _, pt, pb = same_pad_1d(x.shape[1], k, stride)
    _, pl, pr = same_pad_1d(x.shape[2], k, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb), value=float("-inf"))
    return _nhwc(F.max_pool2d(xp, k, stride))


def avg_pool_same(x, k=2, stride=2):
    """[TF-sem] tf.layers.AveragePooling2D(padding='same'): divide by the number of VALID
    (un-padded) cells in each window."""
    _, pt, pb = same_pad_1d(x.shape[1], k, stride)
    _, pl, pr = same_pad_1d(x.shape[2], k, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    ones = F.pad(torch.ones_like(_nchw(x)[:1, :1]), (pl, pr, pt, pb))
    s = F.avg_pool2d(xp, k, stride) * (k * k)
    cnt = F.avg_pool2d(ones, k, stride) * (k * k)
    return _nhwc(s / cnt)


# --------------------------------------------------------------------------- linspace
def linspace_f32(start, stop, num):
    """[TF-sem] tf.linspace on float32: step=(stop-start)/(num-1);
    value[i]=start+step*i, every operation rounded to float32, no FMA contraction."""
    start = np.float32(start)
    stop = np.float32(stop)
    if num == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((stop - start) / np.float32(num - 1))
    idx = np.arange(num, dtype=np.float32)
    return (start + (step * idx).astype(np.float32)).astype(np.float32)


def cell_centers(size):
    """Cell-centre coordinates of a grid of `size` cells, as dataset.py:16-25 and
    utils.py:24-28 compute them: cell=to_float(1/size) [float64 divide, round to f32],
    linspace(cell/2, 1-cell/2, size)."""
    cell = np.float32(np.float64(1.0) / np.float64(size))
    half = np.float32(cell / np.float32(2))
    return linspace_f32(half, np.float32(np.float32(1) - half), size)


def he_fan_in_std(kh, kw, cin):
    """variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False):
    truncated normal with stddev sqrt(1.3*2/fan_in) (mobilenet_v2.py:104-105)."""
    return math.sqrt(1.3 * 2.0 / (kh * kw * cin))
