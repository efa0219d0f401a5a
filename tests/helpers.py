"""Shared helpers for the parity tests (oracle <-> HIP path)."""
import re

import numpy as np
import torch


def to_oracle_name(name):
    """Product state_dict name -> oracle parameter name (oracle/model_ref.py)."""
    name = re.sub(r"^base\.", "", name)
    name = re.sub(r"\._mods\.(\d+)\._mods\.0\.", r".\1.conv.", name)
    name = re.sub(r"\._mods\.(\d+)\._mods\.1\.", r".\1.norm.", name)
    name = re.sub(r"\._mods\.0\.", ".conv.", name)
    name = re.sub(r"\._mods\.1\.", ".norm.", name)
    return name


def load_oracle_params(net, params):
    """Copy an oracle parameter dict into the product model (shapes must agree)."""
    with torch.no_grad():
        seen = set()
        for name, p in net.named_parameters():
            o = to_oracle_name(name)
            assert o in params, "no oracle parameter for %s (%s)" % (name, o)
            assert tuple(p.shape) == tuple(params[o].shape), (name, p.shape, params[o].shape)
            p.copy_(params[o].to(p.device))
            seen.add(o)
        assert seen == set(params.keys()), sorted(set(params.keys()) - seen)[:5]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    return d / max(np.abs(b).max() if b.size else 0.0, 1e-30)


def elementwise_rel_err(a, b, floor_frac=1e-3):
    """max_i |a_i - b_i| / max(|b_i|, floor_frac * max|b|): the element-wise relative error, with a floor so that
    elements that are (analytically) zero and hold only rounding noise do not divide by ~0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    floor = max(floor_frac * np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def assert_close(a, b, tol, what="", elementwise_tol=None):
    """Max-norm relative error <= tol (the bar for tensors); the element-wise figure is computed and reported beside it,
    and asserted too when `elementwise_tol` is given (losses, boxes: the quantities north_star's 1e-4 is stated on)."""
    e = rel_err(a, b)
    ew = elementwise_rel_err(a, b)
    assert e <= tol, "%s: max-norm relative error %.3e > %.1e (element-wise %.3e)" % (what, e, tol, ew)
    if elementwise_tol is not None:
        assert ew <= elementwise_tol, "%s: element-wise relative error %.3e > %.1e (max-norm %.3e)" % (what, ew, elementwise_tol, e)
    return e


def random_boxes(rng, n, min_size=0.05, max_size=0.5):
    c = rng.uniform(0.15, 0.85, (n, 2))
    s = rng.uniform(min_size, max_size, (n, 2))
    b = np.concatenate([c - s / 2, c + s / 2], 1)
    return np.clip(b, 0.0, 1.0).astype(np.float32)


def coco_like_objects(rng, image_size, max_obj=32):
    """O ~ clip(Poisson(7), 1, 32); class ~ U{0..79}; centre ~ U(0,1)^2;
    side = S * 2^U(-4,-1) px with aspect 2^U(-1,1), clipped to the image."""
    o = int(np.clip(rng.poisson(7), 1, max_obj))
    cy, cx = rng.uniform(0, 1, o), rng.uniform(0, 1, o)
    side = 2.0 ** rng.uniform(-4, -1, o)
    asp = 2.0 ** rng.uniform(-1, 1, o)
    h, w = side * np.sqrt(asp), side / np.sqrt(asp)
    y1, x1 = np.clip(cy - h / 2, 0, 1), np.clip(cx - w / 2, 0, 1)
    y2, x2 = np.clip(cy + h / 2, 0, 1), np.clip(cx + w / 2, 0, 1)
    y2 = np.maximum(y2, y1 + 2.0 / image_size)
    x2 = np.maximum(x2, x1 + 2.0 / image_size)
    boxes = np.stack([y1, x1, np.minimum(y2, 1.0), np.minimum(x2, 1.0)], 1).astype(np.float32)
    cls = rng.integers(0, 80, o).astype(np.int32)
    return boxes, cls


def dropout_sites(net, rate, counter=0):
    """The oracle's dropout hook (oracle/dropout_ref.Sites) carrying the seeds of the PRODUCT's Dropout layers: site name =
    the layer's module path without the leading ``base.``; for the [conv, Normalization, act, Dropout] blocks of MobileNetV2
    the trailing ``._mods.<i>`` becomes ``.dropout`` (oracle/model_ref._cna), DenseNet's composite functions / transition
    layers keep their ``._mods.<i>`` (oracle/backbones_ref.densenet_forward)."""
    import layers
    from oracle import dropout_ref
    seeds = {}
    for name, m in net.named_modules():
        if isinstance(m, layers.Dropout):
            name = re.sub(r"^base\.", "", name)
            if "._fns." not in name and "transition_layer" not in name:
                name = re.sub(r"\._mods\.\d+$", ".dropout", name)
            if not name.startswith("backbone"):
                name = "backbone." + name if name else name
            seeds[name] = m.seed
    return dropout_ref.Sites(seeds, rate, counter)
