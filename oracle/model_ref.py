"""Oracle: RetinaNet forward (backbones, FPN, shared heads) on torch-CPU fp32, functional.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parameters live in a flat ``dict[str, torch.Tensor]`` whose keys are the product model's
``state_dict`` names, so a test can move weights either way.

Follows the reference:
  * MobileNetV2            mobilenet_v2.py:41-223   -> mobilenet_v2_forward()
  * ResNeXt-50 ("resnet_50") resnet.py:15-215      -> resnext50_forward()
  * DenseNet-BC-121/169    densenet.py:26-310       -> densenet_forward()
  * FeaturePyramidNetwork  retinanet.py:118-221     -> fpn_forward()
  * Classification/RegressionSubnet retinanet.py:24-115 -> subnet_forward()
  * RetinaNetBase.call     retinanet.py:272-296     -> retinanet_forward()
Dropout: the reference's sites (mobilenet_v2.py:62,71,79,117,184) call the ``dropout`` hook when one is given
(``dropout(site, x) -> x``, see oracle/dropout_ref.py: the product's counter-based masks injected);
with ``dropout=None`` they are the identity (dropout_rate 0 / training=False).  TF's own RNG stream cannot be reproduced.
"""
import math

import torch

from . import tf_ops_ref as T

# (name, filters, expansion, stride) -- mobilenet_v2.py:120-176
MOBILENET_V2_BLOCKS = (
    ("bottleneck_1_1", 16, 1, 1),
    ("bottleneck_2_1", 24, 6, 2), ("bottleneck_2_2", 24, 6, 1),
    ("bottleneck_3_1", 32, 6, 2), ("bottleneck_3_2", 32, 6, 1), ("bottleneck_3_3", 32, 6, 1),
    ("bottleneck_4_1", 64, 6, 2), ("bottleneck_4_2", 64, 6, 1), ("bottleneck_4_3", 64, 6, 1),
    ("bottleneck_4_4", 64, 6, 1),
    ("bottleneck_5_1", 96, 6, 1), ("bottleneck_5_2", 96, 6, 1), ("bottleneck_5_3", 96, 6, 1),
    ("bottleneck_6_1", 160, 6, 2), ("bottleneck_6_2", 160, 6, 1), ("bottleneck_6_3", 160, 6, 1),
    ("bottleneck_7_1", 320, 6, 1),
)
# taps: C1 after 1_1, C2 after 2_2, C3 after 3_3, C4 after 5_3, C5 after output_conv
MOBILENET_V2_TAPS = {"bottleneck_1_1": "C1", "bottleneck_2_2": "C2", "bottleneck_3_3": "C3",
                     "bottleneck_5_3": "C4"}


def _cna(p, prefix, x, act, stride=1, depthwise=False, dropout=None):
    """conv -> GroupNorm -> activation [-> Dropout] (one reference Sequential([...]) block); the Dropout of the
    MobileNetV2 blocks is the hook's site ``prefix + ".dropout"``."""
    w = p[prefix + ".conv.weight"]
    x = T.depthwise_conv2d_same(x, w, stride) if depthwise else T.conv2d_same(x, w, stride)
    x = T.group_norm(x, p[prefix + ".norm.gamma"], p[prefix + ".norm.beta"])
    x = T.activation(x, act)
    return dropout(prefix + ".dropout", x) if dropout is not None else x


# --------------------------------------------------------------------------- MobileNetV2
def mobilenet_v2_forward(p, x, act="elu", prefix="backbone", dropout=None):
    """mobilenet_v2.py:187-223; every block ends with tf.layers.Dropout (:62 expand, :71 depthwise, :79 linear, :117
    input conv, :184 output conv)."""
    out = {}
    x = _cna(p, prefix + ".input_conv", x, act, stride=2, dropout=dropout)
    for name, _filters, _t, stride in MOBILENET_V2_BLOCKS:
        b = "%s.%s" % (prefix, name)
        identity = x
        x = _cna(p, b + ".expand_conv", x, act, dropout=dropout)
        x = _cna(p, b + ".depthwise_conv", x, act, stride=stride, depthwise=True, dropout=dropout)
        x = _cna(p, b + ".linear_conv", x, None, dropout=dropout)
        if x.shape == identity.shape:            # mobilenet_v2.py:91-92
            x = x + identity
        if name in MOBILENET_V2_TAPS:
            out[MOBILENET_V2_TAPS[name]] = x
    x = _cna(p, prefix + ".output_conv", x, act, dropout=dropout)
    out["C5"] = x
    return out


def mobilenet_v2_param_shapes(prefix="backbone"):
    shapes = {}

    def cna(name, kh, cin, cout, depthwise=False):
        shapes[name + ".conv.weight"] = (kh, kh, cin, 1) if depthwise else (kh, kh, cin, cout)
        shapes[name + ".norm.gamma"] = (cout,)
        shapes[name + ".norm.beta"] = (cout,)

    cna(prefix + ".input_conv", 3, 3, 32)
    c = 32
    for name, filters, t, _stride in MOBILENET_V2_BLOCKS:
        b = "%s.%s" % (prefix, name)
        cna(b + ".expand_conv", 1, c, c * t)
        cna(b + ".depthwise_conv", 3, c * t, c * t, depthwise=True)
        cna(b + ".linear_conv", 1, c * t, filters)
        c = filters
    cna(prefix + ".output_conv", 1, c, 32)       # 32 filters, mobilenet_v2.py:178-185 (Q8)
    return shapes


# --------------------------------------------------------------------------- FPN + heads
def fpn_forward(p, feats, act="elu", prefix="fpn"):
    """retinanet.py:214-221 and UpsampleMerge.call :151-160."""
    def cn(name, x, stride=1):
        return _cna(p, "%s.%s" % (prefix, name), x, None, stride=stride)

    p6 = cn("p6_from_c5", feats["C5"], stride=2)
    p7 = cn("p7_from_p6", T.activation(p6, act), stride=2)
    p5 = cn("p5_from_c5", feats["C5"])

    def merge(name, lateral, top):
        lat = cn(name + ".conv_lateral", lateral)
        up = T.upsample_nearest_align_corners(top, lat.shape[1], lat.shape[2])
        return cn(name + ".conv_merge", lat + up)

    p4 = merge("p4_from_c4p5", feats["C4"], p5)
    p3 = merge("p3_from_c3p4", feats["C3"], p4)
    return {"P3": p3, "P4": p4, "P5": p5, "P6": p6, "P7": p7}


def subnet_forward(p, x, prefix, num_anchors, last_dim, act="elu"):
    """retinanet.py:64-71 / :108-115 -- 4x[conv3x3, GN, act] + conv3x3(+bias), reshape."""
    for i in range(4):
        x = _cna(p, "%s.pre_conv.%d" % (prefix, i), x, act)
    x = T.conv2d_same(x, p[prefix + ".out_conv.weight"], 1, bias=p[prefix + ".out_conv.bias"])
    n, h, w, _ = x.shape
    return x.reshape(n, h, w, num_anchors, last_dim)


def fpn_head_param_shapes(c3, c4, c5, num_anchors, num_classes):
    shapes = {}

    def cn(name, k, cin, cout=256):
        shapes[name + ".conv.weight"] = (k, k, cin, cout)
        shapes[name + ".norm.gamma"] = (cout,)
        shapes[name + ".norm.beta"] = (cout,)

    cn("fpn.p6_from_c5", 3, c5)
    cn("fpn.p7_from_p6", 3, 256)
    cn("fpn.p5_from_c5", 1, c5)
    cn("fpn.p4_from_c4p5.conv_lateral", 1, c4)
    cn("fpn.p4_from_c4p5.conv_merge", 3, 256)
    cn("fpn.p3_from_c3p4.conv_lateral", 1, c3)
    cn("fpn.p3_from_c3p4.conv_merge", 3, 256)
    for sub, last in (("classification_subnet", num_classes), ("regression_subnet", 4)):
        for i in range(4):
            cn("%s.pre_conv.%d" % (sub, i), 3, 256)
        shapes[sub + ".out_conv.weight"] = (3, 3, 256, num_anchors * last)
        shapes[sub + ".out_conv.bias"] = (num_anchors * last,)
    return shapes


BACKBONE_TAP_CHANNELS = {"mobilenet_v2": (32, 96, 32)}


def init_params(backbone="mobilenet_v2", num_classes=80, num_anchors=9, seed=0):
    """Reference initialisation: heads/FPN N(0, 0.01) (retinanet.py:303);
    class out-conv bias -log(99) (retinanet.py:52-53); backbone variance-scaling(2.0,
    fan-in, normal) (mobilenet_v2.py:104-105); gamma 1, beta 0 (normalization.py:16-17).
    Plain (un-truncated) normals are used: the exact TF RNG stream is irreproducible, so
    only the distribution family matters."""
    g = torch.Generator().manual_seed(seed)
    if backbone != "mobilenet_v2":
        raise NotImplementedError(backbone)
    shapes = mobilenet_v2_param_shapes()
    c3, c4, c5 = BACKBONE_TAP_CHANNELS[backbone]
    head_shapes = fpn_head_param_shapes(c3, c4, c5, num_anchors, num_classes)
    params = {}
    for name, shape in list(shapes.items()) + list(head_shapes.items()):
        if name.endswith(".gamma"):
            params[name] = torch.ones(shape)
        elif name.endswith(".beta"):
            params[name] = torch.zeros(shape)
        elif name.endswith(".bias"):
            fill = -math.log(99.0) if name.startswith("classification_subnet") else 0.0
            params[name] = torch.full(shape, fill)
        elif name.startswith("backbone"):
            kh, kw, cin, _ = shape
            params[name] = torch.randn(shape, generator=g) * T.he_fan_in_std(kh, kw, cin)
        else:
            params[name] = torch.randn(shape, generator=g) * 0.01
    return params


def retinanet_forward(p, image, num_classes, num_anchors=9, act="elu", backbone="mobilenet_v2", dropout=None):
    """RetinaNetBase.call retinanet.py:272-296: same subnet weights for all 5 levels.  `dropout`: the backbone's dropout
    hook (retinanet.py:12-21 hands dropout_rate to the backbone only: FPN and subnets have no dropout)."""
    if backbone != "mobilenet_v2":
        raise NotImplementedError(backbone)
    feats = mobilenet_v2_forward(p, image, act, dropout=dropout)
    pyr = fpn_forward(p, feats, act)
    cls = {k: subnet_forward(p, v, "classification_subnet", num_anchors, num_classes, act)
           for k, v in pyr.items()}
    reg = {k: subnet_forward(p, v, "regression_subnet", num_anchors, 4, act)
           for k, v in pyr.items()}
    return {"classifications": cls, "regressions": reg}


def l2_regularization(p):
    """tf.contrib.layers.l2_regularizer(scale)(w) = scale * sum(w^2)/2 on conv kernels only
    (retinanet.py:304 scale 1e-4; mobilenet_v2.py:106-108 scale 4e-5 incl. depthwise)."""
    total = torch.zeros(())
    for name, w in p.items():
        if name.endswith(".weight"):
            scale = 4e-5 if name.startswith("backbone") else 1e-4
            total = total + scale * 0.5 * (w.double() ** 2).sum().float()
    return total
