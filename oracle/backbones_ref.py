"""Oracle: ResNeXt-50 ('resnet_50') and DenseNet-BC-121/169 forward on torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parameters are a flat dict keyed by the
product model's ``named_parameters()`` names with the leading ``base.`` removed.

Follows the reference LITERALLY where the product restructures:
  * ResNeXt bottleneck resnet.py:80-103: ``tf.split`` into 32 groups, a separate conv and a separate
    ``Normalization()`` (GroupNorm with min(32, c) groups, normalization.py:24) per split, ReLU, concat
    -- the product runs one grouped conv + one per-channel GroupNorm; this oracle proves they agree.
  * DenseNet block densenet.py:117-121: concat of the input and each composite function's output.
"""
import torch

from . import tf_ops_ref as T

RESNEXT_STAGES = (("_conv_2", 64, 3, False), ("_conv_3", 128, 4, True), ("_conv_4", 256, 6, True),
                  ("_conv_5", 512, 3, True))
CARDINALITY = 32


def _gn(p, name, x, groups=32):
    return T.group_norm(x, p[name + ".gamma"], p[name + ".beta"], groups)


def resnext_bottleneck(p, pre, x, project):
    """resnet.py:75-103."""
    identity = x
    if project == "down":
        identity = _gn(p, pre + "._identity_bn", T.conv2d_same(x, p[pre + "._identity_conv.weight"], 2))
    elif project:
        identity = _gn(p, pre + "._identity_bn", T.conv2d_same(x, p[pre + "._identity_conv.weight"], 1))
    y = torch.relu(_gn(p, pre + "._bn_1", T.conv2d_same(x, p[pre + "._conv_1.weight"], 1)))
    w2, g2, b2 = p[pre + "._conv_2.weight"], p[pre + "._bn_2.gamma"], p[pre + "._bn_2.beta"]
    cg = w2.shape[2]
    stride = 2 if project == "down" else 1
    outs = []
    for g, split in enumerate(torch.split(y, cg, dim=-1)):          # resnet.py:88-95
        sl = slice(g * cg, (g + 1) * cg)
        o = T.conv2d_same(split, w2[..., sl], stride)
        o = T.group_norm(o, g2[sl], b2[sl], groups=32)                # Normalization(): min(32, cg) groups
        outs.append(torch.relu(o))
    y = torch.cat(outs, -1)
    y = _gn(p, pre + "._bn_3", T.conv2d_same(y, p[pre + "._conv_3.weight"], 1))
    return torch.relu(y + identity)


def resnext50_forward(p, x, prefix="backbone"):
    """resnet.py:169-215."""
    out = {}
    x = torch.relu(_gn(p, prefix + "._conv_1._bn", T.conv2d_same(x, p[prefix + "._conv_1._conv.weight"], 2)))
    out["C1"] = x
    x = T.max_pool_same(x, 3, 2)
    for i, (name, _f, depth, down) in enumerate(RESNEXT_STAGES):
        for d in range(depth):
            project = ("down" if down else True) if d == 0 else False
            x = resnext_bottleneck(p, "%s.%s._mods.%d" % (prefix, name, d), x, project)
        out["C%d" % (i + 2)] = x
    return out


def densenet_forward(p, x, blocks, act="elu", prefix="backbone", dropout=None):
    """densenet.py:246-262.  Dropout sites (densenet.py:23 ``Dropout = tf.layers.Dropout``): after the 1x1 and after the
    3x3 conv of a BottleneckCompositeFunction (:67, :77) and after a TransitionLayer's conv (:143); site names are the
    product's module paths (``..._mods.2`` / ``..._mods.5``).  `dropout`: hook(site, x) -> x (oracle/dropout_ref.Sites),
    None = identity."""
    drop = dropout if dropout is not None else (lambda _site, t: t)
    out = {}
    x = T.conv2d_same(x, p[prefix + ".conv1._mods.0.weight"], 2)
    x = T.activation(_gn(p, prefix + ".conv1._mods.1", x), act)
    out["C1"] = x
    x = T.max_pool_same(x, 3, 2)
    for i in range(1, 5):
        for d in range(blocks[i]):
            f = "%s.dense_block_%d._fns.%d._mods" % (prefix, i, d)
            # BottleneckCompositeFunction: [GN, act, conv1x1, drop, GN, act, conv3x3, drop] -> mods 0,1,2(drop),3,4,5(drop)
            y = T.activation(_gn(p, f + ".0", x), act)
            y = drop(f + ".2", T.conv2d_same(y, p[f + ".1.weight"], 1))
            y = T.activation(_gn(p, f + ".3", y), act)
            y = drop(f + ".5", T.conv2d_same(y, p[f + ".4.weight"], 1))
            x = torch.cat([x, y], -1)
        out["C%d" % (i + 1)] = x
        if i < 4:
            t = "%s.transition_layer_%d._mods" % (prefix, i)
            x = _gn(p, t + ".0", x)
            x = drop(t + ".2", T.conv2d_same(x, p[t + ".1.weight"], 1))
            x = T.avg_pool_same(x, 2, 2)
    return out


DENSENET_BLOCKS = {"densenet_121": [None, 6, 12, 24, 16], "densenet_169": [None, 6, 12, 32, 32]}


def backbone_forward(name, p, x, act="elu", dropout=None):
    if name == "resnet_50":
        return resnext50_forward(p, x)           # no dropout in ResNeXt
    return densenet_forward(p, x, DENSENET_BLOCKS[name], act, dropout=dropout)
