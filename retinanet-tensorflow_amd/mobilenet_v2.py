"""MobileNetV2 bottom-up network (drop-in for reference mobilenet_v2.py:15-223), GroupNorm
variant, NHWC fp32 on gfx950 kernels.

    net = MobileNetV2(activation=layers.elu, dropout_rate=0.2)
    feats = net(image, training=True)      # {'C1'..'C5'}, strides 2..32

Block table (filters, expansion, stride) and the taps follow mobilenet_v2.py:120-223:
stem 3x3/2 -> 32; 17 inverted-residual bottlenecks (expand 1x1 -> depthwise 3x3 -> linear 1x1,
each + GroupNorm, dropout; residual iff shapes match, :91-92); final 1x1 -> 32 channels (:178-185,
SURVEY Q8).  Each [conv, Normalization, activation, Dropout] run is conv kernel + ONE fused
GroupNorm kernel (the residual add rides in the same kernel).
"""
import os

import torch

import layers as L
import ops
import ops_mb
from model import Model, Sequential
from normalization import Normalization

# stage -> (filters, expansion, [strides of its blocks])
_STAGES = (
    (1, 16, 1, (1,)),
    (2, 24, 6, (2, 1)),
    (3, 32, 6, (2, 1, 1)),
    (4, 64, 6, (2, 1, 1, 1)),
    (5, 96, 6, (1, 1, 1)),
    (6, 160, 6, (2, 1, 1)),
    (7, 320, 6, (1,)),
)
_TAP_AFTER = {'bottleneck_1_1': 'C1', 'bottleneck_2_2': 'C2', 'bottleneck_3_3': 'C3', 'bottleneck_5_3': 'C4'}

# The bottlenecks as ONE autograd node over the rn_mb_* kernels (ops_mb.mb_chain): every GroupNorm applied by its consumer,
# 3 launches per bottleneck forward and 3 backward instead of 6 + 7.  It starts at the first bottleneck from which on every
# layer's shape qualifies (maps whose statistic rows a consumer block can merge: <= 128 x 128 at the headline size); the
# bottlenecks in front of it run layer by layer.  RN_MB_CHAIN=0: layer by layer everywhere.
MB_CHAIN = os.environ.get("RN_MB_CHAIN", "1") == "1"
# Maps of more than this many pixels per sample stay on the layer-by-layer path; 0 = no limit (the default since the
# depthwise kernels walk several tiles per block on big maps: 427 vs 423 images/s with the 256 x 256 maps in the chain; their
# statistic rows go through rn_mb_compact_rows first).  RN_MB_CHAIN_MAX_HW=16384: the round-3 first version (chain from
# bottleneck_2_2 on).
MB_CHAIN_MAX_HW = int(os.environ.get("RN_MB_CHAIN_MAX_HW", 0))
STAGE_CUT_AFTER = 'bottleneck_3_3'      # (the C3 tap: everything above it is 95 % of the backbone's gradient bytes and runs on maps <= 32 x 32)


class DepthwiseConv2D(L.DepthwiseConv2D):
    pass


class Bottleneck(Model):
    def __init__(self, filters, strides, expansion_factor, activation, dropout_rate, kernel_initializer,
                 kernel_regularizer, name='bottleneck', in_channels=None):
        super().__init__(name=name)
        self._cfg = (filters, strides, expansion_factor, L.get_activation(activation), dropout_rate,
                     kernel_initializer, kernel_regularizer)
        self.expand_conv = self.depthwise_conv = self.linear_conv = None
        if in_channels is not None:
            self.build(in_channels)

    def build(self, in_channels):
        filters, strides, t, act, rate, init, reg = self._cfg
        wide = in_channels * t

        def pointwise(cout, cin):
            return L.Conv2D(cout, 1, use_bias=False, kernel_initializer=init, kernel_regularizer=reg, in_channels=cin)

        self.expand_conv = Sequential([pointwise(wide, in_channels), Normalization(channels=wide), act, L.Dropout(rate)])
        self.depthwise_conv = Sequential([
            DepthwiseConv2D(3, strides=strides, padding='same', use_bias=False, kernel_initializer=init,
                            kernel_regularizer=reg, in_channels=wide),
            Normalization(channels=wide), act, L.Dropout(rate)])
        self.linear_conv = Sequential([pointwise(filters, wide), Normalization(channels=filters), L.Dropout(rate)])
        self._same_shape = (strides == 1 and filters == in_channels)

    def call(self, input, training):
        if self.expand_conv is None:
            self.build(input.shape[3])
            self.to(input.device)
        identity = None
        if self._same_shape:        # the input feeds the expand conv and the residual: their gradients are summed by our kernel
            input, identity = ops.fanout(input, 2)
        mid = self._fused_middle(input, training)
        if mid is None:
            mid = self.depthwise_conv(self.expand_conv(input, training), training)
        return self.linear_conv(mid, training, residual=identity)

    def _fused_middle(self, input, training):
        """[GroupNorm, act, Dropout] -> depthwise 3x3 -> [GroupNorm, act, Dropout] as ONE kernel (ops.dw_gn_fused) when a
        (sample, group) slice fits a CU's LDS; None otherwise (the layers then run one by one)."""
        conv, norm1, act1, drop1 = self.expand_conv.layers
        dw, norm2, act2, drop2 = self.depthwise_conv.layers
        if not torch.is_tensor(input) or input.dtype != torch.float32 or (L.INFERENCE_F16 and not training):
            return None
        c = conv.filters
        g = ops.gn_groups(c, norm1.groups)
        act = L.activation_name(act1)
        need_bwd = training and torch.is_grad_enabled()
        shape = (input.shape[0], input.shape[1], input.shape[2], c)                 # the expand conv is 1x1 / stride 1
        if (not input.is_cuda or dw.weight is None or norm1.gamma is None or norm2.gamma is None or
                norm1.groups != norm2.groups or norm1.eps != norm2.eps or act != L.activation_name(act2) or
                not ops.dw_gn_ok(shape, dw.weight, dw.strides, g, act, need_bwd)):
            return None
        rate = drop1.rate if (training and drop1.rate > 0.0) else 0.0
        y1 = conv(input)
        return ops.dw_gn_fused(y1, norm1.gamma, norm1.beta, dw.weight, norm2.gamma, norm2.beta, dw.strides, g, norm1.eps, act,
                               rate, drop1.seed, drop2.seed, L.Dropout.seed_device_counter)


class MobileNetV2(Model):
    def __init__(self, activation, dropout_rate, name='mobilenet_v2'):
        super().__init__(name=name)
        act = L.relu6 if activation is None else L.get_activation(activation)
        init = L.VarianceScaling(factor=2.0)
        reg = L.L2Regularizer(scale=4e-5)

        def conv_block(cout, k, stride, cin):
            return Sequential([
                L.Conv2D(cout, k, strides=stride, padding='same', use_bias=False, kernel_initializer=init,
                         kernel_regularizer=reg, in_channels=cin),
                Normalization(channels=cout), act, L.Dropout(dropout_rate)])

        self.input_conv = conv_block(32, 3, 2, 3)
        self.block_names = []
        channels = 32
        for stage, filters, t, strides in _STAGES:
            for i, s in enumerate(strides):
                name = 'bottleneck_%d_%d' % (stage, i + 1)
                setattr(self, name, Bottleneck(filters, strides=s, expansion_factor=t, activation=act,
                                               dropout_rate=dropout_rate, kernel_initializer=init,
                                               kernel_regularizer=reg, name=name, in_channels=channels))
                self.block_names.append(name)
                channels = filters
        self.output_conv = conv_block(32, 1, 1, channels)
        self.out_channels = {'C3': 32, 'C4': 96, 'C5': 32}
        # train.Trainer installs a callable here while a collective is active (see resnet.ResNeXt.stage_cut): the backward pass
        # of the chain then runs in two parts, cut behind STAGE_CUT_AFTER, and the upper part's 7.3 MB of gradients (95 % of the
        # backbone's) are all-reduced underneath the lower part's backward pass on the large maps.  The cut costs one identity
        # 1x1 conv forward and backward on the C3 map (the chain kernels apply a GroupNorm in their CONSUMER: the first half
        # needs one) -- about 25 us per step, so it is taken only where there is a collective to hide.
        self.stage_cut = None
        self.stage_cut_needs_collective = True

    def _chain_blocks(self, first):
        blocks = []
        for name in self.block_names[first:]:
            b = getattr(self, name)
            conv1, norm1, act1, drop1 = b.expand_conv.layers
            dw, norm2, act2, drop2 = b.depthwise_conv.layers
            conv3, norm3, drop3 = b.linear_conv.layers
            blocks.append(ops_mb.Block(
                conv1.weight, ops_mb.Norm(norm1.gamma, norm1.beta, norm1.groups, norm1.eps, L.activation_name(act1), drop1.rate, drop1.seed),
                dw.weight, ops_mb.Norm(norm2.gamma, norm2.beta, norm2.groups, norm2.eps, L.activation_name(act2), drop2.rate, drop2.seed),
                conv3.weight, ops_mb.Norm(norm3.gamma, norm3.beta, norm3.groups, norm3.eps, None, drop3.rate, drop3.seed),
                dw.strides, b._same_shape))
        return blocks

    def _identity(self, c, device):
        """[1, 1, c, c] identity kernel (a constant, not a parameter) for the stage cut's pass-through conv."""
        cache = self.__dict__.setdefault('_eye_cache', {})
        key = (c, str(device))
        if key not in cache:
            cache[key] = torch.eye(c, dtype=torch.float32, device=device).reshape(1, 1, c, c).contiguous()
        return cache[key]

    def _chain_start(self, x, training):
        """Index of the first bottleneck the fused chain runs (None: no chain) for the stem output x."""
        if not (MB_CHAIN and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32) or (L.INFERENCE_F16 and not training):
            return None
        key = tuple(x.shape)
        cache = self.__dict__.setdefault('_chain_cache', {})
        if key not in cache:
            n, h, w, c = key
            shapes = []
            for name in self.block_names:
                b = getattr(self, name)
                shapes.append((n, h, w, c))
                s = b.depthwise_conv.layers[0].strides
                h, w, c = -(-h // s), -(-w // s), b.linear_conv.layers[0].filters
            start = None
            for i in range(len(self.block_names) - 1):
                if MB_CHAIN_MAX_HW and shapes[i][1] * shapes[i][2] > MB_CHAIN_MAX_HW:
                    continue
                if ops_mb.chain_supported(shapes[i], self._chain_blocks(i)):
                    start = i
                    break
            cache[key] = start
        return cache[key]

    def call(self, input, training):
        out = {}
        input = self.input_conv(input, training)
        start = self._chain_start(input, training)
        cut_at = self.block_names[self.block_names.index(STAGE_CUT_AFTER) + 1]
        for name in (self.block_names if start is None else self.block_names[:start]):
            if name == cut_at and self.stage_cut is not None and training and torch.is_grad_enabled():
                input = self.stage_cut(getattr(self, name), input, tuple(out.keys()))      # (layer by layer: the cut costs nothing)
            input = getattr(self, name)(input, training)
            if name in _TAP_AFTER:      # a tap feeds the next block and (C3, C4) the pyramid
                out[_TAP_AFTER[name]], input = ops.fanout(input, 2)
        conv_o, norm_o, act_o, drop_o = self.output_conv.layers
        if start is None:
            out['C5'] = self.output_conv(input, training)
            return out
        names = self.block_names[start:]
        blocks = self._chain_blocks(start)
        seed_dev = L.Dropout.seed_device_counter
        if (self.stage_cut is not None and training and torch.is_grad_enabled() and STAGE_CUT_AFTER in names[:-1]):
            k = names.index(STAGE_CUT_AFTER) + 1          # first half: names[:k], ends with the identity conv; second half: names[k:]
            c_mid = blocks[k - 1].w3.shape[3]
            eye = self._identity(c_mid, input.device)
            tap_a = [i for i, name in enumerate(names[:k]) if name in _TAP_AFTER]
            taps, y_mid = ops_mb.mb_chain(input, blocks[:k], eye, tap_a, training=training, seed_dev=seed_dev, tail_const=True)
            for i, t in zip(tap_a, taps):
                out[_TAP_AFTER[names[i]]] = t
            x_mid = self.stage_cut(getattr(self, names[k]), y_mid, tuple(out.keys()))
            input, blocks, names = x_mid, blocks[k:], names[k:]
        tap_after = [i for i, name in enumerate(names) if name in _TAP_AFTER]
        taps, y_tail = ops_mb.mb_chain(input, blocks, conv_o.weight, tap_after, training=training, seed_dev=seed_dev)
        for i, t in zip(tap_after, taps):
            out[_TAP_AFTER[names[i]]] = t
        # the output block's own GroupNorm + activation + dropout (its 1x1 conv ran as the chain's tail)
        out['C5'] = norm_o.fused(y_tail, training, act=L.activation_name(act_o), dropout=drop_o)
        return out
