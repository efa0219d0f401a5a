"""DenseNet-BC-121 / -169 bottom-up networks (drop-in for reference densenet.py:26-310), GroupNorm
variant, NHWC fp32 on gfx950 kernels.

    net = DenseNetBC_121(activation=layers.elu, dropout_rate=0.2)
    feats = net(image, training=True)      # {'C1'..'C5'}

Structure (densenet.py:154-262): 7x7/2 conv (2k filters) -> GN -> act (C1); 3x3/2 max pool; dense
blocks of BottleneckCompositeFunction = GN-act-1x1(4k)-drop-GN-act-3x3(k)-drop whose output is
concatenated to its input (densenet.py:50-80,117-121); TransitionLayer = GN-1x1(C/2)-drop-avgpool2
with no activation (densenet.py:124-151).  k = 32; blocks 6/12/24/16 (121) or 6/12/32/32 (169).
[GN, act] runs are one fused kernel (model.Sequential); dropout after a conv is the stand-alone
counter-based dropout kernel.
"""
import torch

import layers as L
from model import Model, Sequential
from normalization import Normalization

Dropout = L.Dropout
CONCAT_FREE = True      # dense blocks on one buffer per block (ops.dense_block) instead of a concat per layer


class CompositeFunction(Sequential):
    def __init__(self, filters, activation, dropout_rate, kernel_initializer, kernel_regularizer,
                 name='composite_function', in_channels=None):
        layers = [
            Normalization(channels=in_channels), L.get_activation(activation),
            L.Conv2D(filters, 3, padding='same', use_bias=False, kernel_initializer=kernel_initializer,
                     kernel_regularizer=kernel_regularizer, in_channels=in_channels),
            Dropout(dropout_rate),
        ]
        super().__init__(layers, name=name)


class BottleneckCompositeFunction(Sequential):
    def __init__(self, filters, activation, dropout_rate, kernel_initializer, kernel_regularizer,
                 name='bottleneck_composite_function', in_channels=None):
        act = L.get_activation(activation)
        layers = [
            Normalization(channels=in_channels), act,
            L.Conv2D(filters * 4, 1, use_bias=False, kernel_initializer=kernel_initializer,
                     kernel_regularizer=kernel_regularizer, in_channels=in_channels),
            Dropout(dropout_rate),
            Normalization(channels=filters * 4), act,
            L.Conv2D(filters, 3, padding='same', use_bias=False, kernel_initializer=kernel_initializer,
                     kernel_regularizer=kernel_regularizer, in_channels=filters * 4),
            Dropout(dropout_rate),
        ]
        super().__init__(layers, name=name)


class DenseNet_Block(Model):
    def __init__(self, growth_rate, depth, bottleneck, activation, dropout_rate, kernel_initializer,
                 kernel_regularizer, name='densnet_block', in_channels=None):
        super().__init__(name=name)
        fn = BottleneckCompositeFunction if bottleneck else CompositeFunction
        self.composite_functions = []
        c = in_channels
        for i in range(depth):
            self.composite_functions.append(fn(growth_rate, activation=activation, dropout_rate=dropout_rate,
                                               kernel_initializer=kernel_initializer,
                                               kernel_regularizer=kernel_regularizer,
                                               name='composite_function{}'.format(i + 1), in_channels=c))
            c = c + growth_rate if c is not None else None
        self._fns = torch.nn.ModuleList(self.composite_functions)
        self.out_channels = c

    def call(self, input, training):
        out = self._concat_free(input, training)
        if out is not None:
            return out
        for f in self.composite_functions:               # (fp16 inference / non-bottleneck blocks: layer by layer)
            output = f(input, training)
            input = torch.cat([input, output], -1)       # growth concat (densenet.py:119)
        return input

    def _concat_free(self, input, training):
        """The block on ONE pre-allocated [n,h,w,c_total] buffer (ops.dense_block): every layer normalises a channel prefix of
        it in place and writes its k channels into their slice -- no concat, no copy of the growing tensor."""
        import ops
        fns = self.composite_functions
        if (not CONCAT_FREE or not fns or not isinstance(fns[0], BottleneckCompositeFunction) or not input.is_cuda or
                input.dtype != torch.float32 or (L.INFERENCE_F16 and not training) or input.shape[3] % 4):
            return None
        layers, seeds = [], []
        for f in fns:
            n1, a1, c1, d1, n2, a2, c2, d2 = f.layers
            if c1.weight is None or c2.weight is None or n1.gamma is None or n2.gamma is None:
                return None
            layers.append((n1.gamma, n1.beta, c1.weight, n2.gamma, n2.beta, c2.weight))
            seeds.append((d1.seed, d2.seed))
        n1, a1, _c1, d1 = fns[0].layers[:4]
        rate = d1.rate if training else 0.0
        return ops.dense_block(input, layers, fns[0].layers[6].filters, n1.groups, n1.eps, L.activation_name(a1), rate, seeds,
                               L.Dropout.seed_device_counter if rate > 0.0 else None)


class TransitionLayer(Sequential):
    def __init__(self, input_filters, compression_factor, dropout_rate, kernel_initializer, kernel_regularizer,
                 name='transition_layer'):
        self.input_filters = input_filters
        filters = int(input_filters * compression_factor)
        layers = [
            Normalization(channels=input_filters),
            L.Conv2D(filters, 1, use_bias=False, kernel_initializer=kernel_initializer,
                     kernel_regularizer=kernel_regularizer, in_channels=input_filters),
            Dropout(dropout_rate),
            L.AveragePooling2D(2, 2, padding='same'),
        ]
        super().__init__(layers, name=name)
        self.filters = filters

    def call(self, input, training):
        assert input.shape[-1] == self.input_filters
        return super().call(input, training)


class DenseNetBC_ImageNet(Model):
    def __init__(self, blocks, growth_rate, compression_factor, bottleneck, activation, dropout_rate,
                 kernel_initializer, kernel_regularizer, name='densenet_bc_imagenet'):
        super().__init__(name=name)
        self.stage_cut = None       # installed by train.Trainer (see resnet.ResNeXt.stage_cut)
        act = L.get_activation(activation)
        common = dict(kernel_initializer=kernel_initializer, kernel_regularizer=kernel_regularizer)
        self.conv1 = Sequential([
            L.Conv2D(2 * growth_rate, 7, 2, padding='same', use_bias=False, in_channels=3, **common),
            Normalization(channels=2 * growth_rate), act])
        self.conv1_max_pool = L.MaxPooling2D(3, 2, padding='same')
        c = 2 * growth_rate
        self.out_channels = {}
        for i in range(1, 5):
            block = DenseNet_Block(growth_rate, depth=blocks[i], bottleneck=bottleneck, activation=act,
                                   dropout_rate=dropout_rate, name='dense_block%d' % i, in_channels=c, **common)
            setattr(self, 'dense_block_%d' % i, block)
            c = block.out_channels
            self.out_channels['C%d' % (i + 1)] = c
            if i < 4:
                t = TransitionLayer(input_filters=c, compression_factor=compression_factor,
                                    dropout_rate=dropout_rate, name='transition_layer_%d' % i, **common)
                setattr(self, 'transition_layer_%d' % i, t)
                c = t.filters

    def call(self, input, training):
        out = {}
        input = self.conv1(input, training)
        out['C1'] = input
        input = self.conv1_max_pool(input)
        import ops
        for i in range(1, 5):
            block = getattr(self, 'dense_block_%d' % i)
            if i > 2 and getattr(self, 'stage_cut', None) is not None and training and torch.is_grad_enabled():
                input = self.stage_cut(block, input, tuple(out.keys()))    # (see resnet.ResNeXt.stage_cut)
            input = block(input, training)
            if i in (2, 3) and input.dtype == torch.float32:        # C3, C4 feed the transition AND the pyramid: one summed gradient
                out['C%d' % (i + 1)], input = ops.fanout(input, 2)
            else:
                out['C%d' % (i + 1)] = input
            if i < 4:
                input = getattr(self, 'transition_layer_%d' % i)(input, training)
        return out


def _defaults(kernel_initializer, kernel_regularizer):
    if kernel_initializer is None:
        kernel_initializer = L.VarianceScaling(factor=2.0)
    if kernel_regularizer is None:
        kernel_regularizer = L.L2Regularizer(scale=1e-4)
    return kernel_initializer, kernel_regularizer


class DenseNetBC_121(DenseNetBC_ImageNet):
    def __init__(self, activation, dropout_rate, kernel_initializer=None, kernel_regularizer=None,
                 name='densenet_bc_121'):
        init, reg = _defaults(kernel_initializer, kernel_regularizer)
        super().__init__(blocks=[None, 6, 12, 24, 16], growth_rate=32, compression_factor=0.5, bottleneck=True,
                         activation=activation, dropout_rate=dropout_rate, kernel_initializer=init,
                         kernel_regularizer=reg, name=name)


class DenseNetBC_169(DenseNetBC_ImageNet):
    def __init__(self, activation, dropout_rate, kernel_initializer=None, kernel_regularizer=None,
                 name='densenet_bc_169'):
        init, reg = _defaults(kernel_initializer, kernel_regularizer)
        super().__init__(blocks=[None, 6, 12, 32, 32], growth_rate=32, compression_factor=0.5, bottleneck=True,
                         activation=activation, dropout_rate=dropout_rate, kernel_initializer=init,
                         kernel_regularizer=reg, name=name)
