"""Layers standing in for the tf.layers / tf.nn objects the reference builds its blocks from.

  Conv2D              tf.layers.Conv2D          (retinanet.py:39-46 ..., mobilenet_v2.py:57-59)
  DepthwiseConv2D     mobilenet_v2.py:15-38     (tf.nn.depthwise_conv2d)
  GroupNormalization  normalization.py:4-35
  Dropout             tf.layers.Dropout         (mobilenet_v2.py:62)
  elu / relu / relu6  tf.nn.elu / relu / relu6  (train.py:214, resnet.py:85, mobilenet_v2.py:102)
All tensors NHWC fp32 on the GPU; kernels HWIO; every op is a HIP kernel from librn_hip.so.
Every layer also accepts a LIST of tensors (pyramid levels sharing the layer): one launch.
"""
import math

import torch

import ops
import ops_f16

# Inference in fp16 storage (BASELINE configs[4]): set with layers.set_inference_dtype('f16').  When on, a
# forward with training=False keeps activations in fp16 after the first GroupNorm, convs run on the f16
# matrix cores (fp32 accumulate), GroupNorm statistics stay fp32; layers that must emit fp32 (the subnets'
# output convs) say so with `f16_out_f32`.  Training is always fp32.
INFERENCE_F16 = False
# ... and what the subnets' output convs emit then: fp16 (BASELINE configs[4] as stated: the logits / box deltas leave the
# net in fp16 and utils.detect_raw reads them as stored, 2 bytes per element) or fp32 (outputs='f32')
F16_OUTPUTS_F32 = False


def set_inference_dtype(name, outputs='f16'):
    global INFERENCE_F16, F16_OUTPUTS_F32
    assert name in ('f32', 'f16') and outputs in ('f32', 'f16')
    INFERENCE_F16 = (name == 'f16')
    F16_OUTPUTS_F32 = (outputs == 'f32')


# ------------------------------------------------------------------ activations
class _Activation(object):
    def __init__(self, name):
        self.rn_act = name

    def __call__(self, input):
        if isinstance(input, (list, tuple)):
            return [self(x) for x in input]
        if input.dtype == torch.float16:
            return ops_f16.activation(input, self.rn_act)
        return ops.activation(input, self.rn_act)

    def __repr__(self):
        return 'activation(%s)' % self.rn_act


elu = _Activation('elu')
relu = _Activation('relu')
relu6 = _Activation('relu6')
_BY_NAME = {'elu': elu, 'relu': relu, 'relu6': relu6}


def get_activation(activation):
    """Accepts 'elu' / 'relu' / 'relu6', one of the objects above, or None."""
    if activation is None or isinstance(activation, _Activation):
        return activation
    if isinstance(activation, str) and activation in _BY_NAME:
        return _BY_NAME[activation]
    raise AssertionError('unsupported activation {!r}: use layers.elu / relu / relu6'.format(activation))


def activation_name(obj):
    return obj.rn_act if isinstance(obj, _Activation) else None


def add(a, b):
    first = a[0] if isinstance(a, (list, tuple)) else a
    if first.dtype == torch.float16:
        return [x + y for x, y in zip(a, b)] if isinstance(a, (list, tuple)) else a + b
    return ops.add(a, b)


# ------------------------------------------------------------------ initialisers / regularisers
class RandomNormal(object):
    """tf.random_normal_initializer(mean, stddev) (retinanet.py:303)."""

    def __init__(self, mean=0.0, stddev=0.01):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, generator=None):
        return torch.randn(shape, generator=generator) * self.stddev + self.mean


class VarianceScaling(object):
    """tf.contrib.layers.variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False)
    (mobilenet_v2.py:104-105): truncated normal, stddev sqrt(1.3 * factor / fan_in)."""

    def __init__(self, factor=2.0):
        self.factor = factor

    def __call__(self, shape, generator=None):
        fan_in = shape[0] * shape[1] * shape[2]
        std = math.sqrt(1.3 * self.factor / fan_in)
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=generator)
        return t


class Constant(object):
    def __init__(self, value):
        self.value = value

    def __call__(self, shape, generator=None):
        return torch.full(shape, float(self.value))


class L2Regularizer(object):
    """tf.contrib.layers.l2_regularizer(scale): scale * sum(w^2) / 2 (retinanet.py:304)."""

    def __init__(self, scale):
        self.scale = float(scale)


# ------------------------------------------------------------------ layers
class Conv2D(torch.nn.Module):
    """tf.layers.Conv2D(filters, kernel_size, strides, padding='same', use_bias, ...).
    The kernel is created on the first call (input channels known then), like tf.layers."""

    def __init__(self, filters, kernel_size, strides=1, padding='same', use_bias=True, kernel_initializer=None,
                 kernel_regularizer=None, bias_initializer=None, in_channels=None, groups=1, name=None):
        super().__init__()
        assert padding == 'same', "the reference only uses padding='same'"
        self.filters, self.kernel_size, self.strides = filters, kernel_size, strides
        self.groups = groups        # > 1: the `groups` parallel convs of a ResNeXt bottleneck as ONE grouped conv
        self.f16_out_f32 = False    # fp16 inference: may emit fp32 (set on the subnets' output convs; layers.F16_OUTPUTS_F32)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer or VarianceScaling(1.0)
        self.bias_initializer = bias_initializer or Constant(0.0)
        self.l2_scale = kernel_regularizer.scale if kernel_regularizer is not None else 0.0
        self.weight = None
        self.bias = None
        if in_channels is not None:
            self.build(in_channels)

    def build(self, in_channels, device=None):
        k = self.kernel_size
        assert in_channels % self.groups == 0 and self.filters % self.groups == 0
        w = self.kernel_initializer((k, k, in_channels // self.groups, self.filters))
        self.weight = torch.nn.Parameter(w.to(device) if device is not None else w)
        self.weight.l2_scale = self.l2_scale
        if self.use_bias:
            b = self.bias_initializer((self.filters,))
            self.bias = torch.nn.Parameter(b.to(device) if device is not None else b)

    def forward(self, input, norm=None):
        """`norm`: the GroupNormalization layer applied to the output next (model.Sequential passes it): the conv kernel then
        also emits that layer's statistics where it can, and the GroupNorm only applies them."""
        first = input[0] if isinstance(input, (list, tuple)) else input
        if self.weight is None:
            self.build(first.shape[3], first.device)
        if first.dtype == torch.float16:
            return ops_f16.conv2d(input, self.weight, self.bias, self.strides, self.groups,
                                  out_f32=self.f16_out_f32 and F16_OUTPUTS_F32)
        gn = (norm.groups, norm.eps) if (norm is not None and self.bias is None and self.groups == 1) else None
        return ops.conv2d(input, self.weight, self.bias, self.strides, self.groups, gn=gn)


class DepthwiseConv2D(torch.nn.Module):
    """mobilenet_v2.py:15-38: kernel [k, k, C, 1], no bias."""

    def __init__(self, kernel_size, strides, padding, use_bias, kernel_initializer, kernel_regularizer,
                 name='separable_conv2d', in_channels=None):
        super().__init__()
        assert padding == 'same' and not use_bias
        self.kernel_size, self.strides = kernel_size, strides
        self.kernel_initializer = kernel_initializer
        self.l2_scale = kernel_regularizer.scale if kernel_regularizer is not None else 0.0
        self.weight = None
        if in_channels is not None:
            self.build(in_channels)

    def build(self, in_channels, device=None):
        k = self.kernel_size
        w = self.kernel_initializer((k, k, in_channels, 1))
        self.weight = torch.nn.Parameter(w.to(device) if device is not None else w)
        self.weight.l2_scale = self.l2_scale

    def forward(self, input, norm=None):
        if self.weight is None:
            self.build(input.shape[3], input.device)
        return ops.depthwise_conv2d(input, self.weight, self.strides, gn=(norm.groups, norm.eps) if norm is not None else None)


class Dropout(torch.nn.Module):
    """tf.layers.Dropout(rate): inverted dropout when training, identity otherwise.  The mask is
    counter-based (seed, element index), regenerated in backward -- never stored.  Normally
    fused into the preceding GroupNorm kernel by model.Sequential."""
    _next_seed = [0x5EED]
    seed_device_counter = None     # optional uint64 device tensor advanced once per step (train.py)

    def __init__(self, rate):
        super().__init__()
        self.rate = float(rate)
        Dropout._next_seed[0] += 0x9E3779B1
        self.seed = Dropout._next_seed[0]

    def forward(self, input, training):
        if not training or self.rate == 0.0:
            return input
        return ops.dropout(input, self.rate, self.seed, Dropout.seed_device_counter)


class MaxPooling2D(torch.nn.Module):
    """tf.layers.MaxPooling2D(pool_size, strides, padding='same') (resnet.py:200, densenet.py:180)."""

    def __init__(self, pool_size, strides, padding='same'):
        super().__init__()
        assert padding == 'same'
        self.pool_size, self.strides = pool_size, strides

    def forward(self, input):
        if input.dtype == torch.float16:
            return ops_f16.max_pool(input, self.pool_size, self.strides)
        return ops.max_pool(input, self.pool_size, self.strides)


class AveragePooling2D(torch.nn.Module):
    """tf.layers.AveragePooling2D(pool_size, strides, padding='same') (densenet.py:144)."""

    def __init__(self, pool_size, strides, padding='same'):
        super().__init__()
        assert padding == 'same'
        self.pool_size, self.strides = pool_size, strides

    def forward(self, input):
        return ops.avg_pool(input, self.pool_size, self.strides)


class GroupNormalization(torch.nn.Module):
    """normalization.py:4-35; gamma/beta [C] (the reference's [1,1,1,C])."""

    def __init__(self, groups=32, eps=1e-5, name='group_normalization', channels=None):
        super().__init__()
        self.groups, self.eps = groups, eps
        self.gamma = None
        self.beta = None
        if channels is not None:
            self.build(channels)

    def build(self, c, device=None):
        self.gamma = torch.nn.Parameter(torch.ones(c, device=device))
        self.beta = torch.nn.Parameter(torch.zeros(c, device=device))

    def fused(self, input, training, act=None, dropout=None, residual=None, act_after_residual=False):
        first = input[0] if isinstance(input, (list, tuple)) else input
        if self.gamma is None:
            self.build(first.shape[3], first.device)
        if (first.dtype == torch.float16 or INFERENCE_F16) and not training:
            return ops_f16.group_norm_act(input, self.gamma, self.beta, self.groups, self.eps, act, residual,
                                          act_after_residual)
        rate, seed, seed_dev = 0.0, 0, None
        if dropout is not None and training and dropout.rate > 0.0:
            rate, seed, seed_dev = dropout.rate, dropout.seed, Dropout.seed_device_counter
        return ops.group_norm_act(input, self.gamma, self.beta, self.groups, self.eps, act, residual, rate, seed,
                                  seed_dev, act_after_residual)

    def call(self, input):
        return self.fused(input, False)

    def forward(self, input, training=None):
        return self.call(input)
