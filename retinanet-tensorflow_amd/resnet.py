"""ResNeXt-50 bottom-up network, CLI name 'resnet_50' (drop-in for reference resnet.py:15-215),
GroupNorm variant, NHWC fp32 on gfx950 kernels.

    net = ResNeXt_50(activation=layers.elu)     # `activation` is accepted and ignored exactly as in the
    feats = net(image, training=True)           # reference: ReLU is hard-coded (resnet.py:85,93,101,157, Q4)

Structure (resnet.py:162-215): 7x7/2 conv -> GN -> ReLU (C1); 3x3/2 max pool; four stages of
bottlenecks (filters 64/128/256/512, depths 3/4/6/3), widths 256/512/1024/2048 (C2..C5).
Bottleneck (resnet.py:29-103): 1x1 -> 2f, GN, ReLU; cardinality-32 3x3 (stride 2 on 'down');
1x1 -> 4f, GN; + identity (1x1 projection, or a 3x3/2 conv on 'down', or the input); ReLU.

MI355X-first restatement with identical results: the reference's 32 separate Conv2D + 32 separate
GroupNorm on the channel splits (resnet.py:53-64,88-95) are ONE grouped-conv launch + ONE GroupNorm
launch: a split has c = 2f/32 <= 32 channels, so its GroupNorm uses min(32, c) = c groups of one
channel (normalization.py:24) -- per-channel statistics -- and the 32 per-split gamma/beta vectors
concatenate into one [2f] vector.  The final `GN(conv_3) + identity -> ReLU` is one fused kernel.
"""
import layers as L
from model import Model
from normalization import Normalization

CARDINALITY = 32


class ResNeXt_Bottleneck(Model):
    def __init__(self, filters, project, kernel_initializer, kernel_regularizer, cardinality=CARDINALITY,
                 name='resnext_bottleneck', in_channels=None):
        assert filters % cardinality == 0
        assert project in [True, False, 'down']
        super().__init__(name=name)
        self._cfg = (filters, project, kernel_initializer, kernel_regularizer, cardinality)
        self._built = False
        if in_channels is not None:
            self.build(in_channels)

    def build(self, in_channels):
        f, project, init, reg, card = self._cfg

        def conv(cout, k, stride=1, cin=None, groups=1):
            return L.Conv2D(cout, k, stride, padding='same', use_bias=False, kernel_initializer=init,
                            kernel_regularizer=reg, in_channels=cin, groups=groups)

        if project == 'down':
            self._identity_conv, self._identity_bn = conv(f * 4, 3, 2, in_channels), Normalization(channels=f * 4)
        elif project:
            self._identity_conv, self._identity_bn = conv(f * 4, 1, 1, in_channels), Normalization(channels=f * 4)
        else:
            self._identity_conv = self._identity_bn = None
        self._conv_1, self._bn_1 = conv(f * 2, 1, 1, in_channels), Normalization(channels=f * 2)
        stride = 2 if project == 'down' else 1
        # the 32 split convs as one grouped conv; per-split GroupNorm == per-channel groups (see module doc)
        self._conv_2 = conv(f * 2, 3, stride, f * 2, groups=card)
        self._bn_2 = L.GroupNormalization(groups=f * 2, channels=f * 2)
        self._conv_3, self._bn_3 = conv(f * 4, 1, 1, f * 2), Normalization(channels=f * 4)
        self._built = True

    def call(self, input, training):
        if not self._built:
            self.build(input.shape[3])
            self.to(input.device)
        import ops
        if input.dtype == L.torch.float16 and not training:
            out = self._call_f16_folded(input)
            if out is not None:
                return out
        input, identity = ops.fanout(input, 2) if input.dtype != L.torch.float16 else (input, input)   # two consumers: one summed gradient
        if self._identity_conv is not None:
            identity = self._identity_bn.fused(self._identity_conv(identity), training)
        x = self._bn_1.fused(self._conv_1(input), training, act='relu')
        x = self._bn_2.fused(self._conv_2(x), training, act='relu')
        return self._bn_3.fused(self._conv_3(x), training, act='relu', residual=identity, act_after_residual=True)


    def _call_f16_folded(self, input):
        """fp16 inference with the block's GroupNorms folded into its convs (ops_f16.conv2d_norm): every conv's epilogue emits
        the statistics of its output, conv 2 and conv 3 apply GroupNorm + ReLU to their operand on load, and only the block's
        output -- ReLU(GN(conv 3) + identity), read by the next block twice -- is written by an apply pass.  None: a shape
        that cannot fold (the caller takes the layer-by-layer path)."""
        import ops_f16
        if not ops_f16.FOLD:
            return None
        stride = self._conv_2.strides
        p1 = ops_f16.conv2d_norm(input, self._conv_1.weight, self._bn_1, act='relu')
        if p1 is None:
            return None
        # (the 3 x 3 conv reads every input element through nine taps: applying the GroupNorm on load would repeat its
        # arithmetic nine times -- measured 232 vs ~125 + 60 us per block at cfg 5 -- so its input is materialised once)
        # (... except where the grouped conv runs on LDS-resident input patches: there the GroupNorm is applied once per patch
        # element on the way into LDS and conv 1's apply pass disappears too)
        fold_in = ops_f16.FOLD_INTO_3X3 or ops_f16.sg_kernel_takes(p1.y.shape, self._conv_2.weight, stride, self._conv_2.groups)
        a1 = p1 if fold_in else p1.materialise()
        p2 = ops_f16.conv2d_norm(a1, self._conv_2.weight, self._bn_2, act='relu', stride=stride, groups=self._conv_2.groups)
        if p2 is None:
            return None
        p3 = ops_f16.conv2d_norm(p2, self._conv_3.weight, self._bn_3, act='relu')
        if p3 is None:
            return None
        identity = input
        if self._identity_conv is not None:
            pi = ops_f16.conv2d_norm(input, self._identity_conv.weight, self._identity_bn, act=None, stride=self._identity_conv.strides)
            # (the projection's GroupNorm has no activation: it is applied inside the block's final apply pass, never written)
            identity = pi if pi is not None else self._identity_bn.fused(self._identity_conv(input), False)
        return p3.materialise(residual=identity, act_after_residual=True)


class ResNeXt_Block(Model):
    def __init__(self, filters, depth, downsample, kernel_initializer, kernel_regularizer, name='resnext_block',
                 in_channels=None):
        super().__init__(name=name)
        blocks = []
        c = in_channels
        for i in range(depth):
            project = ('down' if downsample else True) if i == 0 else False
            blocks.append(ResNeXt_Bottleneck(filters, project=project, kernel_initializer=kernel_initializer,
                                             kernel_regularizer=kernel_regularizer, in_channels=c))
            c = filters * 4 if c is not None else None
        self._layers = blocks
        import torch
        self._mods = torch.nn.ModuleList(blocks)

    def call(self, input, training):
        for f in self._layers:
            input = f(input, training=training)
        return input


class ResNeXt_ConvInput(Model):
    def __init__(self, kernel_initializer, kernel_regularizer, name='resnext_conv1'):
        super().__init__(name=name)
        self._conv = L.Conv2D(64, 7, 2, padding='same', use_bias=False, kernel_initializer=kernel_initializer,
                              kernel_regularizer=kernel_regularizer, in_channels=3)
        self._bn = Normalization(channels=64)

    def call(self, input, training, pool=None):
        """`pool` (fp16 inference): the MaxPooling2D behind the stem -- conv with the GroupNorm statistics from its epilogue, then
        GroupNorm + ReLU + max pool in ONE pass over the conv output (ops_f16.max_pool_norm); returns the pooled tensor."""
        if L.INFERENCE_F16 and not training and input.dtype != L.torch.float16:
            import ops_f16
            input = ops_f16.image_to_half4(input)        # fp16 inference: the stem runs on the f16 matrix cores too
            if pool is not None and ops_f16.FOLD and self._conv.weight is not None:
                p = ops_f16.conv2d_norm(input, self._conv.weight, self._bn, 'relu', self._conv.strides, 1)
                if p is not None:
                    return ops_f16.max_pool_norm(p, pool.pool_size, pool.strides)
        out = self._bn.fused(self._conv(input), training, act='relu')
        return pool(out) if pool is not None else out


class ResNeXt(Model):
    def __init__(self, kernel_initializer, kernel_regularizer, name='resnext'):
        super().__init__(name=name)
        self._kernel_initializer = kernel_initializer
        self._kernel_regularizer = kernel_regularizer
        # train.Trainer installs a callable here: stage_cut(stage module, its input, names of the taps made so far) -> the
        # tensor the stage reads (a detached leaf: the backward pass then runs stage by stage, each stage's gradient
        # all-reduce underneath the stages below it)
        self.stage_cut = None

    def call(self, input, training):
        out = {}
        if L.INFERENCE_F16 and not training:
            # (fp16 inference: stem and pool as one normalise-and-pool pass; C1 itself -- which no caller of an inference pass
            # reads -- is not materialised)
            input = self._conv_1(input, training=training, pool=self._conv_1_max_pool)
            out['C1'] = None            # (the key the other paths and the reference return, resnet.py:203: present, not materialised here)
        else:
            input = self._conv_1(input, training=training)
            out['C1'] = input
            input = self._conv_1_max_pool(input)
        import ops
        for i, stage in enumerate((self._conv_2, self._conv_3, self._conv_4, self._conv_5)):
            if i > 0 and self.stage_cut is not None and training and L.torch.is_grad_enabled():
                input = self.stage_cut(stage, input, tuple(out.keys()))
            input = stage(input, training=training)
            if i in (1, 2) and input.dtype == L.torch.float32:      # C3, C4 feed the next stage AND the pyramid: one summed gradient
                out['C%d' % (i + 2)], input = ops.fanout(input, 2)
            else:
                out['C%d' % (i + 2)] = input
        return out


class ResNeXt_50(ResNeXt):
    def __init__(self, activation, kernel_initializer=None, kernel_regularizer=None, name='resnext_v2_50'):
        if kernel_initializer is None:
            kernel_initializer = L.VarianceScaling(factor=2.0)
        if kernel_regularizer is None:
            kernel_regularizer = L.L2Regularizer(scale=1e-4)
        super().__init__(kernel_initializer=kernel_initializer, kernel_regularizer=kernel_regularizer, name=name)
        init, reg = kernel_initializer, kernel_regularizer
        self._conv_1 = ResNeXt_ConvInput(kernel_initializer=init, kernel_regularizer=reg)
        self._conv_1_max_pool = L.MaxPooling2D(3, 2, padding='same')
        self._conv_2 = ResNeXt_Block(64, depth=3, downsample=False, kernel_initializer=init, kernel_regularizer=reg,
                                     in_channels=64)
        self._conv_3 = ResNeXt_Block(128, depth=4, downsample=True, kernel_initializer=init, kernel_regularizer=reg,
                                     in_channels=256)
        self._conv_4 = ResNeXt_Block(256, depth=6, downsample=True, kernel_initializer=init, kernel_regularizer=reg,
                                     in_channels=512)
        self._conv_5 = ResNeXt_Block(512, depth=3, downsample=True, kernel_initializer=init, kernel_regularizer=reg,
                                     in_channels=1024)
        self.out_channels = {'C3': 512, 'C4': 1024, 'C5': 2048}
