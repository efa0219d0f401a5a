"""RetinaNet builder (drop-in for reference retinanet.py:12-316) on gfx950 kernels.

    net = RetinaNet(backbone='mobilenet_v2', levels=build_levels(), num_classes=80,
                    activation=layers.elu, dropout_rate=0.2)
    out = net(image_nhwc, training=True)
    out['classifications']['P3']   # [N, H/8, W/8, A, C]
    out['regressions']['P7']       # [N, H/128, W/128, A, 4]

Same constructor arguments, class names and return structure as the reference.  MI355X-first
differences that do not change results:
  * the class / box subnets are applied to all five pyramid levels in ONE launch per layer
    (the levels share the subnet's weights AND GroupNorm gamma/beta, retinanet.py:257-270,
    :283-291; GroupNorm statistics stay per level and per sample);
  * every [Conv2D, Normalization, activation] run is a conv kernel + one fused GroupNorm kernel;
  * FPN merge = one upsample+add kernel (nearest, align_corners, retinanet.py:153-157).
"""
import math

import os

import torch

import _rn
import layers as L
import mobilenet_v2
import ops
from model import Model, Sequential
from normalization import Normalization

BACKBONES = ['resnet_50', 'densenet_121', 'densenet_169', 'mobilenet_v2']

# The class and box towers have identical shapes (4 x [conv3x3 256->256, GN, act]): run them as ONE
# tower on 512 channels -- layer 1 a dense 256->512 conv (kernels concatenated along cout), layers 2-4
# grouped convs with 2 groups, GroupNorm with 64 groups of 8 channels (= 32 + 32) -- so every head
# launch carries twice the tiles (the per-level maps are small: 2.7 tiles per CU for one tower) and the
# number of launches halves.  Results are identical to running the subnets one after the other.
FUSE_HEAD_TOWERS = False   # measured: 163 vs 166 img/s (the fused wgrad is slower), kept as an option

# The class and box subnets are independent given the pyramid: run them on two HIP streams so their
# (small, launch- and occupancy-bound) kernels overlap -- forward here, and backward too because
# autograd replays each node on its forward stream.  Captured into the step's hipGraph as two branches.
# Measured on MI355X with the Winograd towers (4 short kernels per conv): 256 vs 249 img/s.
HEADS_TWO_STREAMS = os.environ.get("RN_HEADS_TWO_STREAMS", "1") == "1"


# the two independent chains of the FPN (P6 -> P7 | P5 -> P4 -> P3) on two streams
FPN_TWO_STREAMS = os.environ.get("RN_FPN_TWO_STREAMS", "1") == "1"   # measured: 363.7 vs 358.3 images/s


def side_stream(device):
    return _rn.side_stream(device, 1)


HEADS_OFFSET_US = float(os.environ.get("RN_HEADS_OFFSET_US", "0"))
HEADS_BOX_FIRST = os.environ.get("RN_HEADS_BOX_FIRST", "0") == "1"
# fp16 inference: GroupNorm statistics of the head towers from the conv epilogues on levels with at least this many rows (n h w)
F16_HEAD_STATS = os.environ.get("RN_F16_HEAD_STATS", "0") == "1"      # measured: 678 vs 680 images/s at cfg 5 -- off (the two subnets' streams already hide the pass)
F16_HEAD_STATS_MIN_ROWS = int(os.environ.get("RN_F16_HEAD_STATS_MIN_ROWS", "32768"))
_delay_buf = {}


def _delay(device, stream, us):
    """One workgroup that sleeps ~`us` microseconds on `stream` (rn_debug_collective_standin with a 4 KB copy)."""
    b = _delay_buf.get(device)
    if b is None:
        b = _delay_buf[device] = torch.zeros(2048, dtype=torch.float32, device=device)
    import ctypes
    _rn.check(_rn.lib().rn_debug_collective_standin(b[:1024].data_ptr(), b[1024:].data_ptr(), 4096, 1, float(us),
                                                    ctypes.c_void_p(stream.cuda_stream)), "rn_debug_collective_standin")


def build_backbone(backbone, activation, dropout_rate):
    assert backbone in BACKBONES
    if backbone == 'mobilenet_v2':
        return mobilenet_v2.MobileNetV2(activation=activation, dropout_rate=dropout_rate)
    if backbone == 'resnet_50':
        import resnet
        return resnet.ResNeXt_50(activation=activation)
    import densenet
    if backbone == 'densenet_121':
        return densenet.DenseNetBC_121(activation=activation, dropout_rate=dropout_rate)
    return densenet.DenseNetBC_169(activation=activation, dropout_rate=dropout_rate)


def _conv_norm(filters, kernel_size, strides, kernel_initializer, kernel_regularizer, in_channels=None, pre=None,
               post=None):
    """[pre?] Conv2D(no bias) -> Normalization [-> post?]"""
    seq = [] if pre is None else [pre]
    seq += [L.Conv2D(filters, kernel_size, strides, padding='same', use_bias=False,
                     kernel_initializer=kernel_initializer, kernel_regularizer=kernel_regularizer,
                     in_channels=in_channels),
            Normalization(channels=filters)]
    if post is not None:
        seq.append(post)
    return Sequential(seq)


class _Subnet(Model):
    """Four [conv3x3 256, GN, act] blocks and a biased conv3x3 producing num_anchors * last_dim
    maps, reshaped to [N, H, W, A, last_dim] (retinanet.py:37-71 / :85-115)."""

    def __init__(self, num_anchors, last_dim, activation, kernel_initializer, kernel_regularizer,
                 bias_initializer, name):
        super().__init__(name=name)
        self.num_anchors, self.last_dim = num_anchors, last_dim
        act = L.get_activation(activation)
        self.pre_conv = Sequential([
            _conv_norm(256, 3, 1, kernel_initializer, kernel_regularizer, in_channels=256, post=act)
            for _ in range(4)])
        self.out_conv = L.Conv2D(num_anchors * last_dim, 3, 1, padding='same', kernel_initializer=kernel_initializer,
                                 kernel_regularizer=kernel_regularizer, bias_initializer=bias_initializer,
                                 in_channels=256)
        self.out_conv.f16_out_f32 = True     # fp16 inference: logits / box deltas leave the net in fp16, or fp32 on request

    def _reshape(self, t):
        return t.reshape(t.shape[0], t.shape[1], t.shape[2], self.num_anchors, self.last_dim)

    def _folded(self, maps, training):
        """The subnet as GroupNorm-folded Winograd layers (ops.wino_tower): no GroupNorm kernels, the normalised tensors are
        never written.  None when the shapes / dtype do not qualify (then the layers run one by one)."""
        if L.INFERENCE_F16 and not training:
            return None
        blocks = [blk.layers for blk in self.pre_conv.layers]          # [conv, norm, act] each
        if any(b[0].weight is None or b[1].gamma is None for b in blocks) or self.out_conv.weight is None:
            return None
        tower = [(b[0].weight, b[1].gamma, b[1].beta) for b in blocks]
        norm, act = blocks[0][1], L.activation_name(blocks[0][2])
        if ops.wino_tower_ok(maps, tower, self.out_conv.weight, norm.groups):
            return ops.wino_tower(maps, tower, self.out_conv.weight, self.out_conv.bias, norm.groups, norm.eps, act)
        if ops.wino_tower_ok(maps, tower, None, norm.groups):
            # the output conv cannot run as a Winograd layer (cout % 4 != 0): the last GroupNorm is materialised
            raw = ops.wino_tower(maps, tower, None, None, norm.groups, norm.eps, act)
            return self.out_conv(blocks[-1][1].fused(raw, training, act=act))
        return None

    def tower_kernels(self):
        """The 3 x 3 kernels the folded Winograd path transforms (the four tower convs and the output conv), or [] while the
        layers are not built / the path is off: train.Trainer transforms them once per step ahead of the layers
        (ops.WinoPretransform)."""
        blocks = [blk.layers for blk in self.pre_conv.layers]
        ws = [b[0].weight for b in blocks] + [self.out_conv.weight]
        if any(w is None or not w.is_cuda or w.dtype != torch.float32 or w.shape[0] != 3 or w.shape[1] != 3 or w.shape[2] % 64 or w.shape[3] % 4
               for w in ws):
            return []
        return ws

    def _f16_tower_levels(self, maps):
        """fp16 inference: the four [conv, GroupNorm, act] blocks on ALL levels, three launches per block -- the conv of every level
        with the GroupNorm statistics from its epilogue, finalise, apply (ops_f16.conv_norm_act_levels): no statistics pass over the
        conv outputs.  None when a block does not qualify (then the layers run one by one)."""
        import ops_f16
        blocks = [blk.layers for blk in self.pre_conv.layers]          # [conv, norm, act] each
        if any(b[0].weight is None or b[0].bias is not None for b in blocks) or not all(m.dtype == torch.float16 for m in maps):
            return None
        cur = list(maps)
        for conv, norm, act in blocks:
            cur = ops_f16.conv_norm_act_levels(cur, conv.weight, norm, L.activation_name(act))
            if cur is None:
                return None
        return cur

    def _f16_tower(self, maps):
        """fp16 inference: the four [conv, GroupNorm, act] blocks with the GroupNorm STATISTICS taken from the conv's epilogue
        (ops_f16.conv2d_norm: no statistics pass over the conv output) on the levels where that pass costs -- the large maps,
        one launch per level -- while the small levels keep the one-launch-for-all-levels path.  Returns the tower's outputs
        (normalised, activated) per level, or None when nothing folds."""
        import ops_f16
        if not (ops_f16.FOLD and F16_HEAD_STATS):
            return None
        blocks = [blk.layers for blk in self.pre_conv.layers]          # [conv, norm, act] each
        big = [i for i, m in enumerate(maps) if m.dtype == torch.float16 and m.shape[0] * m.shape[1] * m.shape[2] >= F16_HEAD_STATS_MIN_ROWS]
        if not big:
            return None
        small = [i for i in range(len(maps)) if i not in big]
        cur = list(maps)
        for conv, norm, act in blocks:
            nxt = [None] * len(cur)
            for i in big:
                p = ops_f16.conv2d_norm(cur[i], conv.weight, norm, L.activation_name(act), 1, 1)
                if p is None:
                    return None
                nxt[i] = p.materialise()
            if small:
                ys = norm.fused(conv([cur[i] for i in small]), False, act=L.activation_name(act))
                for i, y in zip(small, ys):
                    nxt[i] = y
            cur = nxt
        return cur

    def call(self, input, training):
        """`input`: one feature map, or a list of maps (all pyramid levels, one launch per layer)."""
        multi = isinstance(input, (list, tuple))
        out = self._folded(list(input) if multi else [input], training)
        if out is None and L.INFERENCE_F16 and not training and multi and all(torch.is_tensor(m) and m.is_cuda for m in input):
            tower = self._f16_tower_levels(list(input))
            if tower is None:
                tower = self._f16_tower(list(input))
            if tower is not None:
                out = list(self.out_conv(tower))
        if out is None:
            out = self.out_conv(self.pre_conv(input, training))
            out = list(out) if multi else [out]
        out = [self._reshape(t) for t in out]
        return out if multi else out[0]


class ClassificationSubnet(_Subnet):
    def __init__(self, num_anchors, num_classes, activation, kernel_initializer, kernel_regularizer,
                 name='classification_subnet'):
        pi = 0.01                                     # prior: sigmoid(bias) = pi (retinanet.py:52-53)
        super().__init__(num_anchors, num_classes, activation, kernel_initializer, kernel_regularizer,
                         L.Constant(-math.log((1 - pi) / pi)), name)
        self.num_classes = num_classes


class RegressionSubnet(_Subnet):
    def __init__(self, num_anchors, activation, kernel_initializer, kernel_regularizer,
                 name='classification_subnet'):
        super().__init__(num_anchors, 4, activation, kernel_initializer, kernel_regularizer, None, name)


class FeaturePyramidNetwork(Model):
    class UpsampleMerge(Model):
        def __init__(self, kernel_initializer, kernel_regularizer, name='upsample_merge', in_channels=None):
            super().__init__(name=name)
            self.conv_lateral = _conv_norm(256, 1, 1, kernel_initializer, kernel_regularizer, in_channels=in_channels)
            self.conv_merge = _conv_norm(256, 3, 1, kernel_initializer, kernel_regularizer, in_channels=256)

        def call(self, lateral, downsampled, training):
            lateral = self.conv_lateral(lateral, training)
            merged = ops.upsample_add(lateral, downsampled)
            return self.conv_merge(merged, training)

    def __init__(self, activation, kernel_initializer, kernel_regularizer, name='feature_pyramid_network',
                 feature_channels=None):
        """`feature_channels` (optional {'C3','C4','C5'} -> channels) builds the lateral kernels
        eagerly; without it they are created on the first call, like tf.layers."""
        super().__init__(name=name)
        act = L.get_activation(activation)
        fc = feature_channels or {}
        self.p6_from_c5 = _conv_norm(256, 3, 2, kernel_initializer, kernel_regularizer, in_channels=fc.get('C5'))
        self.p7_from_p6 = _conv_norm(256, 3, 2, kernel_initializer, kernel_regularizer, in_channels=256, pre=act)
        self.p5_from_c5 = _conv_norm(256, 1, 1, kernel_initializer, kernel_regularizer, in_channels=fc.get('C5'))
        self.p4_from_c4p5 = FeaturePyramidNetwork.UpsampleMerge(kernel_initializer, kernel_regularizer,
                                                                name='upsample_merge_c4p5', in_channels=fc.get('C4'))
        self.p3_from_c3p4 = FeaturePyramidNetwork.UpsampleMerge(kernel_initializer, kernel_regularizer,
                                                                name='upsample_merge_c3p4', in_channels=fc.get('C3'))

    def call(self, input, training):
        # a tensor with two consumers goes through ops.fanout: its two gradients are summed by one launch of our add kernel
        c5a, c5b = ops.fanout(input['C5'], 2)
        two = FPN_TWO_STREAMS and c5a.is_cuda
        if two:     # the P6 -> P7 chain is independent of the P5 -> P4 -> P3 chain: a second stream (forward and, through autograd, backward)
            main, side = torch.cuda.current_stream(), side_stream(c5a.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                P6, p6 = ops.fanout(self.p6_from_c5(c5a, training), 2)
                P7 = self.p7_from_p6(p6, training)
        else:
            P6, p6 = ops.fanout(self.p6_from_c5(c5a, training), 2)
            P7 = self.p7_from_p6(p6, training)
        P5, p5 = ops.fanout(self.p5_from_c5(c5b, training), 2)
        P4, p4 = ops.fanout(self.p4_from_c4p5(input['C4'], p5, training), 2)
        P3 = self.p3_from_c3p4(input['C3'], p4, training)
        if two:
            main.wait_stream(side)
        return {'P3': P3, 'P4': P4, 'P5': P5, 'P6': P6, 'P7': P7}       # order matters


class RetinaNetBase(Model):
    def __init__(self, backbone, levels, num_classes, activation, dropout_rate, kernel_initializer,
                 kernel_regularizer, name='retinanet_base'):
        super().__init__(name=name)
        self.backbone = build_backbone(backbone, activation=activation, dropout_rate=dropout_rate)
        # the reference's `if backbone == 'densenet'` post-norm never fires for any accepted
        # backbone name (retinanet.py:238-250 vs :13), so it is not built.
        self.fpn = FeaturePyramidNetwork(activation=activation, kernel_initializer=kernel_initializer,
                                         kernel_regularizer=kernel_regularizer,
                                         feature_channels=getattr(self.backbone, 'out_channels', None))
        self.classification_subnet = ClassificationSubnet(
            num_anchors=levels.num_anchors, num_classes=num_classes, activation=activation,
            kernel_initializer=kernel_initializer, kernel_regularizer=kernel_regularizer,
            name='classification_subnet')
        self.regression_subnet = RegressionSubnet(
            num_anchors=levels.num_anchors, activation=activation, kernel_initializer=kernel_initializer,
            kernel_regularizer=kernel_regularizer, name='regression_subnet')
        # train.Trainer installs a callable here that replaces the taps the FPN reads by detached leaves, so the
        # backward pass runs in two segments (heads + FPN first, their gradient all-reduce under the backbone's)
        self.backward_cut = None

    def call(self, input, training):
        bottom_up = self.backbone(input, training)
        if self.backward_cut is not None and training and torch.is_grad_enabled():
            bottom_up = self.backward_cut(bottom_up)
        top_down = self.fpn(bottom_up, training)
        keys = list(top_down.keys())
        maps = [top_down[k] for k in keys]
        if FUSE_HEAD_TOWERS and maps[0].is_cuda:
            cls_out, reg_out = self._fused_heads(maps, training)
        else:
            maps_c, maps_r = ops.fanout(maps, 2) if maps[0].is_cuda else (maps, maps)   # both subnets read every level
            if HEADS_TWO_STREAMS and maps[0].is_cuda:
                main, side = torch.cuda.current_stream(), side_stream(maps[0].device)
                side.wait_stream(main)
                if HEADS_BOX_FIRST:                    # tuning aid: the box subnet's launches recorded before the class subnet's
                    with torch.cuda.stream(side):
                        reg_out = self.regression_subnet(maps_r, training)
                    cls_out = self.classification_subnet(maps_c, training)
                else:
                    cls_out = self.classification_subnet(maps_c, training)
                    with torch.cuda.stream(side):
                        if HEADS_OFFSET_US > 0:        # tuning aid: start the box subnet a fraction of a layer later
                            _delay(maps[0].device, side, HEADS_OFFSET_US)
                        reg_out = self.regression_subnet(maps_r, training)
                main.wait_stream(side)
            else:
                cls_out = self.classification_subnet(maps_c, training)
                reg_out = self.regression_subnet(maps_r, training)
        classifications = dict(zip(keys, cls_out))
        regressions = dict(zip(keys, reg_out))
        return {'classifications': classifications, 'regressions': regressions}


    def _fused_heads(self, maps, training):
        """Both subnets as one 512-channel tower (see FUSE_HEAD_TOWERS); parameters stay the subnets' own
        tensors and are concatenated on the fly (a few MB of copies per step)."""
        cs, rs = self.classification_subnet, self.regression_subnet
        x = maps
        for i in range(4):
            bc, br = cs.pre_conv.layers[i].layers, rs.pre_conv.layers[i].layers      # [conv, norm, act]
            w = torch.cat([bc[0].weight, br[0].weight], 3)
            x = ops.conv2d(x, w, None, 1, groups=1 if i == 0 else 2)
            gamma = torch.cat([bc[1].gamma, br[1].gamma])
            beta = torch.cat([bc[1].beta, br[1].beta])
            x = ops.group_norm_act(x, gamma, beta, groups=2 * ops.gn_groups(256, bc[1].groups), eps=bc[1].eps,
                                   act=L.activation_name(bc[2]))
        outs = ops.conv2d_channel_split(x, [cs.out_conv.weight, rs.out_conv.weight],
                                        [cs.out_conv.bias, rs.out_conv.bias], 1)
        return [cs._reshape(t) for t in outs[0]], [rs._reshape(t) for t in outs[1]]


class RetinaNet(Model):
    def __init__(self, backbone, levels, num_classes, activation, dropout_rate, name='retinanet'):
        super().__init__(name=name)
        self.base = RetinaNetBase(
            backbone=backbone, levels=levels, num_classes=num_classes, activation=activation,
            dropout_rate=dropout_rate,
            kernel_initializer=L.RandomNormal(mean=0.0, stddev=0.01),
            kernel_regularizer=L.L2Regularizer(scale=1e-4))

    def call(self, input, training):
        return self.base(input, training)
