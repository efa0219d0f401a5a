"""fp16 inference ops (forward only, no autograd): fp16-storage activations, f16 matrix-core convs with fp32
accumulation, GroupNorm statistics in fp32.  BASELINE configs[4] (ResNeXt-50-FPN, 1024x1024, batch 16, fp16).
Kernels: csrc/conv_f16.hip and the fp16-storage variants in group_norm.hip / elementwise.hip."""
import ctypes as C

import torch

import _rn
from ops import gn_groups

# Packed fp16 kernels are cached ON the weight tensor object (attribute `_rn_f16_cache`), keyed by the tensor's version and
# WEIGHTS_EPOCH: the optimizer kernel and checkpoint.load write weights through raw pointers (no version bump), so they
# advance the epoch.  (A cache keyed by data_ptr served another model's kernel once the allocator reused an address.)
WEIGHTS_EPOCH = 0


def weights_changed():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


SUPER_GROUP = 32    # narrow groups are merged into block-diagonal groups of this many input channels


def _pack(w):
    kh, kw, cin_g, cout = w.shape
    # Wt[cout, K] and, behind it, the fragment-ordered copy the large-tile kernel reads (rn_hip.h: rn_pack_weights_f16_bytes)
    wt = torch.empty((int(_rn.lib().rn_pack_weights_f16_bytes(kh, kw, cin_g, cout)) // 2,), dtype=torch.float16, device=w.device)
    _rn.check(_rn.lib().rn_pack_weights_f16(_rn.f32(w.contiguous()), _rn.f16(wt), kh, kw, cin_g, cout, _rn.stream()),
              "rn_pack_weights_f16")
    return wt


def packed_weight(w, groups=1, pad_cin_to=None):
    """fp32 HWIO [kh,kw,cin_g,cout] -> (fp16 Wt[cout, K], groups', cin').  Cached per weight version.

    * groups with fewer than 32 channels (ResNeXt stages 2-4: 4 / 8 / 16 per group) are merged into
      block-diagonal super-groups of 32 input channels (zeros off the diagonal): 8x / 4x / 2x fewer, full-width
      tiles and 16-byte gathers instead of one mostly-empty tile per group -- on the f16 matrix cores the extra
      multiplies by zero are free compared with the tile overhead they remove;
    * pad_cin_to: zero-pad the input-channel axis (the RGB stem reads an image padded to 4 channels).
    """
    key = (WEIGHTS_EPOCH, w._version, w.data_ptr(), groups, pad_cin_to)
    owner = w
    cache = getattr(owner, '_rn_f16_cache', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    w = w.detach()
    kh, kw, cin_g, cout = w.shape
    g2 = groups
    if pad_cin_to is not None and pad_cin_to > cin_g:
        assert groups == 1
        w = torch.cat([w, torch.zeros((kh, kw, pad_cin_to - cin_g, cout), dtype=w.dtype, device=w.device)], 2)
        cin_g = pad_cin_to
    elif groups > 1 and cin_g < SUPER_GROUP and SUPER_GROUP % cin_g == 0 and (cin_g * groups) % SUPER_GROUP == 0:
        per = SUPER_GROUP // cin_g                      # original groups per super-group
        cout_g = cout // groups
        wide = torch.zeros((kh, kw, SUPER_GROUP, cout), dtype=w.dtype, device=w.device)
        for j in range(per):                            # output channels of the j-th member of every super-group
            cols = torch.arange(cout, device=w.device).view(groups // per, per, cout_g)[:, j, :].reshape(-1)
            wide[:, :, j * cin_g:(j + 1) * cin_g, cols] = w[:, :, :, cols]
        w, cin_g, g2 = wide, SUPER_GROUP, groups // per
    hit = (_pack(w), g2, cin_g * g2)
    owner._rn_f16_cache = (key, hit)
    return hit


def image_to_half4(x):
    """[N,H,W,3] fp32 image -> [N,H,W,4] fp16 with a zero 4th channel (8-byte gathers in the stem conv)."""
    x = x.contiguous()
    assert x.shape[3] == 3 and x.dtype == torch.float32
    y = torch.empty(x.shape[:3] + (4,), dtype=torch.float16, device=x.device)
    _rn.check(_rn.lib().rn_pad_cast_rgb_f16(_rn.f32(x), _rn.f16(y), x.numel() // 3, _rn.stream()), "rn_pad_cast_rgb_f16")
    return y


def to_half(x):
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _rn.check(_rn.lib().rn_cast_f32_to_f16(_rn.f32(x), _rn.f16(y), x.numel(), _rn.stream()), "rn_cast_f32_to_f16")
    return y


def conv2d(x, w, bias=None, stride=1, groups=1, out_f32=False):
    """x: fp16 NHWC tensor or list of them (shared kernel, one launch); w: the fp32 parameter (packed on first use)."""
    multi = isinstance(x, (list, tuple))
    xs = [t.contiguous() for t in (x if multi else [x])]
    kh, kw, cin_g, cout = w.shape
    pad = xs[0].shape[3] if (groups == 1 and xs[0].shape[3] > cin_g) else None     # RGB image padded to 4 channels
    wt, groups, cin = packed_weight(w, groups, pad)
    L = _rn.lib()
    geom = _rn.ConvGeom(kh, kw, stride, cin, groups)
    segs = (_rn.ConvSeg * len(xs))()
    ys = []
    for i, t in enumerate(xs):
        assert t.dtype == torch.float16 and t.shape[3] == cin
        oh, _ = _rn.same_pad(t.shape[1], kh, stride)
        ow, _ = _rn.same_pad(t.shape[2], kw, stride)
        y = torch.empty((t.shape[0], oh, ow, cout), dtype=torch.float32 if out_f32 else torch.float16, device=t.device)
        ys.append(y)
        s = segs[i]
        s.x, s.wgt, s.y = _rn.f16(t), _rn.f16(wt), _rn.ptr(y)
        s.wgt_bytes = wt.numel() * 2          # proves that the fragment-ordered copy sits behind Wt (rn_conv_seg.wgt_bytes)
        s.bias = _rn.f32(bias) if bias is not None else None
        s.n, s.h, s.w, s.cout = t.shape[0], t.shape[1], t.shape[2], cout
    _rn.check(L.rn_conv2d_fwd_f16(segs, len(xs), C.byref(geom), 1 if out_f32 else 0, _rn.stream()), "rn_conv2d_fwd_f16")
    return ys if multi else ys[0]


# ---------------------------------------------------------------------------------------------------------------------
# GroupNorms folded into the convs around them (rn_conv2d_fwd_f16_fold): a conv's epilogue emits the statistics of its
# output, the NEXT conv applies GroupNorm + activation to its operand on load.  A `Pending` is a conv output whose
# GroupNorm has not been applied yet: another folded conv consumes it as is, everything else calls materialise().
# ---------------------------------------------------------------------------------------------------------------------
import os
FOLD_INTO_3X3 = os.environ.get("RN_F16_FOLD_3X3", "0") == "1"     # also apply a pending GroupNorm on the operand load of 3 x 3 convs
SG_KERNEL = os.environ.get("RN_F16_SG", "1") != "0"


def sg_kernel_takes(shape, w, stride, groups):
    """True where rn_conv2d_fwd_f16(_fold) runs a grouped 3 x 3 conv on 16 x 16-pixel tiles of 32-channel super-groups
    (csrc/conv_f16.hip, conv3x3_sg32_f16_kernel): it applies a pending GroupNorm + activation ONCE per input element (while the
    patch goes to LDS), so the conv in front need not materialise its GroupNorm (resnet.ResNeXt_Bottleneck)."""
    kh, kw, cin_g, cout = w.shape
    return (SG_KERNEL and groups > 1 and stride in (1, 2) and kh == 3 and kw == 3 and cout == cin_g * groups and cin_g in (4, 8, 16, 32)
            and shape[3] == cout and shape[1] % (16 * stride) == 0 and shape[2] % (16 * stride) == 0)


FOLD = os.environ.get("RN_F16_FOLD", "1") == "1"     # (0: conv, then the three-kernel GroupNorm, as before -- A/B measurements, tests)


class Pending(object):
    """y = raw fp16 conv output [N,H,W,C]; (mean, rstd) [N,G]; the GroupNorm's gamma / beta / groups and the activation behind it."""

    def __init__(self, y, mean, rstd, gamma, beta, groups, act):
        self.y, self.mean, self.rstd, self.gamma, self.beta, self.groups, self.act = y, mean, rstd, gamma, beta, groups, act

    def materialise(self, residual=None, act_after_residual=False):
        y = self.y
        n, h, w, c = y.shape
        out = torch.empty_like(y)
        if isinstance(residual, Pending):
            # the residual's own (activation-free) GroupNorm is applied inside this pass: no apply pass of its own
            assert residual.act is None and residual.y.shape == y.shape
            rn_ = _rn.GnResidualNorm(residual.mean.data_ptr(), residual.rstd.data_ptr(), residual.gamma.data_ptr(), residual.beta.data_ptr(),
                                     residual.groups)
            _rn.check(_rn.lib().rn_group_norm_apply_res_f16(_rn.f16(y), _rn.f16(residual.y), C.byref(rn_), _rn.f16(out), n, h * w, c, self.groups,
                                                            _rn.f32(self.mean), _rn.f32(self.rstd), _rn.f32(self.gamma), _rn.f32(self.beta),
                                                            _rn.ACT[self.act], 1 if act_after_residual else 0, _rn.stream()),
                      "rn_group_norm_apply_res_f16")
            return out
        residual = residual.contiguous() if residual is not None else None   # (a copy must outlive the launch: keep the name)
        _rn.check(_rn.lib().rn_group_norm_apply_f16(_rn.f16(y), _rn.f16(residual) if residual is not None else None, _rn.f16(out),
                                                    n, h * w, c, self.groups, _rn.f32(self.mean), _rn.f32(self.rstd), _rn.f32(self.gamma),
                                                    _rn.f32(self.beta), _rn.ACT[self.act], 1 if act_after_residual else 0, _rn.stream()),
                  "rn_group_norm_apply_f16")
        return out


def max_pool_norm(p, k=3, stride=2):
    """max_pool(act(GroupNorm(y))) of a Pending conv output in one pass over y (rn_maxpool_gn_fwd_f16): the normalised tensor is
    never written.  Bit-equal to max_pool(p.materialise())."""
    y = p.y
    n, h, w, c = y.shape
    oh, _ = _rn.same_pad(h, k, stride)
    ow, _ = _rn.same_pad(w, k, stride)
    out = torch.empty((n, oh, ow, c), dtype=torch.float16, device=y.device)
    _rn.check(_rn.lib().rn_maxpool_gn_fwd_f16(_rn.f16(y), _rn.f16(out), n, h, w, c, k, stride, _rn.f32(p.mean), _rn.f32(p.rstd),
                                              _rn.f32(p.gamma), _rn.f32(p.beta), p.groups, _rn.ACT[p.act], _rn.stream()),
              "rn_maxpool_gn_fwd_f16")
    return out


def conv2d_norm(x, w, norm, act=None, stride=1, groups=1):
    """conv (no bias) -> GroupNorm `norm` (a layers.GroupNormalization) -> act, the GroupNorm NOT applied: returns a Pending, or
    None when the shape cannot fold (the caller then takes conv2d + group_norm_act).  `x`: an fp16 tensor or a Pending (its
    GroupNorm + activation are applied by this conv's operand load)."""
    src = x.y if isinstance(x, Pending) else x.contiguous()
    kh, kw, cin_g, cout = w.shape
    # the RGB image padded to 4 fp16 channels against a 3-input kernel -- and nothing else: any other channel mismatch is a
    # mis-wired layer and must not be papered over with zero weights (the check below then returns None and the caller raises)
    pad = 4 if (groups == 1 and src.shape[3] == 4 and cin_g == 3 and not isinstance(x, Pending)) else None
    wt, g2, cin = packed_weight(w, groups, pad)
    if src.shape[3] != cin:
        return None
    L = _rn.lib()
    geom = _rn.ConvGeom(kh, kw, stride, cin, g2)
    oh, _ = _rn.same_pad(src.shape[1], kh, stride)
    ow, _ = _rn.same_pad(src.shape[2], kw, stride)
    n = src.shape[0]
    y = torch.empty((n, oh, ow, cout), dtype=torch.float16, device=src.device)
    seg = (_rn.ConvSeg * 1)()
    s = seg[0]
    s.x, s.wgt, s.y, s.bias = _rn.f16(src), _rn.f16(wt), _rn.f16(y), None
    s.wgt_bytes = wt.numel() * 2
    s.n, s.h, s.w, s.cout = n, src.shape[1], src.shape[2], cout
    rows = L.rn_conv2d_f16_fold_rows(seg, 1, C.byref(geom))
    if rows <= 0:
        return None
    if norm.gamma is None:
        norm.build(cout, src.device)
    gout = gn_groups(cout, norm.groups)
    partial = torch.empty((2, n * rows, cout), dtype=torch.float32, device=src.device)
    fold = _rn.F16Fold()
    fold.partial = partial.data_ptr()
    if isinstance(x, Pending):
        fold.in_mean, fold.in_rstd = x.mean.data_ptr(), x.rstd.data_ptr()
        fold.in_gamma, fold.in_beta = x.gamma.data_ptr(), x.beta.data_ptr()
        fold.in_groups, fold.in_act = x.groups, _rn.ACT[x.act]
    _rn.check(L.rn_conv2d_fwd_f16_fold(seg, 1, C.byref(geom), C.byref(fold), _rn.stream()), "rn_conv2d_fwd_f16_fold")
    mean = torch.empty((n, gout), dtype=torch.float32, device=src.device)
    rstd = torch.empty((n, gout), dtype=torch.float32, device=src.device)
    _rn.check(L.rn_group_norm_finalize(_rn.f32(partial), n, rows, oh * ow, cout, gout, float(norm.eps), _rn.f32(mean), _rn.f32(rstd),
                                       _rn.stream()), "rn_group_norm_finalize")
    return Pending(y, mean, rstd, norm.gamma.detach(), norm.beta.detach(), gout, act)


HEAD_EPILOGUE_STATS = os.environ.get("RN_F16_HEAD_EPILOGUE_STATS", "1") == "1"


def conv_norm_act_levels(xs, w, norm, act=None):
    """act(GroupNorm(conv3x3(x))) for every tensor of the list `xs` (shared kernel, no bias: the head towers' blocks on all pyramid
    levels, retinanet.py:37-62,85-106) in THREE launches: the conv of all levels with the GroupNorm statistics of its output from its
    epilogue (rn_conv2d_fwd_f16_fold, seg_chunk_start), finalise, apply (rn_group_norm_fwd_f16_tiles) -- the statistics pass over the
    conv outputs (179 MB per GroupNorm at cfg 5) is not run.  A level whose conv tiles straddle samples (P7) is summed from its tensor by
    the finalise blocks.  None: the shape does not qualify (the caller runs conv2d + group_norm_act)."""
    if not (FOLD and HEAD_EPILOGUE_STATS):
        return None
    xs = [t.contiguous() for t in xs]
    kh, kw, cin_g, cout = w.shape
    if any(t.dtype != torch.float16 or t.shape[3] != cin_g for t in xs) or cout % 64 != 0:
        return None
    wt, g2, cin = packed_weight(w, 1, None)
    L = _rn.lib()
    n = len(xs)
    dev = xs[0].device
    geom = _rn.ConvGeom(kh, kw, 1, cin, g2)
    segs = (_rn.ConvSeg * n)()
    raws = []
    for i, t in enumerate(xs):
        y = torch.empty((t.shape[0], t.shape[1], t.shape[2], cout), dtype=torch.float16, device=dev)
        raws.append(y)
        s = segs[i]
        s.x, s.wgt, s.y, s.bias = _rn.f16(t), _rn.f16(wt), _rn.f16(y), None
        s.wgt_bytes = wt.numel() * 2
        s.n, s.h, s.w, s.cout = t.shape[0], t.shape[1], t.shape[2], cout
    tiles = (C.c_int32 * n)()
    if L.rn_conv2d_f16_stats_tiles(segs, n, C.byref(geom), tiles) != 0 or not any(tiles[i] for i in range(n)):
        return None
    if any(tiles[i] == 0 and xs[i].shape[1] * xs[i].shape[2] > 4096 for i in range(n)):
        return None
    starts = (C.c_int32 * n)()
    total = 0
    for i in range(n):
        starts[i] = total
        total += xs[i].shape[0] * tiles[i]
    partial = torch.empty((2, total, cout), dtype=torch.float32, device=dev)
    fold = _rn.F16Fold()
    fold.partial = partial.data_ptr()
    fold.seg_chunk_start = C.cast(starts, C.c_void_p)
    fold.total_chunks = total
    _rn.check(L.rn_conv2d_fwd_f16_fold(segs, n, C.byref(geom), C.byref(fold), _rn.stream()), "rn_conv2d_fwd_f16_fold")
    if norm.gamma is None:
        norm.build(cout, dev)
    g = gn_groups(cout, norm.groups)
    gsegs = (_rn.GnSeg * n)()
    ys, keep = [], []
    for i, t in enumerate(raws):
        y = torch.empty_like(t)
        mean = torch.empty((t.shape[0], g), dtype=torch.float32, device=dev)
        rstd = torch.empty((t.shape[0], g), dtype=torch.float32, device=dev)
        keep += [mean, rstd]
        ys.append(y)
        s = gsegs[i]
        s.x, s.y, s.residual = _rn.ptr(t), _rn.f16(y), None
        s.mean, s.rstd = _rn.f32(mean), _rn.f32(rstd)
        s.n, s.hw = t.shape[0], t.shape[1] * t.shape[2]
    params = _rn.GnParams(c=cout, groups=g, act=_rn.ACT[act], act_after_residual=0, in_f16=1, out_f16=1, eps=float(norm.eps), drop_rate=0.0,
                          drop_seed=0, drop_seed_dev=None)
    _rn.check(L.rn_group_norm_fwd_f16_tiles(gsegs, n, C.byref(params), _rn.f32(norm.gamma.detach()), _rn.f32(norm.beta.detach()), partial.data_ptr(),
                                            C.cast(tiles, C.c_void_p), _rn.stream()), "rn_group_norm_fwd_f16_tiles")
    return ys


def group_norm_act(x, gamma, beta, groups=32, eps=1e-5, act=None, residual=None, act_after_residual=False):
    """GroupNorm -> act (-> + residual) with fp32 or fp16 input and fp16 output/residual.  Lists allowed."""
    multi = isinstance(x, (list, tuple))
    xs = [t.contiguous() for t in (x if multi else [x])]
    ress = list(residual) if isinstance(residual, (list, tuple)) else [residual] * len(xs)
    c = xs[0].shape[3]
    g = gn_groups(c, groups)
    dev = xs[0].device
    in_half = xs[0].dtype == torch.float16
    L = _rn.lib()
    segs = (_rn.GnSeg * len(xs))()
    ys, keep = [], []
    for i, t in enumerate(xs):
        assert (t.dtype == torch.float16) == in_half
        y = torch.empty(t.shape, dtype=torch.float16, device=dev)
        mean = torch.empty((t.shape[0], g), dtype=torch.float32, device=dev)
        rstd = torch.empty((t.shape[0], g), dtype=torch.float32, device=dev)
        keep += [mean, rstd]
        ys.append(y)
        s = segs[i]
        s.x, s.y = _rn.ptr(t), _rn.f16(y)
        s.residual = _rn.f16(ress[i].contiguous()) if ress[i] is not None else None
        s.mean, s.rstd = _rn.f32(mean), _rn.f32(rstd)
        s.n, s.hw = t.shape[0], t.shape[1] * t.shape[2]
    params = _rn.GnParams(c=c, groups=g, act=_rn.ACT[act], act_after_residual=1 if act_after_residual else 0,
                          in_f16=1 if in_half else 0, out_f16=1, eps=float(eps), drop_rate=0.0, drop_seed=0,
                          drop_seed_dev=None)
    need = L.rn_group_norm_workspace(segs, len(xs), C.byref(params))
    ws = _rn.workspace(need, dev)
    _rn.check(L.rn_group_norm_fwd(segs, len(xs), C.byref(params), _rn.f32(gamma), _rn.f32(beta), ws.data_ptr(),
                                  ws.numel(), _rn.stream()), "rn_group_norm_fwd")
    return ys if multi else ys[0]


def max_pool(x, k=3, stride=2):
    x = x.contiguous()
    n, h, w, c = x.shape
    oh, _ = _rn.same_pad(h, k, stride)
    ow, _ = _rn.same_pad(w, k, stride)
    y = torch.empty((n, oh, ow, c), dtype=torch.float16, device=x.device)
    _rn.check(_rn.lib().rn_maxpool_fwd_f16(_rn.f16(x), _rn.f16(y), n, h, w, c, k, stride, _rn.stream()), "rn_maxpool_fwd_f16")
    return y


def upsample_add(lateral, top):
    lateral, top = lateral.contiguous(), top.contiguous()
    n, h, w, c = lateral.shape
    y = torch.empty_like(lateral)
    _rn.check(_rn.lib().rn_upsample_add_fwd_f16(_rn.f16(lateral), _rn.f16(top), _rn.f16(y), n, h, w, top.shape[1],
                                                top.shape[2], c, _rn.stream()), "rn_upsample_add_fwd_f16")
    return y


def activation(x, act):
    if _rn.ACT[act] == 0:
        return x
    x = x.contiguous()
    y = torch.empty_like(x)
    _rn.check(_rn.lib().rn_act_fwd_f16(_rn.f16(x), _rn.f16(y), x.numel(), _rn.ACT[act], _rn.stream()), "rn_act_fwd_f16")
    return y
