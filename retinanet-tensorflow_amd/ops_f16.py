"""fp16 inference ops (forward only, no autograd): fp16-storage activations, f16 matrix-core convs with fp32
accumulation, GroupNorm statistics in fp32.  BASELINE configs[4] (ResNeXt-50-FPN, 1024x1024, batch 16, fp16).
Kernels: csrc/conv_f16.hip and the fp16-storage variants in group_norm.hip / elementwise.hip."""
import ctypes as C

import torch

import _rn
from ops import gn_groups

_packed = {}    # (data_ptr, version, shape) -> packed fp16 kernel Wt[cout][K]


def packed_weight(w):
    """fp32 HWIO [kh,kw,cin_g,cout] -> fp16 [cout, kh*kw*cin_g] (k contiguous); cached per weight version."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    wt = _packed.get(key)
    if wt is None:
        kh, kw, cin_g, cout = w.shape
        wt = torch.empty((cout, kh * kw * cin_g), dtype=torch.float16, device=w.device)
        _rn.check(_rn.lib().rn_pack_weights_f16(_rn.f32(w.detach().contiguous()), _rn.f16(wt), kh, kw, cin_g, cout,
                                                _rn.stream()), "rn_pack_weights_f16")
        _packed[key] = wt
    return wt


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
    cin = cin_g * groups
    wt = packed_weight(w)
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
        s.bias = _rn.f32(bias) if bias is not None else None
        s.n, s.h, s.w, s.cout = t.shape[0], t.shape[1], t.shape[2], cout
    _rn.check(L.rn_conv2d_fwd_f16(segs, len(xs), C.byref(geom), 1 if out_f32 else 0, _rn.stream()), "rn_conv2d_fwd_f16")
    return ys if multi else ys[0]


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
