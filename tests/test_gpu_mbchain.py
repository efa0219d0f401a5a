"""The MobileNetV2 bottleneck chain (rn_mb_* kernels, ops_mb.mb_chain: every GroupNorm applied by its consumer) against the
oracle and against the layer-by-layer product path.  Reference: mobilenet_v2.py:41-94, normalization.py:20-35."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_close, load_oracle_params, to_oracle_name
from oracle import model_ref, tf_ops_ref as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("n,hw,cin,cout,groups", [(2, 256, 64, 96, 32), (1, 4096, 24, 144, 24), (2, 1024, 384, 64, 32), (2, 16384, 16, 24, 24)])
def test_pointwise_rows_and_apply(dev, n, hw, cin, cout, groups):
    """rn_mb_pointwise_fwd (plain A) writes y = x w and y's per-group rows; rn_mb_apply merges them: out = GN(y) as the oracle's
    group_norm gives it (incl. the case where the conv's N-tiles cut through groups: 96 / 32 = 3 channels per group, 64-wide tiles)."""
    import _rn
    rng = np.random.default_rng(hw + cin)
    x = rng.standard_normal((n, hw, 1, cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(cout)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    L = _rn.lib()
    lay = _rn.MbRows()
    nbytes = L.rn_mb_pointwise_rows(n, hw, cin, cout, groups, C.byref(lay))
    assert nbytes
    rows = torch.empty(nbytes // 4, device=dev)
    y = torch.empty((n, hw, 1, cout), device=dev)
    xd, wd = _t(x, dev), _t(w, dev)
    lay.rows = rows.data_ptr()
    _rn.check(L.rn_mb_pointwise_fwd(_rn.f32(xd), None, None, None, _rn.f32(wd), _rn.f32(y), n, hw, cin, cout, C.byref(lay), groups, _rn.stream()), "fwd")
    yref = T.conv2d_same(torch.from_numpy(x), torch.from_numpy(w), 1)
    assert_close(y.cpu().numpy(), yref.numpy(), 1e-5, "pointwise y")
    mean, rstd = torch.empty((n, groups), device=dev), torch.empty((n, groups), device=dev)
    g, b = _t(gamma, dev), _t(beta, dev)
    nm = _rn.MbNorm()
    nm.y, nm.stat, nm.mean, nm.rstd = y.data_ptr(), lay, mean.data_ptr(), rstd.data_ptr()
    nm.gamma, nm.beta, nm.c, nm.groups, nm.act, nm.eps = g.data_ptr(), b.data_ptr(), cout, groups, _rn.ACT["elu"], 1e-5
    out = torch.empty_like(y)
    _rn.check(L.rn_mb_apply(C.byref(nm), None, _rn.f32(out), n, hw, _rn.stream()), "apply")
    ref = T.activation(T.group_norm(yref, torch.from_numpy(gamma), torch.from_numpy(beta), groups), "elu")
    assert_close(out.cpu().numpy(), ref.numpy(), 1e-5, "GN(y) from the rows")
    yg = yref.reshape(n, hw, groups, cout // groups)
    assert_close(mean.cpu().numpy(), yg.mean((1, 3)).numpy(), 1e-5, "mean")


def _backbone(dev, rate, seed=0):
    import layers, levels, retinanet
    params = model_ref.init_params("mobilenet_v2", num_classes=3, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in params:
        if k.endswith(".gamma"):
            params[k] = 1 + 0.2 * torch.randn(params[k].shape, generator=g)
        elif k.endswith(".beta"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    net = retinanet.RetinaNet('mobilenet_v2', levels.build_levels(), 3, layers.elu, rate).to(dev)
    load_oracle_params(net, params)
    return net.base.backbone, params


def _run(bb, x, cot, training=True):
    for p in bb.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    out = bb(xin, training=training)
    loss = sum((out[k] * cot[k]).sum() for k in ("C3", "C4", "C5"))
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in bb.named_parameters()}
    return {k: out[k].detach() for k in ("C3", "C4", "C5")}, grads, xin.grad.detach()


@pytest.mark.parametrize("size,batch,big", [(256, 2, False), (512, 1, False), (512, 1, True)])
def test_chain_matches_oracle(dev, size, batch, big, monkeypatch):
    """Backbone forward (C3 / C4 / C5) and the gradients of every backbone parameter and of the image through the fused chain
    vs the oracle (dropout 0): outputs 1e-4, gradients 5e-4 of max(|g|, 1e-3 x the largest gradient).  big: the chain also
    takes the 256 x 256 maps (their statistic rows go through rn_mb_compact_rows)."""
    import mobilenet_v2
    monkeypatch.setattr(mobilenet_v2, "MB_CHAIN_MAX_HW", 0 if big else 16384)   # (0 = the default: the whole backbone)
    bb, params = _backbone(dev, 0.0)
    rng = np.random.default_rng(size)
    x = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
    stem = bb.input_conv(x.to(dev), training=True)
    start = bb._chain_start(stem, True)
    assert start is not None and start <= (0 if big else 2), "the fused chain must run here (starts at bottleneck %s)" % start
    shapes = {"C3": (batch, size // 8, size // 8, 32), "C4": (batch, size // 16, size // 16, 96), "C5": (batch, size // 32, size // 32, 32)}
    cot = {k: torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for k, s in shapes.items()}
    out, grads, dx = _run(bb, x.to(dev), {k: v.to(dev) for k, v in cot.items()})
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items() if k.startswith("backbone")}
    xr = x.clone().requires_grad_(True)
    ref = model_ref.mobilenet_v2_forward(leaves, xr)
    for k in shapes:
        assert_close(out[k].cpu().numpy(), ref[k].detach().numpy(), 1e-4, "chain " + k)
    loss = sum((ref[k] * cot[k]).sum() for k in shapes)
    names = list(leaves.keys())
    gref = dict(zip(names + ["x"], torch.autograd.grad(loss, [leaves[n] for n in names] + [xr])))
    scale = max(float(gref[n].abs().max()) for n in names)
    errs = []
    for name, g in grads.items():
        o = to_oracle_name("base.backbone." + name)
        r = gref[o].numpy()
        errs.append((float(np.abs(g.cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-3 * scale), name))
    errs.sort(reverse=True)
    bad = [e for e in errs if e[0] > 5e-4]
    assert not bad, "%d of %d gradients off: %s" % (len(bad), len(errs), ", ".join("%s %.2e" % (n, e) for e, n in bad[:12]))
    assert_close(dx.cpu().numpy(), gref["x"].numpy(), 5e-4, "image gradient")
    print("chain vs oracle at %d px: worst gradient error %.2e (%s)" % (size, errs[0][0], errs[0][1]))


def test_chain_equals_layer_by_layer_with_dropout(dev):
    """Dropout 0.2 (counter-based masks: the same seeds draw the same masks on both paths): fused chain == layer-by-layer
    product path, outputs and every gradient."""
    import mobilenet_v2
    bb, _ = _backbone(dev, 0.2, seed=3)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((2, 256, 256, 3)).astype(np.float32)).to(dev)
    shapes = {"C3": (2, 32, 32, 32), "C4": (2, 16, 16, 96), "C5": (2, 8, 8, 32)}
    cot = {k: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev) for k, s in shapes.items()}
    assert mobilenet_v2.MB_CHAIN
    out_f, g_f, dx_f = _run(bb, x, cot)
    mobilenet_v2.MB_CHAIN = False
    try:
        bb.__dict__.pop('_chain_cache', None)
        out_l, g_l, dx_l = _run(bb, x, cot)
    finally:
        mobilenet_v2.MB_CHAIN = True
        bb.__dict__.pop('_chain_cache', None)
    for k in shapes:
        assert float((out_l[k] == 0).float().mean()) > 0.1 or k != "C5"      # dropout is on (C5 ends with a Dropout)
        assert_close(out_f[k].cpu().numpy(), out_l[k].cpu().numpy(), 2e-5, "fused vs layer-by-layer " + k)
    scale = max(float(v.abs().max()) for v in g_l.values())
    for name in g_l:
        r = g_l[name].cpu().numpy()
        err = float(np.abs(g_f[name].cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-3 * scale)
        assert err <= 2e-4, "gradient %s: %.2e" % (name, err)
    assert_close(dx_f.cpu().numpy(), dx_l.cpu().numpy(), 2e-4, "image gradient")
    # a second forward with the same seeds and counter repeats the masks (counter-based, stateless)
    out_f2, _, _ = _run(bb, x, cot)
    for k in shapes:
        assert torch.equal(out_f[k], out_f2[k])


def test_chain_inference_mode(dev):
    """training=False: dropout off, no autograd graph; same numbers as the layer-by-layer path."""
    import mobilenet_v2
    bb, _ = _backbone(dev, 0.2, seed=5)
    x = torch.randn(1, 256, 256, 3, device=dev)
    with torch.no_grad():
        a = bb(x, training=False)
        mobilenet_v2.MB_CHAIN = False
        try:
            bb.__dict__.pop('_chain_cache', None)
            b = bb(x, training=False)
        finally:
            mobilenet_v2.MB_CHAIN = True
            bb.__dict__.pop('_chain_cache', None)
    for k in ("C3", "C4", "C5"):
        assert_close(a[k].cpu().numpy(), b[k].cpu().numpy(), 2e-5, "inference " + k)


@pytest.mark.parametrize("size,batch,rate", [(512, 2, 0.2), (256, 2, 0.0), (512, 1, 0.2), (256, 3, 0.2)])
def test_resident_section_equals_launch_ordered_kernels(dev, size, batch, rate, monkeypatch):
    """The small-map part of the chain as phases of one launch per <= 14 kernels (csrc/mb_resident.hip: per-sample clusters on
    one XCD, same-XCD phase barriers) == the launch-ordered kernels: C3 / C4 / C5 to 2e-5 (same arithmetic per element; the
    statistic rows are summed in the cluster's tiling), every gradient to 2e-4 (the backward kernels are the same on both
    sides and read the mean / rstd the forward pass published), run-to-run bit-identical, error word clean."""
    import _rn, ops_mb
    bb, _ = _backbone(dev, rate, seed=11)
    rng = np.random.default_rng(size + batch)
    x = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32)).to(dev)
    shapes = {"C3": (batch, size // 8, size // 8, 32), "C4": (batch, size // 16, size // 16, 96), "C5": (batch, size // 32, size // 32, 32)}
    cot = {k: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev) for k, s in shapes.items()}
    calls = []
    monkeypatch.setattr(ops_mb, "RESIDENT", True)          # (opt-in: measured slower than the launch-ordered kernels)
    monkeypatch.setattr(ops_mb, "_count_resident", calls.append, raising=False)
    out_r, g_r, dx_r = _run(bb, x, cot)
    assert calls and calls[0] >= 20, "the resident section must run here (phases: %s)" % calls
    out_r2, _, _ = _run(bb, x, cot)
    torch.cuda.synchronize()
    assert _rn.resident_errors() == 0
    for k in shapes:
        assert torch.equal(out_r[k], out_r2[k]), "run-to-run " + k
    monkeypatch.setattr(ops_mb, "RESIDENT", False)
    out_l, g_l, dx_l = _run(bb, x, cot)
    for k in shapes:
        assert_close(out_r[k].cpu().numpy(), out_l[k].cpu().numpy(), 2e-5, "resident vs launch-ordered " + k)
    scale = max(float(v.abs().max()) for v in g_l.values())
    for name in g_l:
        r = g_l[name].cpu().numpy()
        err = float(np.abs(g_r[name].cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-3 * scale)
        assert err <= 2e-4, "gradient %s: %.2e" % (name, err)
    assert_close(dx_r.cpu().numpy(), dx_l.cpu().numpy(), 2e-4, "image gradient")
