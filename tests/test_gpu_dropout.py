"""Dropout parity (SURVEY K9 "rate 0 or injected masks"): the configuration the benchmark times trains with
dropout_rate 0.2 (reference train.py:91; sites mobilenet_v2.py:62,71,79,117,184, densenet.py:44,67,77,143).

The product's mask is a pure function of (site seed + step counter, element index) -- include/rn_hip.h at rn_dropout -- and
oracle/dropout_ref.py restates it in numpy.  Here:
  (a) the mask each KIND of kernel applies == the restatement, bit for bit (stand-alone, strided, every GroupNorm
      implementation, the MobileNetV2 chain's norm_act_drop);
  (b) with those masks injected into the oracle at the reference's dropout sites, the MobileNetV2 backbone (fused chain AND
      layer by layer), the DenseNet-121-FPN net (cfg 4's model, at 256 px) and two consecutive trainer steps agree with the
      oracle at the bars of the dropout-0 tests: outputs / losses 1e-4, gradients 5e-4, weights 1e-5.
The full-size cfg-2 step at dropout 0.2 is tests/test_gpu_fullsize.py::test_cfg2_full_size_train_step_matches_oracle[0.2]."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_close, dropout_sites, load_oracle_params, to_oracle_name
from oracle import backbones_ref, dropout_ref, losses_ref, model_ref, tf_ops_ref as T, train_ref

pytestmark = pytest.mark.gpu
LEVELS = ("P3", "P4", "P5", "P6", "P7")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _expected(x, keep, rate):
    """the header's y = u >= rate ? x * (1.f / (1.f - rate)) : 0 in fp32"""
    ks = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return np.where(keep, x * ks, np.float32(0.0)).astype(np.float32)


@pytest.mark.parametrize("seed,counter", [(0, None), (0x5EED + 0x9E3779B1, None), (77, 3), ((1 << 40) + 12345, 9),
                                          (0x5EED, 0x632BE59BD9B4E019 % (1 << 62))])
@pytest.mark.parametrize("rate", [0.2, 0.5])
def test_standalone_mask_is_the_restatement(dev, seed, counter, rate):
    """rn_dropout (forward; its backward is the same call on dy): zero pattern and kept values, bit for bit, for small and
    64-bit seeds with and without the device counter (a rank's counter starts at rank * 0x632BE59BD9B4E019 mod 2^62)."""
    import ops
    rng = np.random.default_rng(1)
    shape = (2, 9, 7, 20)
    x = rng.standard_normal(shape).astype(np.float32)
    sd = torch.tensor([counter], dtype=torch.int64, device=dev) if counter is not None else None
    y = ops.dropout(_t(x, dev), rate, seed, sd).cpu().numpy()
    keep = dropout_ref.keep_mask(seed + (counter or 0), shape, rate)
    assert np.array_equal(y, _expected(x, keep, rate))
    assert abs(keep.mean() - (1 - rate)) < 0.05
    # the oracle's TF form (x / keep_prob) is within one rounding of the product's x * (1 / keep_prob)
    assert_close(dropout_ref.apply(torch.from_numpy(x), keep, rate).numpy(), y, 2e-7, "tf.nn.dropout form")


def test_strided_mask_is_the_dense_tensors(dev):
    """rn_dropout_strided (DenseNet growth slices): the mask of the dense [pixels, c] tensor whatever the row strides are."""
    import _rn
    rng = np.random.default_rng(2)
    px, c, x_ld, x_off, y_ld, y_off, rate, seed = 37, 32, 48, 8, 96, 60, 0.2, 991
    x = rng.standard_normal((px, x_ld)).astype(np.float32)
    y0 = rng.standard_normal((px, y_ld)).astype(np.float32)
    xd, yd = _t(x, dev), _t(y0, dev)
    _rn.check(_rn.lib().rn_dropout_strided(_rn.f32(xd), _rn.f32(yd), px, c, x_ld, x_off, y_ld, y_off, rate, seed, None, _rn.stream()), "strided")
    keep = dropout_ref.keep_mask(seed, (px, c), rate)
    want = y0.copy()
    want[:, y_off:y_off + c] = _expected(x[:, x_off:x_off + c], keep, rate)
    assert np.array_equal(yd.cpu().numpy(), want)


@pytest.mark.parametrize("mode", ["default", "rows", "three_kernel"])
@pytest.mark.parametrize("shape,act", [((2, 16, 16, 64), "elu"), ((2, 64, 64, 32), None), ((3, 5, 7, 144), "elu")])
def test_group_norm_fused_mask_is_the_restatement(dev, monkeypatch, mode, shape, act):
    """y = drop(act(GN(x))) [+ residual] of every GroupNorm implementation: dropped elements are exactly the restatement's,
    kept ones are the undropped kernel's value times 1 / (1 - rate)."""
    import ops
    if mode != "default":
        monkeypatch.setenv("RN_GN_NO_SLICE", "1")
    if mode == "three_kernel":
        monkeypatch.setenv("RN_GN_NO_COOP", "1")
    rng = np.random.default_rng(shape[1])
    c = shape[3]
    x, r = _t(rng.standard_normal(shape).astype(np.float32), dev), _t(rng.standard_normal(shape).astype(np.float32), dev)
    gamma, beta = _t((1 + 0.2 * rng.standard_normal(c)).astype(np.float32), dev), _t((0.1 * rng.standard_normal(c)).astype(np.float32), dev)
    rate, seed = 0.2, 0x5EED + 5 * 0x9E3779B1
    sd = torch.tensor([17], dtype=torch.int64, device=dev)
    y0 = ops.group_norm_act(x, gamma, beta, 32, 1e-5, act, None, 0.0, 0).cpu().numpy()
    y = ops.group_norm_act(x, gamma, beta, 32, 1e-5, act, None, rate, seed, sd).cpu().numpy()
    keep = dropout_ref.keep_mask(seed + 17, shape, rate)
    assert np.array_equal(y, _expected(y0, keep, rate))
    yr = ops.group_norm_act(x, gamma, beta, 32, 1e-5, act, r, rate, seed, sd).cpu().numpy()     # MobileNetV2's linear conv: + identity
    assert_close(yr, _expected(y0, keep, rate) + r.cpu().numpy(), 1e-6, "drop(GN) + residual")


def test_chain_kernels_mask_is_the_restatement(dev):
    """The chain's norm_act_drop / keep4 (csrc/mbconv.hip), isolated through rn_mb_apply: out = drop(act(GN(y))) from y's rows."""
    import _rn
    n, hw, cin, cout, groups, rate, seed = 2, 1024, 32, 96, 32, 0.2, 0x5EED + 11 * 0x9E3779B1
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, hw, 1, cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    L = _rn.lib()
    lay = _rn.MbRows()
    nbytes = L.rn_mb_pointwise_rows(n, hw, cin, cout, groups, C.byref(lay))
    rows = torch.empty(nbytes // 4, device=dev)
    y = torch.empty((n, hw, 1, cout), device=dev)
    xd, wd = _t(x, dev), _t(w, dev)
    lay.rows = rows.data_ptr()
    _rn.check(L.rn_mb_pointwise_fwd(_rn.f32(xd), None, None, None, _rn.f32(wd), _rn.f32(y), n, hw, cin, cout, C.byref(lay), groups, _rn.stream()), "fwd")
    mean, rstd = torch.empty((n, groups), device=dev), torch.empty((n, groups), device=dev)
    g, b = _t((1 + 0.2 * rng.standard_normal(cout)).astype(np.float32), dev), _t((0.1 * rng.standard_normal(cout)).astype(np.float32), dev)
    sd = torch.tensor([5], dtype=torch.int64, device=dev)
    outs = []
    for r in (0.0, rate):
        nm = _rn.MbNorm()
        nm.y, nm.stat, nm.mean, nm.rstd = y.data_ptr(), lay, mean.data_ptr(), rstd.data_ptr()
        nm.gamma, nm.beta, nm.c, nm.groups, nm.act, nm.eps = g.data_ptr(), b.data_ptr(), cout, groups, _rn.ACT["elu"], 1e-5
        nm.drop_rate, nm.drop_seed, nm.drop_seed_dev = r, seed, sd.data_ptr() if r else None
        out = torch.empty_like(y)
        _rn.check(L.rn_mb_apply(C.byref(nm), None, _rn.f32(out), n, hw, _rn.stream()), "apply")
        outs.append(out.cpu().numpy())
    keep = dropout_ref.keep_mask(seed + 5, (n, hw, 1, cout), rate)
    assert np.array_equal(outs[1], _expected(outs[0], keep, rate))


def _mobilenet(dev, rate, classes=3, seed=0):
    import layers, levels, retinanet
    params = model_ref.init_params("mobilenet_v2", num_classes=classes, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in params:
        if k.endswith(".gamma"):
            params[k] = 1 + 0.2 * torch.randn(params[k].shape, generator=g)
        elif k.endswith(".beta"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    net = retinanet.RetinaNet('mobilenet_v2', levels.build_levels(), classes, layers.elu, rate).to(dev)
    load_oracle_params(net, params)
    return net, params


@pytest.mark.parametrize("chain", [True, False])
def test_mobilenet_backbone_with_dropout_matches_oracle(dev, chain, monkeypatch):
    """MobileNetV2 at dropout 0.2, 256 px, batch 2 -- C3 / C4 / C5 and the gradient of every backbone parameter and of the
    image -- through the fused chain (every mask applied by a consumer kernel's operand load / a data-gradient epilogue) and
    layer by layer (masks applied by the GroupNorm kernels), against the oracle with the same masks at the reference's sites
    (mobilenet_v2.py:62,71,79,117,184).  One flipped mask bit anywhere moves an output by ~1e-2: the bars are 1e-4 / 5e-4."""
    import mobilenet_v2
    monkeypatch.setattr(mobilenet_v2, "MB_CHAIN", chain)
    rate, size, batch = 0.2, 256, 2
    net, params = _mobilenet(dev, rate, seed=4)
    bb = net.base.backbone
    bb.__dict__.pop('_chain_cache', None)
    rng = np.random.default_rng(size)
    x = torch.from_numpy(rng.standard_normal((batch, size, size, 3)).astype(np.float32))
    if chain:
        assert bb._chain_start(bb.input_conv(x.to(dev), training=True), True) == 0
    shapes = {"C3": (batch, size // 8, size // 8, 32), "C4": (batch, size // 16, size // 16, 96), "C5": (batch, size // 32, size // 32, 32)}
    cot = {k: torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for k, s in shapes.items()}
    xin = x.to(dev).requires_grad_(True)
    out = bb(xin, training=True)
    sum((out[k] * cot[k].to(dev)).sum() for k in shapes).backward()
    hook = dropout_sites(bb, rate)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items() if k.startswith("backbone")}
    xr = x.clone().requires_grad_(True)
    ref = model_ref.mobilenet_v2_forward(leaves, xr, dropout=hook)
    assert sorted(hook.seen) == sorted(hook.seeds) and len(hook.seeds) == 53      # stem + 17 x 3 + output conv
    for k in shapes:
        assert float((ref[k] == 0).float().mean()) > 0.1 or k != "C5"
        assert_close(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), 1e-4, "dropout 0.2 " + k)
    names = list(leaves.keys())
    gref = dict(zip(names + ["x"], torch.autograd.grad(sum((ref[k] * cot[k]).sum() for k in shapes), [leaves[n] for n in names] + [xr])))
    scale = max(float(gref[n].abs().max()) for n in names)
    errs = []
    for name, p in bb.named_parameters():
        r = gref[to_oracle_name("base.backbone." + name)].numpy()
        errs.append((float(np.abs(p.grad.cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-3 * scale), name))
    errs.sort(reverse=True)
    bad = [e for e in errs if e[0] > 5e-4]
    assert not bad, "%d of %d gradients off: %s" % (len(bad), len(errs), ", ".join("%s %.2e" % (n, e) for e, n in bad[:12]))
    assert_close(xin.grad.cpu().numpy(), gref["x"].numpy(), 5e-4, "image gradient")
    print("MobileNetV2 dropout 0.2 (%s) vs oracle with injected masks: worst gradient error %.2e (%s)"
          % ("chain" if chain else "layer by layer", errs[0][0], errs[0][1]))


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_trainer_steps_with_dropout_match_oracle(dev, use_graph):
    """Two momentum steps of the Trainer at dropout 0.2 (64 px, 3 classes): the optimizer kernel advances the step counter
    by ops.DROPOUT_COUNTER_STEP per step, so step 2 draws the masks of seed + that -- the oracle is handed exactly those.  Losses 1e-4 per step, weights 1e-4
    after both (the second step's gradients pass through the first step's update).  use_graph: the same through the captured
    segments (the capture's warm-up passes must not consume masks: replay i draws what eager step i draws)."""
    import levels as levels_mod, train
    from oracle import dataset_ref
    rate, c, s = 0.2, 3, 64
    net, params = _mobilenet(dev, rate, classes=c, seed=8)
    rng = np.random.default_rng(5)
    image = torch.from_numpy(rng.standard_normal((2, s, s, 3)).astype(np.float32))
    boxes = np.array([[0.1, 0.15, 0.6, 0.7], [0.5, 0.4, 0.95, 0.9]], dtype=np.float32)
    cls, reg, msk = dataset_ref.build_labels((s, s), np.array([0, 2]), boxes, c)
    fc, fr, fm, _ = dataset_ref.flip(cls, reg, msk)
    labels = {"classifications": {k: torch.from_numpy(np.stack([cls[k], fc[k]])) for k in cls},
              "regressions": {k: torch.from_numpy(np.stack([reg[k], fr[k]])) for k in cls},
              "trainable_masks": {k: torch.from_numpy(np.stack([msk[k], fm[k]])) for k in cls}}
    feats = {"image": image.to(dev),
             "detection": {"classifications": {k: v.to(dev) for k, v in labels["classifications"].items()},
                           "regressions": {k: v.to(dev) for k, v in labels["regressions"].items()}},
             "trainable_masks": {k: v.to(torch.uint8).to(dev) for k, v in labels["trainable_masks"].items()}}
    lv = levels_mod.build_levels()
    trainer = train.Trainer(net, lv, optimizer="momentum", learning_rate=1e-2, device=dev, use_graph=use_graph)
    state = {}
    for step in (1, 2):
        import ops
        counter = int(trainer.drop_counter.item())
        assert counter == (step - 1) * ops.DROPOUT_COUNTER_STEP
        out = trainer.step(feats)
        first, _ = train_ref.train_step(params, image, labels, c, state, lr=1e-2, step=step,
                                        dropout=dropout_sites(net, rate, counter=counter))
        assert_close(out["class_loss"].item(), first[1], 1e-4, "class loss, step %d" % step)
        assert_close(out["regr_loss"].item(), first[2], 1e-4, "regression loss, step %d" % step)
    for name, p in net.named_parameters():
        assert_close(p.detach().cpu().numpy(), params[to_oracle_name(name)].numpy(), 1e-4, "weights after two steps: " + name)


def test_densenet_with_dropout_matches_oracle(dev):
    """DenseNet-121-FPN (cfg 4's model) at 256 px, batch 2, dropout 0.2: logits / box outputs of all five levels 1e-4 and the
    gradient of every parameter 5e-4 against the composed oracle (concatenating blocks, densenet.py:117-121) with the masks
    injected after each composite function's 1x1 and 3x3 conv and each transition conv (densenet.py:67,77,143): 119 sites.
    The product applies them with rn_dropout / rn_dropout_strided inside the concat-free block."""
    import layers, levels, retinanet
    classes, size, batch, rate = 8, 256, 2, 0.2
    torch.manual_seed(31)
    net = retinanet.RetinaNet('densenet_121', levels.build_levels(), classes, layers.elu, rate)
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("beta"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    weights = {k: v.detach().clone() for k, v in net.named_parameters()}
    x = torch.randn(batch, size, size, 3, generator=g)
    hook = dropout_sites(net, rate)
    assert len(hook.seeds) == 2 * (6 + 12 + 24 + 16) + 3
    leaves = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    params = {to_oracle_name(k): v for k, v in leaves.items()}
    bparams = {k[len("base."):]: v for k, v in leaves.items() if k.startswith("base.backbone")}
    fe = backbones_ref.backbone_forward('densenet_121', bparams, x, dropout=hook)
    assert sorted(hook.seen) == sorted(hook.seeds)
    pyr = model_ref.fpn_forward(params, fe, "elu")
    ocls = {k: model_ref.subnet_forward(params, v, "classification_subnet", 9, classes, "elu") for k, v in pyr.items()}
    oreg = {k: model_ref.subnet_forward(params, v, "regression_subnet", 9, 4, "elu") for k, v in pyr.items()}
    cot_c = {k: torch.randn(ocls[k].shape, generator=g) for k in LEVELS}
    cot_r = {k: torch.randn(oreg[k].shape, generator=g) for k in LEVELS}
    names = list(leaves.keys())
    loss = sum((ocls[k] * cot_c[k]).sum() + (oreg[k] * cot_r[k]).sum() for k in LEVELS)
    gref = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names])))
    net.to(dev)
    out = net(x.to(dev), training=True)
    sum((out["classifications"][k] * cot_c[k].to(dev)).sum() + (out["regressions"][k] * cot_r[k].to(dev)).sum() for k in LEVELS).backward()
    for k in LEVELS:
        assert_close(out["classifications"][k].detach().cpu().numpy(), ocls[k].detach().numpy(), 1e-4, "densenet dropout cls " + k)
        assert_close(out["regressions"][k].detach().cpu().numpy(), oreg[k].detach().numpy(), 1e-4, "densenet dropout reg " + k)
    scale = max(float(v.abs().max()) for v in gref.values())
    errs = []
    for name, p in net.named_parameters():
        r = gref[name].numpy()
        errs.append((float(np.abs(p.grad.cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-3 * scale), name))
    errs.sort(reverse=True)
    bad = [e for e in errs if e[0] > 5e-4]
    assert not bad, "%d of %d gradients off: %s" % (len(bad), len(errs), ", ".join("%s %.2e" % (n, e) for e, n in bad[:12]))
    print("DenseNet-121-FPN dropout 0.2 vs oracle with injected masks: worst gradient error %.2e (%s)" % errs[0])


@pytest.mark.parametrize("n,hw,cin,cout", [(2, 64, 96, 128), (4, 40, 544, 128), (1, 96, 64, 128)])
def test_conv_dropout_statistics_in_one_launch_equals_the_three(dev, n, hw, cin, cout):
    """rn_conv2d_fwd_dropout (round 6: DenseNet's 1x1 conv -> Dropout -> GroupNorm statistics as ONE launch of the split-bf16 kernel,
    densenet.py:61-67) against the three launches it replaces: the dropped output bit-identical to rn_conv2d_fwd followed by rn_dropout
    (the same mask AND the same conv values), and its statistic rows equal to the per-tile sums of that tensor; then the
    restated mask (oracle/dropout_ref.py) on top of the oracle's conv, 1e-4."""
    import _rn, ops
    L = _rn.lib()
    rng = np.random.default_rng(cin + hw)
    rate, seed, counter = 0.2, 0x1234567, 3 * ops.DROPOUT_COUNTER_STEP
    x = torch.from_numpy(rng.standard_normal((n, hw, hw, cin)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev)
    sd = torch.tensor([counter], dtype=torch.int64, device=dev)
    geom = _rn.ConvGeom(1, 1, 1, cin, 1)
    y = torch.empty((n, hw, hw, cout), dtype=torch.float32, device=dev)
    segs = ops._conv_segs([x], w, None, [y], None, None)
    ops._conv_fwd(segs, 1, geom, dev)
    want = torch.empty_like(y)
    _rn.check(L.rn_dropout(_rn.f32(y), _rn.f32(want), y.numel(), rate, seed, sd.data_ptr(), _rn.stream()), "rn_dropout")
    ok, lay = C.c_int(0), _rn.GnRows(None, 0, 0, 32)
    nbytes = L.rn_conv2d_dropout_rows(segs, 1, C.byref(geom), 32, C.byref(lay), C.byref(ok))
    assert ok.value == 1 and nbytes > 0
    rows = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=dev)
    lay.rows = rows.data_ptr()
    got = torch.full_like(y, float("nan"))
    segs2 = ops._conv_segs([x], w, None, [got], None, None)
    _rn.check(L.rn_conv2d_fwd_dropout(segs2, 1, C.byref(geom), rate, seed, sd.data_ptr(), C.byref(lay), _rn.stream()), "rn_conv2d_fwd_dropout")
    assert torch.equal(got, want)
    kept = float((got != 0).float().mean())
    assert abs(kept - (1 - rate)) < 0.01, kept
    # the rows: (sum, sum of squares) per (tile of rows_per_sample-th of a sample's pixels, channel)
    rps = lay.rows_per_sample
    t = want.reshape(n, rps, (hw * hw) // rps, cout).double()
    ref = torch.stack([t.sum(2), (t * t).sum(2)], -1).reshape(-1, cout, 2)
    assert_close(rows.reshape(-1, cout, 2).cpu().numpy(), ref.cpu().numpy(), 1e-5, "statistic rows of the dropped output")
    # ... and against the oracle: conv, then the restated mask
    yc = T.conv2d_same(x.cpu(), w.cpu(), 1)
    mask = dropout_ref.keep_mask(seed + counter, tuple(yc.shape), rate)
    assert_close(got.cpu().numpy(), (yc * torch.from_numpy(mask.astype(np.float32)) / (1 - rate)).numpy(), 1e-4, "dropped conv output vs oracle")
