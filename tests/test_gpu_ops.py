"""Parity of every HIP kernel family with the CPU oracle (runs on the MI355X box: -m gpu).

Tolerances: fp32 results within 1e-4 relative (max-norm), as BASELINE.json's north_star
states; masks / class maps / arg-max / NMS indices bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, coco_like_objects, random_boxes
from oracle import dataset_ref, levels_ref, losses_ref, tf_ops_ref, train_ref, utils_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    import _rn
    _rn.lib()          # fails loudly if librn_hip.so is missing
    return torch.device("cuda:0")


def _t(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(grad)


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, bias
    (2, 16, 16, 256, 256, 3, 1, False),      # head tower conv
    (2, 8, 8, 256, 720, 3, 1, True),         # class out conv (A*C = 720)
    (2, 8, 8, 256, 36, 3, 1, True),          # box out conv
    (2, 6, 5, 256, 27, 3, 1, True),          # class out conv, 3 classes: cout % 4 != 0 (scalar paths)
    (2, 33, 31, 3, 32, 3, 2, False),         # stem, odd size, scalar (cin=3) path
    (2, 16, 16, 32, 256, 3, 2, False),       # P6 from C5
    (2, 7, 9, 256, 256, 3, 2, False),        # stride 2 on odd maps
    (2, 32, 32, 16, 96, 1, 1, False),        # MobileNet expand
    (2, 16, 16, 144, 24, 1, 1, False),       # MobileNet linear, C=144
    (1, 12, 10, 96, 256, 1, 1, False),       # FPN lateral
    (1, 20, 20, 8, 12, 7, 2, False),         # 7x7/2 (ResNeXt / DenseNet stem family)
    (2, 7, 9, 64, 128, 3, 1, True),          # odd maps (Winograd edge tiles)
    (3, 1, 1, 256, 256, 3, 1, False),        # P7-sized map: one partial tile per image
    (1, 75, 75, 256, 256, 3, 1, False),      # P3 of a 600x600 image
    (2, 48, 48, 64, 128, 3, 2, False),       # stride 2 on a large map: phase-decomposed data gradient
    (1, 65, 67, 32, 64, 3, 2, True),         # ... odd sizes (unequal phases)
    (2, 48, 48, 64, 32, 1, 2, False),        # ... 1x1 / stride 2: three of the four phases have no taps (zeros)
    (1, 70, 70, 8, 12, 7, 2, False),         # ... 7x7 / stride 2
    (1, 48, 48, 1024, 1024, 3, 1, False),    # big kernel, many pixels: wgrad splits longer than its LDS pixel table (windows)
    (1, 10, 10, 1024, 2048, 3, 1, False),    # ResNeXt stage-5 identity conv: one split, wgrad written straight into dw
]


def _winograd_eligible(case):
    n, h, w, cin, cout, k, stride, use_bias = case
    return k == 3 and stride == 1 and cin % 4 == 0 and cout % 4 == 0 and min(cin, cout) >= 64


@pytest.fixture(params=["winograd4", "winograd2", "direct"])
def conv_path(request):
    """3x3 / stride-1 convs have three kernels (Winograd F(4x4,3x3), F(2x2,3x3), direct implicit GEMM): test all."""
    import ops
    old = ops.WINOGRAD, ops.WINOGRAD_TILE
    ops.WINOGRAD = request.param != "direct"
    ops.WINOGRAD_TILE = 2 if request.param == "winograd2" else 4
    yield request.param
    ops.WINOGRAD, ops.WINOGRAD_TILE = old


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_fwd_bwd(dev, case, conv_path):
    import ops
    if conv_path != "direct" and not _winograd_eligible(case):
        pytest.skip("direct kernel only")
    n, h, w, cin, cout, k, stride, use_bias = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if use_bias else None
    xc, wc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    bc = torch.from_numpy(b).requires_grad_(True) if use_bias else None
    yc = tf_ops_ref.conv2d_same(xc, wc, stride, bc)
    dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    yc.backward(torch.from_numpy(dy))

    xg, wg = _t(x, dev, True), _t(wt, dev, True)
    bg = _t(b, dev, True) if use_bias else None
    yg = ops.conv2d(xg, wg, bg, stride)
    assert tuple(yg.shape) == tuple(yc.shape)
    yg.backward(_t(dy, dev))
    assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "conv fwd")
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "conv dgrad")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "conv wgrad")
    if use_bias:
        assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), TOL, "conv bias grad")


def test_conv2d_shared_kernel_over_pyramid_levels(dev, conv_path):
    """The head layers: one kernel, five maps of different sizes, ONE launch; wgrad sums levels."""
    import ops
    rng = np.random.default_rng(7)
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    wt = (rng.standard_normal((3, 3, 256, 256)) / 48).astype(np.float32)
    xs = [rng.standard_normal((2, h, w, 256)).astype(np.float32) for h, w in sizes]
    wc = torch.from_numpy(wt).requires_grad_(True)
    xcs = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    ycs = [tf_ops_ref.conv2d_same(x, wc, 1) for x in xcs]
    dys = [rng.standard_normal(tuple(y.shape)).astype(np.float32) for y in ycs]
    torch.autograd.backward(ycs, [torch.from_numpy(d) for d in dys])
    wg = _t(wt, dev, True)
    xgs = [_t(x, dev, True) for x in xs]
    ygs = ops.conv2d(xgs, wg, None, 1)
    torch.autograd.backward(ygs, [_t(d, dev) for d in dys])
    for yg, yc, xg, xc in zip(ygs, ycs, xgs, xcs):
        assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "multi fwd")
        assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "multi dgrad")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "multi wgrad")


@pytest.mark.parametrize("case", [(2, 17, 15, 32, 1), (2, 16, 16, 144, 2), (1, 9, 9, 960, 1), (2, 8, 8, 96, 2)])
def test_depthwise_fwd_bwd(dev, case):
    import ops
    n, h, w, c, stride = case
    rng = np.random.default_rng(c + stride)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = rng.standard_normal((3, 3, c, 1)).astype(np.float32) / 3
    xc, wc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    yc = tf_ops_ref.depthwise_conv2d_same(xc, wc, stride)
    dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    yc.backward(torch.from_numpy(dy))
    xg, wg = _t(x, dev, True), _t(wt, dev, True)
    yg = ops.depthwise_conv2d(xg, wg, stride)
    yg.backward(_t(dy, dev))
    assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "dw fwd")
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "dw dgrad")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "dw wgrad")
    # autograd used the one-launch rn_depthwise_bwd; the two separate entry points give the same bits
    import _rn
    L = _rn.lib()
    dyg = _t(dy, dev)
    dx2, dw2 = torch.empty_like(xg), torch.empty_like(wg)
    _rn.check(L.rn_depthwise_dgrad(_rn.f32(dyg), _rn.f32(wg.detach()), _rn.f32(dx2), n, h, w, c, 3, stride, _rn.stream()), "dgrad")
    ws = _rn.workspace(L.rn_depthwise_wgrad_workspace(n, h, w, c, 3, stride), dev)
    _rn.check(L.rn_depthwise_wgrad(_rn.f32(xg.detach()), _rn.f32(dyg), _rn.f32(dw2), n, h, w, c, 3, stride, ws.data_ptr(), ws.numel(),
                                   _rn.stream(), None), "wgrad")
    assert torch.equal(dx2, xg.grad) and torch.equal(dw2, wg.grad)


@pytest.mark.parametrize("c,act,res", [(256, "elu", False), (144, "elu", True), (32, "relu", False), (24, None, True),
                                       (960, "relu6", False), (16, "elu", False)])
def test_group_norm_act_fwd_bwd(dev, c, act, res):
    import ops
    rng = np.random.default_rng(c)
    shapes = [(2, 9, 7, c), (2, 3, 3, c)] if c == 256 else [(2, 9, 7, c)]
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    xs = [(rng.standard_normal(s) * 2 + 0.5).astype(np.float32) for s in shapes]
    rs = [rng.standard_normal(s).astype(np.float32) for s in shapes] if res else None
    gc, bc = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    xcs = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    rcs = [torch.from_numpy(r).requires_grad_(True) for r in rs] if res else None
    ycs = []
    for i, x in enumerate(xcs):
        y = tf_ops_ref.activation(tf_ops_ref.group_norm(x, gc, bc), act)
        ycs.append(y + rcs[i] if res else y)
    dys = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    torch.autograd.backward(ycs, [torch.from_numpy(d) for d in dys])

    gg, bg = _t(gamma, dev, True), _t(beta, dev, True)
    xgs = [_t(x, dev, True) for x in xs]
    rgs = [_t(r, dev, True) for r in rs] if res else None
    ygs = ops.group_norm_act(xgs, gg, bg, 32, 1e-5, act, rgs)
    torch.autograd.backward(ygs, [_t(d, dev) for d in dys])
    for i in range(len(xs)):
        assert_close(ygs[i].detach().cpu().numpy(), ycs[i].detach().numpy(), TOL, "gn fwd")
        assert_close(xgs[i].grad.cpu().numpy(), xcs[i].grad.numpy(), TOL, "gn dx")
        if res:
            assert_close(rgs[i].grad.cpu().numpy(), rcs[i].grad.numpy(), TOL, "gn dres")
    assert_close(gg.grad.cpu().numpy(), gc.grad.numpy(), TOL, "gn dgamma")
    assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), TOL, "gn dbeta")


@pytest.mark.parametrize("mode", ["default", "grid_resident", "three_kernel"])
def test_group_norm_dropout_is_consistent(dev, monkeypatch, mode):
    """Dropout mask: right keep fraction, inverted scaling, and backward uses the SAME mask -- and the SAME mask in all
    three GroupNorm implementations (the mask is a function of the element index only)."""
    import ops
    if mode != "default":
        monkeypatch.setenv("RN_GN_NO_SLICE", "1")
    if mode == "three_kernel":
        monkeypatch.setenv("RN_GN_NO_COOP", "1")
    torch.manual_seed(0)
    c = 64
    x = torch.randn(2, 32, 32, c, device=dev, requires_grad=True)
    gamma = torch.ones(c, device=dev, requires_grad=True)
    beta = torch.zeros(c, device=dev, requires_grad=True)
    y = ops.group_norm_act(x, gamma, beta, 32, 1e-5, None, None, 0.25, 1234)
    y0 = ops.group_norm_act(x, gamma, beta, 32, 1e-5, None, None, 0.0, 0)
    kept = (y != 0)
    frac = kept.float().mean().item()
    assert abs(frac - 0.75) < 0.01
    assert torch.allclose(y[kept], y0[kept] / 0.75, rtol=1e-5, atol=1e-6)
    y2 = ops.group_norm_act(x, gamma, beta, 32, 1e-5, None, None, 0.25, 1234)
    assert torch.equal(y, y2)                                     # counter-based, reproducible
    # backward: beta's gradient of sum(y) is sum(mask)/keep per channel
    y.sum().backward()
    expect = kept.float().sum((0, 1, 2)) / 0.75
    assert torch.allclose(beta.grad, expect, rtol=1e-4)
    _DROPOUT_MASKS.append(kept.cpu())
    assert torch.equal(_DROPOUT_MASKS[0], _DROPOUT_MASKS[-1])       # same mask whichever kernel path produced it


_DROPOUT_MASKS = []


@pytest.mark.parametrize("shape", [((2, 8, 8), (4, 4)), ((2, 64, 64), (32, 32)), ((1, 75, 75), (38, 38)), ((1, 5, 7), (3, 4))])
def test_upsample_add(dev, shape):
    import ops
    (n, h, w), (th, tw) = shape
    c = 8
    rng = np.random.default_rng(h)
    lat = rng.standard_normal((n, h, w, c)).astype(np.float32)
    top = rng.standard_normal((n, th, tw, c)).astype(np.float32)
    lc, tc = torch.from_numpy(lat).requires_grad_(True), torch.from_numpy(top).requires_grad_(True)
    yc = lc + tf_ops_ref.upsample_nearest_align_corners(tc, h, w)
    dy = rng.standard_normal((n, h, w, c)).astype(np.float32)
    yc.backward(torch.from_numpy(dy))
    lg, tg = _t(lat, dev, True), _t(top, dev, True)
    yg = ops.upsample_add(lg, tg)
    yg.backward(_t(dy, dev))
    assert np.array_equal(yg.detach().cpu().numpy(), yc.detach().numpy())
    assert np.array_equal(lg.grad.cpu().numpy(), lc.grad.numpy())
    assert_close(tg.grad.cpu().numpy(), tc.grad.numpy(), 1e-6, "upsample dtop")


def _loss_inputs(rng, c, levels=((2, 8, 8, 9), (2, 4, 4, 9), (2, 2, 2, 9)), fg_rate=0.05):
    out = []
    for shp in levels:
        rows = int(np.prod(shp))
        z = (rng.standard_normal((rows, c)) * 2 - 2).astype(np.float32)
        lab = np.zeros((rows, c), np.float32)
        fg = rng.uniform(size=rows) < fg_rate
        lab[fg, rng.integers(0, c, fg.sum())] = 1.0
        rp = rng.standard_normal((rows, 4)).astype(np.float32)
        rl = (rng.standard_normal((rows, 4)) * 1.5).astype(np.float32)
        m = (rng.uniform(size=rows) < 0.8) | fg
        out.append((z.reshape(shp + (c,)), lab.reshape(shp + (c,)), rp.reshape(shp + (4,)), rl.reshape(shp + (4,)),
                    m.reshape(shp)))
    return out


@pytest.mark.parametrize("mode", ["bce_dice", "focal"])
@pytest.mark.parametrize("c", [3, 4, 12, 80, 128, 130])
def test_detection_loss_fwd_bwd(dev, mode, c):
    """(c % 4 == 0 and <= 128: the four-lanes-per-row kernels, 1 / 1 / 5 / 8 quads per lane; 3 and 130: the wave-per-row kernels)"""
    import ops
    rng = np.random.default_rng(c)
    data = _loss_inputs(rng, c)
    zc = [torch.from_numpy(d[0]).requires_grad_(True) for d in data]
    rc = [torch.from_numpy(d[2]).requires_grad_(True) for d in data]
    masks = [torch.from_numpy(d[4]) for d in data]
    cat = lambda ts: torch.cat([t[m] for t, m in zip(ts, masks)], 0)
    cl, rl = losses_ref.loss(cat([torch.from_numpy(d[1]) for d in data]), cat([torch.from_numpy(d[3]) for d in data]),
                             cat(zc), cat(rc), mode)
    (1.7 * cl + 0.6 * rl).backward()
    zg = [_t(d[0], dev, True) for d in data]
    rg = [_t(d[2], dev, True) for d in data]
    clg, rlg, stats = ops.detection_loss(zg, rg, [_t(d[1], dev) for d in data], [_t(d[3], dev) for d in data],
                                         [_t(d[4].astype(np.uint8), dev) for d in data], c, mode)
    (1.7 * clg + 0.6 * rlg).backward()
    assert_close(clg.item(), cl.item(), TOL, "class loss")
    assert_close(rlg.item(), rl.item(), TOL, "regr loss")
    assert stats[2].item() == sum(int(m.sum()) for m in masks)
    for a, b in zip(zg, zc):
        assert_close(a.grad.cpu().numpy(), b.grad.numpy(), TOL, "d cls logits")
    for a, b in zip(rg, rc):
        assert_close(a.grad.cpu().numpy(), b.grad.numpy(), TOL, "d reg")


def test_regression_loss_reference_kat(dev):
    """losses_test.py:17-27 through the HIP kernel: == 2.0 (1 class, label 1 => fg)."""
    import ops
    import reference_kats as K
    lab = K.HUBER_FG.astype(np.float32).reshape(3, 1)
    z = np.zeros((3, 1), np.float32)
    # labels/logits broadcast over 4 coords with the same numbers keeps the mean: 4*sum/(4*#fg)
    rp = np.repeat(K.HUBER_LOGITS, 4, 1).astype(np.float32)
    rl = np.repeat(K.HUBER_LABELS, 4, 1).astype(np.float32)
    _, reg, _ = ops.detection_loss([_t(z, dev)], [_t(rp, dev)], [_t(lab, dev)], [_t(rl, dev)],
                                   [_t(np.ones(3, np.uint8), dev)], 1, "bce_dice")
    assert reg.item() == K.HUBER_EXPECTED


@pytest.mark.parametrize("mode", ["trunc_int", "float"])
def test_anchor_assignment_bit_exact(dev, mode):
    import dataset
    import levels
    dataset.ANCHOR_SIZE_MODE = mode
    lv = levels.build_levels()
    rng = np.random.default_rng(5)
    size = (256, 320)
    nimg, max_obj, c = 4, 12, 80
    boxes = np.zeros((nimg, max_obj, 4), np.float32)
    cids = np.zeros((nimg, max_obj), np.int32)
    nobj = np.zeros(nimg, np.int32)
    for i in range(nimg):
        b, k = coco_like_objects(rng, 256, max_obj)
        nobj[i] = len(b)
        boxes[i, :len(b)], cids[i, :len(b)] = b, k
    # exact ties and the ignore band: duplicate a box (arg-max must keep the FIRST) and add
    # anchor-shaped boxes whose IoU lands on / near the thresholds
    boxes[0, 1] = boxes[0, 0]
    cids[0, 1] = (cids[0, 0] + 1) % c
    try:
        for pn in lv:
            factor = 2 ** int(pn[-1])
            cls, reg, msk, arg = dataset.level_labels(size, _t(cids, dev), _t(boxes, dev), lv[pn], factor, c,
                                                      num_obj=_t(nobj, dev), return_argmax=True)
            for i in range(nimg):
                oc, orr, om, oarg = dataset_ref.level_labels(size, cids[i, :nobj[i]], boxes[i, :nobj[i]],
                                                            lv[pn].anchor_sizes, factor, c, mode)
                assert np.array_equal(arg[i].cpu().numpy(), oarg), pn
                assert np.array_equal(msk[i].cpu().numpy().astype(bool), om), pn
                assert np.array_equal(cls[i].cpu().numpy(), oc), pn
                assert_close(reg[i].cpu().numpy(), orr, TOL, "regression targets " + pn)
        # build_labels fills every level from ONE launch (rn_anchor_assign_levels): same bits as the per-level calls
        ac, ar, am = dataset.build_labels(size, _t(cids, dev), _t(boxes, dev), lv, c, num_obj=_t(nobj, dev))
        assert list(ac.keys()) == list(lv)
        for pn in lv:
            cls, reg, msk = dataset.level_labels(size, _t(cids, dev), _t(boxes, dev), lv[pn], 2 ** int(pn[-1]), c,
                                                 num_obj=_t(nobj, dev))
            assert torch.equal(ac[pn], cls) and torch.equal(ar[pn], reg) and torch.equal(am[pn], msk), pn
    finally:
        dataset.ANCHOR_SIZE_MODE = "trunc_int"


def test_assignment_flip_pair_is_the_flipped_maps(dev):
    """dataset.py:182-204 / augmentation.py:5-22: build_labels(flip_pair=True) writes [labels, flip(labels)] from ONE assignment:
    slot 2i is the plain assignment of image i, slot 2i+1 the oracle's flip of those maps (W reversed, x shift negated), bit
    for bit -- and equal to what augmentation.flip (the flip kernel) makes of the plain maps."""
    import augmentation, dataset, levels
    lv = levels.build_levels()
    rng = np.random.default_rng(11)
    size, c, nimg, max_obj = (192, 160), 7, 3, 9
    boxes = np.zeros((nimg, max_obj, 4), np.float32)
    cids = np.zeros((nimg, max_obj), np.int32)
    nobj = np.zeros(nimg, np.int32)
    for i in range(nimg):
        b, k = coco_like_objects(rng, 160, max_obj)
        nobj[i] = len(b)
        boxes[i, :len(b)], cids[i, :len(b)] = b, k % c
    pc, pr, pm = dataset.build_labels(size, _t(cids, dev), _t(boxes, dev), lv, c, num_obj=_t(nobj, dev), flip_pair=True)
    ac, ar, am = dataset.build_labels(size, _t(cids, dev), _t(boxes, dev), lv, c, num_obj=_t(nobj, dev))
    for i in range(nimg):
        oc, orr, om = dataset_ref.build_labels(size, cids[i, :nobj[i]], boxes[i, :nobj[i]], c)
        fc, fr, fm, _ = dataset_ref.flip(oc, orr, om)
        one = {'image': torch.zeros((size[0], size[1], 3), device=dev),
               'detection': {'classifications': {k: ac[k][i] for k in lv}, 'regressions': {k: ar[k][i] for k in lv}},
               'trainable_masks': {k: am[k][i] for k in lv}}
        fl = augmentation.flip(one)
        for k in lv:
            assert pc[k].shape[0] == 2 * nimg
            assert torch.equal(pc[k][2 * i], ac[k][i]) and torch.equal(pr[k][2 * i], ar[k][i]) and torch.equal(pm[k][2 * i], am[k][i]), k
            assert np.array_equal(pc[k][2 * i + 1].cpu().numpy(), fc[k]), k
            assert np.array_equal(pm[k][2 * i + 1].cpu().numpy().astype(bool), fm[k]), k
            assert_close(pr[k][2 * i + 1].cpu().numpy(), fr[k], TOL, "flipped regression targets " + k)
            assert torch.equal(pc[k][2 * i + 1], fl['detection']['classifications'][k]), k
            assert torch.equal(pr[k][2 * i + 1], fl['detection']['regressions'][k]), k
            assert torch.equal(pm[k][2 * i + 1], fl['trainable_masks'][k]), k


def test_assignment_degenerate_box_is_the_references_nan(dev):
    """dataset.py:118-121 picks the arg-max object's target with reduce_sum(regression * one_hot, 0): a zero-extent box
    (log 0 = -inf) makes the log-size component NaN for every anchor NOT assigned to it (-inf * 0).  Kernel == oracle,
    NaN for NaN; the finite components and the masks / class maps are untouched."""
    import dataset, levels
    lv = levels.build_levels()
    size, c = (128, 128), 4
    boxes = np.array([[[0.1, 0.1, 0.5, 0.6], [0.3, 0.7, 0.3, 0.9], [0.55, 0.2, 0.95, 0.6]]], np.float32)   # object 1: zero height
    cids = np.array([[0, 1, 2]], np.int32)
    ac, ar, am = dataset.build_labels(size, _t(cids, dev), _t(boxes, dev), lv, c)
    oc, orr, om = dataset_ref.build_labels(size, cids[0], boxes[0], c)
    for k in lv:
        got, want = ar[k][0].cpu().numpy(), orr[k]
        assert np.array_equal(np.isnan(got), np.isnan(want)), k
        assert np.isnan(want[..., 2]).any() and not np.isnan(want[..., :2]).any() and not np.isnan(want[..., 3]).any(), k
        ok = ~np.isnan(want)
        assert_close(got[ok], want[ok], TOL, "finite regression targets " + k)
        assert np.array_equal(ac[k][0].cpu().numpy(), oc[k]) and np.array_equal(am[k][0].cpu().numpy().astype(bool), om[k]), k


def test_assignment_reference_kat(dev):
    """dataset_test.py:8-45 class map through the HIP kernel."""
    import dataset
    import levels
    import reference_kats as K
    lvl = levels.Level(K.ASSIGN_LEVEL["base"], K.ASSIGN_LEVEL["aspects"], K.ASSIGN_LEVEL["scales"])
    cls, reg, msk = dataset.level_labels(K.ASSIGN_IMAGE_SIZE, _t(K.ASSIGN_CLASS_IDS.astype(np.int32)[None], dev),
                                         _t(K.ASSIGN_BOXES.astype(np.float32)[None], dev), lvl, K.ASSIGN_FACTOR, 401)
    cls = cls[0].cpu().numpy()
    ids = np.where(cls.max(-1) > 0, cls.argmax(-1), 0)
    assert np.array_equal(ids, K.ASSIGN_CLASSMAP_EXPECTED)


def test_decode_boxes(dev):
    import levels
    import utils
    lv = levels.build_levels()
    rng = np.random.default_rng(11)
    for (h, w) in ((3, 4), (16, 20), (1, 1)):
        reg = (rng.standard_normal((2, h, w, 9, 4)) * 0.5).astype(np.float32)
        anchors = lv["P4"].normalized_anchor_sizes((h * 16, w * 16))
        got = utils.regression_postprocess(_t(reg, dev), anchors).cpu().numpy()
        exp = utils_ref.regression_postprocess(reg, anchors)
        assert_close(got, exp, TOL, "decode")
    import reference_kats as K
    # utils_test.py:44-74 through the kernel: zero regression == the anchor box map
    z = np.zeros((1, 3, 4, 1, 4), np.float32)
    got = utils.regression_postprocess(_t(z, dev), K.ANCHOR_BOXMAP_ANCHORS).cpu().numpy()
    assert np.allclose(got, K.ANCHOR_BOXMAP_EXPECTED, atol=1e-6)


def _detection_inputs(rng, n, c, sizes, hot=0.02):
    probs, boxes = {}, {}
    for i, (h, w) in enumerate(sizes):
        k = "P%d" % (3 + i)
        p = rng.uniform(0, 0.45, (n, h, w, 9, c)).astype(np.float32)
        sel = rng.uniform(size=(n, h, w, 9)) < hot
        cls = rng.integers(0, c, sel.sum())
        p[sel, cls] = rng.uniform(0.5, 1.0, sel.sum()).astype(np.float32)
        # clustered boxes so that suppression actually happens
        centres = rng.uniform(0.2, 0.8, (6, 2))
        pick = rng.integers(0, 6, (n, h, w, 9))
        ctr = centres[pick] + rng.normal(0, 0.02, (n, h, w, 9, 2))
        sz = rng.uniform(0.1, 0.3, (n, h, w, 9, 2))
        b = np.concatenate([ctr - sz / 2, ctr + sz / 2], -1).astype(np.float32)
        probs[k], boxes[k] = p, b
    return probs, boxes


@pytest.mark.parametrize("n,c,hot", [(2, 5, 0.05), (3, 80, 0.02), (1, 3, 0.6), (3, 2800, 0.3)])
def test_detect_bit_exact_indices(dev, n, c, hot):
    """boxes_decode + merge + nms_classwise for a batch == the oracle, index for index.  (3 x 2800 = 8400 (image, class)
    segments: more than the scatter kernel scans in LDS -- the stand-alone segment scan + global scatter path.)"""
    import utils
    rng = np.random.default_rng(n * 100 + c)
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    probs, boxes = _detection_inputs(rng, n, c, sizes, hot)
    # score ties inside one class
    probs["P3"][0, 0, 0, 0, :] = 0; probs["P3"][0, 0, 0, 0, 1] = 0.9
    probs["P3"][0, 0, 1, 0, :] = 0; probs["P3"][0, 0, 1, 0, 1] = 0.9
    got = utils.detect({k: _t(v, dev) for k, v in probs.items()}, {k: _t(v, dev) for k, v in boxes.items()}, c)
    total_kept = 0
    for i in range(n):
        parts = [utils_ref.boxes_decode(probs[k][i], boxes[k][i]) for k in probs]
        exp = utils_ref.nms_classwise(utils_ref.merge_boxes_decoded(parts), c)
        assert np.array_equal(got[i].class_ids.cpu().numpy(), exp.class_ids)
        assert np.array_equal(got[i].scores.cpu().numpy(), exp.scores)
        assert np.array_equal(got[i].boxes.cpu().numpy(), exp.boxes)
        total_kept += len(exp.scores)
    assert total_kept > 0


def test_detect_raw_equals_decode_then_detect(dev):
    """utils.detect_raw (candidates decoded on the fly inside the emit pass) == regression_postprocess + detect, bit for
    bit, and == the oracle; ragged pyramid, two images, non-square grids."""
    import levels
    import utils
    rng = np.random.default_rng(21)
    n, c = 2, 5
    lv = levels.build_levels()
    size = (96, 160)
    probs, regs, anchors = {}, {}, {}
    for k in lv:
        f = 2 ** int(k[1])
        gh, gw = -(-size[0] // f), -(-size[1] // f)
        p = rng.uniform(0, 0.45, (n, gh, gw, 9, c)).astype(np.float32)
        hot = rng.uniform(size=(n, gh, gw, 9)) < 0.15
        p[hot, rng.integers(0, c, int(hot.sum()))] = rng.uniform(0.5, 1.0, int(hot.sum())).astype(np.float32)
        probs[k] = p
        regs[k] = (rng.standard_normal((n, gh, gw, 9, 4)) * 0.3).astype(np.float32)
        anchors[k] = lv[k].normalized_anchor_sizes(size)
    tp = {k: _t(v, dev) for k, v in probs.items()}
    tr = {k: _t(v, dev) for k, v in regs.items()}
    dec = {k: utils.regression_postprocess(tr[k], anchors[k]) for k in lv}
    want = utils.detect(tp, dec, c)
    got = utils.detect_raw(tp, tr, anchors, c)
    kept = 0
    for i in range(n):
        assert torch.equal(got[i].boxes, want[i].boxes) and torch.equal(got[i].scores, want[i].scores)
        assert torch.equal(got[i].class_ids, want[i].class_ids)
        parts = [utils_ref.boxes_decode(probs[k][i], dec[k][i].cpu().numpy()) for k in lv]
        exp = utils_ref.nms_classwise(utils_ref.merge_boxes_decoded(parts), c)
        assert np.array_equal(got[i].boxes.cpu().numpy(), exp.boxes) and np.array_equal(got[i].class_ids.cpu().numpy(), exp.class_ids)
        kept += len(exp.scores)
    assert kept > 0


def test_boxes_decode_and_nms_api(dev):
    import utils
    rng = np.random.default_rng(3)
    probs, boxes = _detection_inputs(rng, 1, 4, [(6, 6)], 0.3)
    p, b = probs["P3"][0], boxes["P3"][0]
    dec = utils.boxes_decode(_t(p, dev), _t(b, dev))
    exp = utils_ref.boxes_decode(p, b)
    assert np.array_equal(dec.boxes.cpu().numpy(), exp.boxes)
    assert np.array_equal(dec.scores.cpu().numpy(), exp.scores)
    assert np.array_equal(dec.class_ids.cpu().numpy(), exp.class_ids)
    kept = utils.nms(dec)
    ek = utils_ref.nms(exp, fast=False)
    assert np.array_equal(kept.boxes.cpu().numpy(), ek.boxes) and np.array_equal(kept.class_ids.cpu().numpy(), ek.class_ids)
    cw = utils.nms_classwise(dec, 4)
    ecw = utils_ref.nms_classwise(exp, 4)
    assert np.array_equal(cw.scores.cpu().numpy(), ecw.scores) and np.array_equal(cw.boxes.cpu().numpy(), ecw.boxes)
    # empty input
    empty = utils.BoxesDecoded(dec.boxes[:0], dec.scores[:0], dec.class_ids[:0])
    assert utils.nms(empty).boxes.shape[0] == 0


def test_nms_max_output_and_degenerate_boxes(dev):
    import utils
    rng = np.random.default_rng(9)
    k = 1500
    c = rng.uniform(0, 1, (k, 2))
    b = np.concatenate([c, c + 1e-3], 1).astype(np.float32)      # tiny, disjoint: nothing suppressed
    b[5] = [0.3, 0.3, 0.3, 0.3]                                   # zero area
    b[6] = [0.6, 0.6, 0.5, 0.5]                                   # inverted corners
    s = rng.uniform(0.5, 1, k).astype(np.float32)
    dec = utils.BoxesDecoded(_t(b, dev), _t(s, dev), torch.zeros(k, dtype=torch.int64, device=dev))
    kept = utils.nms(dec)
    exp = utils_ref.nms_indices(b, s)
    assert kept.boxes.shape[0] == utils.NMS_MAX_OUTPUT_SIZE == len(exp)
    assert np.array_equal(kept.boxes.cpu().numpy(), b[exp])


@pytest.mark.parametrize("kind", ["momentum", "rmsprop", "adam"])
def test_optimizer_matches_tf_semantics(dev, kind):
    import train
    torch.manual_seed(0)
    lin = torch.nn.Module()
    lin.a = torch.nn.Parameter(torch.randn(3, 3, 8, 16))
    lin.a.l2_scale = 1e-4
    lin.b = torch.nn.Parameter(torch.randn(700))
    params = {"a": lin.a.detach().clone(), "b": lin.b.detach().clone()}
    lin.to(dev)
    arena = train.ParamArena(lin, dev)
    opt = train.Optimizer(arena, kind, 1e-2, grad_clip_norm=0.5)
    state = {}
    for step in range(1, 4):
        grads = {"a": torch.randn(3, 3, 8, 16), "b": torch.randn(700)}
        lin.a.grad.copy_(grads["a"].to(dev)); lin.b.grad.copy_(grads["b"].to(dev))
        opt.step(grad_scale=0.5)
        tot = {"a": grads["a"] * 0.5 + 1e-4 * params["a"], "b": grads["b"] * 0.5}
        clipped, gn = train_ref.clip_by_global_norm([tot["a"], tot["b"]], 0.5)
        assert_close(opt.norm_reg[0].item() ** 0.5, gn.item(), TOL, "global norm")
        train_ref.apply_optimizer(kind, params, {"a": clipped[0], "b": clipped[1]}, state, 1e-2, step)
        assert_close(lin.a.detach().cpu().numpy(), params["a"].numpy(), TOL, "weights a step %d" % step)
        assert_close(lin.b.detach().cpu().numpy(), params["b"].numpy(), TOL, "weights b step %d" % step)


@pytest.mark.parametrize("case", [(2, 9, 8, 128, 128, 3, 1, 32), (2, 9, 8, 256, 256, 3, 2, 32), (1, 7, 7, 1024, 1024, 3, 1, 32),
                                  (2, 48, 50, 128, 128, 3, 2, 32), (1, 64, 64, 256, 256, 3, 2, 32),
                                  (2, 6, 6, 64, 128, 3, 1, 4)])
def test_grouped_conv_fwd_bwd(dev, case):
    """ResNeXt cardinality conv: one grouped launch == torch grouped conv (and see test_gpu_backbones for
    the literal 32-split reference form)."""
    import ops
    import torch.nn.functional as F
    n, h, w, cin, cout, k, stride, groups = case
    rng = np.random.default_rng(cin + stride)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin // groups, cout)) / np.sqrt(k * k * cin / groups)).astype(np.float32)
    xc, wc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    _, pt, pb = tf_ops_ref.same_pad_1d(h, k, stride)
    _, pl, pr = tf_ops_ref.same_pad_1d(w, k, stride)
    yc = F.conv2d(F.pad(xc.permute(0, 3, 1, 2), (pl, pr, pt, pb)), wc.permute(3, 2, 0, 1), stride=stride,
                  groups=groups).permute(0, 2, 3, 1)
    dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    yc.backward(torch.from_numpy(dy))
    xg, wg = _t(x, dev, True), _t(wt, dev, True)
    yg = ops.conv2d(xg, wg, None, stride, groups)
    yg.backward(_t(dy, dev))
    assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "grouped fwd")
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "grouped dgrad")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "grouped wgrad")


@pytest.mark.parametrize("c,groups", [(2048, 32), (256, 256), (1664, 32)])
def test_group_norm_wide_and_per_channel(dev, c, groups):
    """C > 1024 (ResNeXt 2048, DenseNet-169 1664) and per-channel groups (ResNeXt split norm), with the
    ResNeXt tail y = relu(GN(x) + identity)."""
    import ops
    rng = np.random.default_rng(c)
    shape = (2, 5, 6, c)
    x = (rng.standard_normal(shape) * 1.5 + 0.3).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32)
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    xc, rc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(r).requires_grad_(True)
    gc, bc = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    yc = torch.relu(tf_ops_ref.group_norm(xc, gc, bc, groups) + rc)
    dy = rng.standard_normal(shape).astype(np.float32)
    yc.backward(torch.from_numpy(dy))
    xg, rg, gg, bg = _t(x, dev, True), _t(r, dev, True), _t(gamma, dev, True), _t(beta, dev, True)
    yg = ops.group_norm_act(xg, gg, bg, groups, 1e-5, "relu", rg, act_after_residual=True)
    yg.backward(_t(dy, dev))
    assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "gn fwd")
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "gn dx")
    assert_close(rg.grad.cpu().numpy(), rc.grad.numpy(), TOL, "gn dres")
    assert_close(gg.grad.cpu().numpy(), gc.grad.numpy(), TOL, "gn dgamma")
    assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), TOL, "gn dbeta")


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 7, 9, 8), (2, 25, 25, 32)])
def test_pools_and_dropout(dev, shape):
    import ops
    rng = np.random.default_rng(shape[1])
    x = rng.standard_normal(shape).astype(np.float32)
    for name, fn, ref, k, s in (("max", ops.max_pool, tf_ops_ref.max_pool_same, 3, 2),
                                ("avg", ops.avg_pool, tf_ops_ref.avg_pool_same, 2, 2)):
        xc = torch.from_numpy(x).requires_grad_(True)
        yc = ref(xc, k, s)
        dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
        yc.backward(torch.from_numpy(dy))
        xg = _t(x, dev, True)
        yg = fn(xg, k, s)
        yg.backward(_t(dy, dev))
        assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), 1e-6, name + " pool fwd")
        assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), 1e-6, name + " pool bwd")
        if name == "max":        # the other ABI form of the same gradient: re-scan the windows of x (no arg-max bytes)
            import _rn
            dx = torch.empty_like(xg)
            n, h, w, c = shape
            _rn.check(_rn.lib().rn_maxpool_bwd(_rn.f32(xg.detach()), _rn.f32(_t(dy, dev)), _rn.f32(dx), n, h, w, c, k, s,
                                               _rn.stream()), "rn_maxpool_bwd")
            assert torch.equal(dx, xg.grad)
    xg = _t(x, dev, True)
    y = ops.dropout(xg, 0.3, seed=77)
    keep = (y != 0)
    assert abs(keep.float().mean().item() - 0.7) < 0.03
    assert torch.allclose(y[keep], xg.detach()[keep] / 0.7, rtol=1e-6)
    y.sum().backward()
    assert torch.allclose(xg.grad, keep.float() / 0.7, rtol=1e-6)


def test_two_group_wide_conv_and_channel_split(dev):
    """Fused head towers: G=2 grouped conv with 256-wide groups (several N-tiles per group) and the
    output convs reading channel slices of one tensor == the separate convs."""
    import ops
    rng = np.random.default_rng(21)
    sizes = [(8, 8), (4, 4), (2, 2)]
    xs = [rng.standard_normal((2, h, w, 512)).astype(np.float32) for h, w in sizes]
    wa = (rng.standard_normal((3, 3, 256, 256)) / 48).astype(np.float32)
    wb = (rng.standard_normal((3, 3, 256, 256)) / 48).astype(np.float32)
    woa = (rng.standard_normal((3, 3, 256, 45)) / 48).astype(np.float32)
    wob = (rng.standard_normal((3, 3, 256, 36)) / 48).astype(np.float32)
    boa, bob = rng.standard_normal(45).astype(np.float32), rng.standard_normal(36).astype(np.float32)
    # oracle: separate dense convs on the two halves
    leaves = [torch.from_numpy(a).requires_grad_(True) for a in (wa, wb, woa, wob, boa, bob)]
    xcs = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    outs_c = []
    for x in xcs:
        ya = tf_ops_ref.conv2d_same(x[..., :256], leaves[0], 1)
        yb = tf_ops_ref.conv2d_same(x[..., 256:], leaves[1], 1)
        t = torch.cat([ya, yb], -1)
        outs_c += [tf_ops_ref.conv2d_same(t[..., :256], leaves[2], 1, leaves[4]),
                   tf_ops_ref.conv2d_same(t[..., 256:], leaves[3], 1, leaves[5])]
    dys = [rng.standard_normal(tuple(o.shape)).astype(np.float32) for o in outs_c]
    torch.autograd.backward(outs_c, [torch.from_numpy(d) for d in dys])
    g = [_t(a, dev, True) for a in (wa, wb, woa, wob, boa, bob)]
    xgs = [_t(x, dev, True) for x in xs]
    t = ops.conv2d(xgs, torch.cat([g[0], g[1]], 3), None, 1, groups=2)
    oa, ob = ops.conv2d_channel_split(t, [g[2], g[3]], [g[4], g[5]], 1)
    outs_g = [o for pair in zip(oa, ob) for o in pair]
    torch.autograd.backward(outs_g, [_t(d, dev) for d in dys])
    for a, b in zip(outs_g, outs_c):
        assert_close(a.detach().cpu().numpy(), b.detach().numpy(), TOL, "fused heads fwd")
    for a, b in zip(xgs, xcs):
        assert_close(a.grad.cpu().numpy(), b.grad.numpy(), TOL, "fused heads dx")
    for a, b, name in zip(g, leaves, ("wa", "wb", "woa", "wob", "boa", "bob")):
        assert_close(a.grad.cpu().numpy(), b.grad.numpy(), TOL, "fused heads d" + name)


# ---------------------------------------------------------------------------------- edge cases
def test_detect_empty_and_overflow(dev):
    import utils, _rn
    rng = np.random.default_rng(0)
    probs = {"P3": rng.uniform(0, 0.49, (2, 4, 4, 9, 7)).astype(np.float32)}
    boxes = {"P3": rng.uniform(0, 1, (2, 4, 4, 9, 4)).astype(np.float32)}
    out = utils.detect({k: _t(v, dev) for k, v in probs.items()}, {k: _t(v, dev) for k, v in boxes.items()}, 7)
    assert len(out) == 2 and all(o.boxes.shape == (0, 4) and o.scores.numel() == 0 for o in out)
    dec = utils.boxes_decode(_t(probs["P3"][0], dev), _t(boxes["P3"][0], dev))
    assert dec.boxes.shape == (0, 4) and utils.nms_classwise(dec, 7).boxes.shape[0] == 0
    probs["P3"][:] = 0.9                                          # every anchor is a candidate
    with pytest.raises(_rn.RnError):
        utils.detect({k: _t(v, dev) for k, v in probs.items()}, {k: _t(v, dev) for k, v in boxes.items()}, 7, capacity=10)


def test_loss_without_foreground(dev):
    """No anchor above 0.5 IoU: regr loss is 0 (SUM_BY_NONZERO_WEIGHTS, Q7), focal divides by max(#fg,1)."""
    import ops
    rng = np.random.default_rng(1)
    rows, c = 300, 6
    z = (rng.standard_normal((rows, c)) - 2).astype(np.float32)
    lab = np.zeros((rows, c), np.float32)
    rp, rl = rng.standard_normal((rows, 4)).astype(np.float32), rng.standard_normal((rows, 4)).astype(np.float32)
    m = rng.uniform(size=rows) < 0.7
    for mode in ("bce_dice", "focal"):
        zc = torch.from_numpy(z).requires_grad_(True)
        cl, rlc = losses_ref.loss(torch.from_numpy(lab)[m], torch.from_numpy(rl)[m], zc[m], torch.from_numpy(rp)[m], mode)
        cl.backward()
        zg, rg = _t(z, dev, True), _t(rp, dev, True)
        clg, rlg, _ = ops.detection_loss([zg], [rg], [_t(lab, dev)], [_t(rl, dev)], [_t(m.astype(np.uint8), dev)], c, mode)
        (clg + rlg).backward()
        assert rlg.item() == 0.0 and rlc.item() == 0.0
        assert_close(clg.item(), cl.item(), TOL, "class loss, no fg, " + mode)
        assert_close(zg.grad.cpu().numpy(), zc.grad.numpy(), TOL, "d logits, no fg")
        assert float(rg.grad.abs().max()) == 0.0


@pytest.mark.parametrize("c", [80, 3])
def test_loss_nan_targets_follow_the_references_masking(dev, c):
    """A degenerate ground-truth box leaves NaN log-size targets on every anchor (dataset.py:118-121, see
    test_assignment_degenerate_box_is_the_references_nan).  The reference REMOVES the rows outside the trainable mask
    (boolean_mask, utils.py:270-278) and MULTIPLIES the Huber term by the foreground weight (losses.py:146-152): NaN targets
    only on non-trainable rows -> finite loss and gradients; a NaN target on a trainable row (also a background one) -> NaN
    loss, NaN gradient on that row, as the oracle.  Both kernel families (four lanes per row: C = 80; wave per row: C = 3)."""
    import ops
    rng = np.random.default_rng(40 + c)
    z, lab, rp, rl, m = _loss_inputs(rng, c, levels=((2, 8, 8, 9),), fg_rate=0.1)[0]
    m = m.copy()
    m.reshape(-1)[:40] = False
    m.reshape(-1)[40:80] = True

    def run(rl_):
        zc, rc = torch.from_numpy(z).requires_grad_(True), torch.from_numpy(rp).requires_grad_(True)
        mt = torch.from_numpy(m)
        cl, rlc = losses_ref.loss(torch.from_numpy(lab)[mt], torch.from_numpy(rl_)[mt], zc[mt], rc[mt], "focal")
        (cl + rlc).backward()
        zg, rg = _t(z, dev, True), _t(rp, dev, True)
        clg, rlg, _ = ops.detection_loss([zg], [rg], [_t(lab, dev)], [_t(rl_, dev)], [_t(m.astype(np.uint8), dev)], c, "focal")
        (clg + rlg).backward()
        return rlc.item(), rc.grad.numpy(), rlg.item(), rg.grad.cpu().numpy(), clg.item(), cl.item()

    # (1) NaN only where the mask removes the row: everything finite and equal to the oracle
    a = rl.copy()
    a.reshape(-1, 4)[:40, 2] = np.nan
    ro, go, rk, gk, ck, co = run(a)
    assert np.isfinite(rk) and np.isfinite(gk).all()
    assert_close(rk, ro, TOL, "regr loss, NaN targets outside the mask")
    assert_close(gk, go, TOL, "d reg, NaN targets outside the mask")
    assert_close(ck, co, TOL, "class loss")
    # (2) NaN on trainable rows (foreground or not): NaN * weight = NaN in the loss and in those rows' gradients
    b = rl.copy()
    b.reshape(-1, 4)[40:80, 2] = np.nan
    ro, go, rk, gk, ck, co = run(b)
    assert np.isnan(ro) and np.isnan(rk)
    # (the loss is NaN either way; at the NaN elements themselves the kernel keeps NaN * weight = NaN, the torch oracle's
    # abs / clamp backward happens to give 0 there -- every other element must agree)
    ok = np.ones_like(gk, bool)
    ok.reshape(-1, 4)[40:80, 2] = False
    assert np.isnan(gk[~ok]).all() and np.isfinite(gk[ok]).all()
    assert_close(gk[ok], go[ok], TOL, "d reg away from the NaN targets")
    assert_close(ck, co, TOL, "class loss")


def test_assignment_single_object_and_padding(dev):
    """One valid object per image; the padded slots of the [N, max_obj] arrays must be ignored."""
    import dataset, levels
    lv = levels.build_levels()
    boxes = np.zeros((2, 8, 4), np.float32); cids = np.full((2, 8), 5, np.int32)
    boxes[0, 0] = [0.2, 0.2, 0.7, 0.8]; cids[0, 0] = 3
    boxes[1, 0] = [0.0, 0.0, 1.0, 1.0]; cids[1, 0] = 1
    boxes[:, 1:] = [0.4, 0.4, 0.6, 0.6]                           # garbage in the padding: must not be assigned
    nobj = np.array([1, 1], np.int32)
    for pn in ("P3", "P5", "P7"):
        f = 2 ** int(pn[-1])
        cls, reg, msk, arg = dataset.level_labels((128, 128), _t(cids, dev), _t(boxes, dev), lv[pn], f, 8,
                                                  num_obj=_t(nobj, dev), return_argmax=True)
        assert int(arg.max()) == 0
        for i in range(2):
            oc, orr, om, _ = dataset_ref.level_labels((128, 128), cids[i, :1], boxes[i, :1], lv[pn].anchor_sizes, f, 8)
            assert np.array_equal(cls[i].cpu().numpy(), oc) and np.array_equal(msk[i].cpu().numpy().astype(bool), om)


def test_tiny_and_ragged_shapes(dev):
    """1x1 maps, batch 1, odd sizes through conv / GroupNorm / upsample (P7 of small images, scale-600 sizes)."""
    import ops
    rng = np.random.default_rng(5)
    for (n, h, w) in ((1, 1, 1), (1, 1, 3), (3, 5, 1)):
        x = rng.standard_normal((n, h, w, 256)).astype(np.float32)
        wt = (rng.standard_normal((3, 3, 256, 256)) / 48).astype(np.float32)
        g, b = rng.standard_normal(256).astype(np.float32), rng.standard_normal(256).astype(np.float32)
        ref = tf_ops_ref.activation(tf_ops_ref.group_norm(tf_ops_ref.conv2d_same(torch.from_numpy(x), torch.from_numpy(wt), 1),
                                                          torch.from_numpy(g), torch.from_numpy(b)), "elu").numpy()
        got = ops.group_norm_act(ops.conv2d(_t(x, dev), _t(wt, dev)), _t(g, dev), _t(b, dev), 32, 1e-5, "elu")
        # hw == 1: 8 values per group, variance can be tiny -> rstd ~ 1/sqrt(eps): compare with matching slack
        assert_close(got.cpu().numpy(), ref, 5e-4, "tiny conv+gn %s" % ((n, h, w),))


def test_flip_augmentation_matches_reference_semantics(dev):
    """augmentation.flip on the device == the oracle (augmentation_test.py:7-45 vectors included), and the
    flipped labels equal labels built from the flipped boxes where the assignment is unambiguous."""
    import augmentation
    import reference_kats as K
    rng = np.random.default_rng(2)
    cls = {"P3": rng.uniform(size=(4, 6, 9, 5)).astype(np.float32), "P4": rng.uniform(size=(2, 3, 9, 5)).astype(np.float32)}
    reg = {"P3": rng.standard_normal((4, 6, 9, 4)).astype(np.float32), "P4": K.FLIP_REGR_INPUT.repeat(9, 2).reshape(2, 3, 9, 4)}
    msk = {"P3": (rng.uniform(size=(4, 6, 9)) < 0.5), "P4": (rng.uniform(size=(2, 3, 9)) < 0.5)}
    img = rng.standard_normal((32, 48, 3)).astype(np.float32)
    sample = {"image": _t(img, dev), "detection": {"classifications": {k: _t(v, dev) for k, v in cls.items()},
                                                   "regressions": {k: _t(v, dev) for k, v in reg.items()}},
              "trainable_masks": {k: _t(v.astype(np.uint8), dev) for k, v in msk.items()}}
    got = augmentation.flip(sample)
    ec, er, em, ei = dataset_ref.flip(cls, reg, msk, img)
    assert np.array_equal(got["image"].cpu().numpy(), ei)
    for k in cls:
        assert np.array_equal(got["detection"]["classifications"][k].cpu().numpy(), ec[k])
        assert np.array_equal(got["detection"]["regressions"][k].cpu().numpy(), er[k])
        assert np.array_equal(got["trainable_masks"][k].cpu().numpy().astype(bool), em[k])
    pair = augmentation.make_pair(sample)
    assert pair["image"].shape == (2, 32, 48, 3) and pair["detection"]["regressions"]["P3"].shape == (2, 4, 6, 9, 4)
    assert torch.equal(pair["image"][1], got["image"])


def test_checkpoint_roundtrip(dev, tmp_path):
    import checkpoint, layers, levels, retinanet, train
    lv = levels.build_levels()
    torch.manual_seed(0)
    net = retinanet.RetinaNet('mobilenet_v2', lv, 3, layers.elu, 0.0).to(dev)
    tr = train.Trainer(net, lv, optimizer="rmsprop", device=dev)
    tr.opt.state2.normal_(); tr.opt.step_count = 7
    tr.drop_counter.fill_(424242)
    ref = {k: v.detach().clone() for k, v in net.named_parameters()}
    s1, s2 = tr.opt.state1.clone(), tr.opt.state2.clone()
    path = str(tmp_path / "ckpt" / "model.safetensors")
    checkpoint.save(path, net, tr, step=123)
    with torch.no_grad():
        tr.arena.weights.zero_(); tr.opt.state1.zero_(); tr.opt.state2.zero_(); tr.drop_counter.zero_()
    assert checkpoint.load(path, net, tr) == 123 and tr.opt.step_count == 7 and int(tr.drop_counter.item()) == 424242
    for k, v in net.named_parameters():
        assert torch.equal(v, ref[k]) and v.data_ptr() >= tr.arena.weights.data_ptr()      # still views of the arena
    for off, size in tr.arena.offsets:       # optimizer slots are stored per parameter (the padding between them is not state)
        assert torch.equal(tr.opt.state1[off:off + size], s1[off:off + size]) and torch.equal(tr.opt.state2[off:off + size], s2[off:off + size])
    # a model the file does not fit says which tensor
    other = retinanet.RetinaNet('mobilenet_v2', lv, 5, layers.elu, 0.0).to(dev)
    with pytest.raises(ValueError, match="out_conv"):
        checkpoint.load(path, other)


def test_batched_gemm_both_layouts(dev):
    """rn_gemm_batched (the Winograd product stage): C_b = A_b B_b with B as [K,N] and as [N,K]; ragged M."""
    import _rn
    L = _rn.lib()
    rng = np.random.default_rng(11)
    for (nb, m, k, n) in [(36, 75, 256, 64), (16, 130, 64, 256), (3, 1, 32, 36)]:
        a = rng.standard_normal((nb, m, k)).astype(np.float32)
        b = rng.standard_normal((nb, k, n)).astype(np.float32)
        ref = np.einsum("bmk,bkn->bmn", a.astype(np.float64), b.astype(np.float64))
        ag, bg = _t(a, dev), _t(b, dev)
        bt = _t(np.ascontiguousarray(b.transpose(0, 2, 1)), dev)
        for b_nk, bb in ((0, bg), (1, bt)):
            c = torch.empty((nb, m, n), dtype=torch.float32, device=dev)
            _rn.check(L.rn_gemm_batched(_rn.f32(ag), _rn.f32(bb), _rn.f32(c), m, k, n, nb, b_nk, _rn.stream()), "gemm")
            assert_close(c.cpu().numpy(), ref, 1e-5, "batched gemm b_nk=%d" % b_nk)


def test_deferred_reductions_are_bitwise_identical(dev):
    """Deferred reductions (rn_reduce_list + rn_flush_reductions): the gradients after the single flushed launch equal the immediate ones bit for bit."""
    import ops
    rng = np.random.default_rng(5)
    x = _t(rng.standard_normal((2, 16, 16, 96)).astype(np.float32), dev)
    convw = torch.nn.Parameter(_t(rng.standard_normal((1, 1, 96, 24)).astype(np.float32) / 10, dev))
    bias = torch.nn.Parameter(_t(rng.standard_normal(24).astype(np.float32), dev))
    dww = torch.nn.Parameter(_t(rng.standard_normal((3, 3, 96, 1)).astype(np.float32) / 3, dev))
    gamma = torch.nn.Parameter(_t(1 + 0.1 * rng.standard_normal(96).astype(np.float32), dev))
    beta = torch.nn.Parameter(_t(0.1 * rng.standard_normal(96).astype(np.float32), dev))
    params = [convw, bias, dww, gamma, beta]

    def run(defer):
        for p in params:
            p.grad = torch.zeros_like(p)
        old = ops.DIRECT_PARAM_GRADS
        ops.DIRECT_PARAM_GRADS = True
        ops.begin_direct_grad_step()
        try:
            h = ops.depthwise_conv2d(x, dww, 1)
            h = ops.group_norm_act(h, gamma, beta, groups=32, act="relu6")
            y = ops.conv2d(h, convw, bias, 1)
            if defer:
                ops.begin_deferred_reductions()
            y.square().sum().backward()
            if defer:
                ops.end_deferred_reductions()
        finally:
            ops.DIRECT_PARAM_GRADS = old
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    immediate, deferred = run(False), run(True)
    for a, b in zip(immediate, deferred):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 4, 4, 256, 256, 3, 1), (2, 16, 16, 320, 256, 3, 2), (1, 8, 8, 256, 64, 3, 2), (2, 2, 2, 1280, 256, 1, 1)])
def test_split_k_tiny_grids(dev, case):
    """Tiny output grids with a long reduction (P6 / P7 convs) take the split-K path of rn_conv2d_fwd / dgrad."""
    import ops
    import _rn
    n, h, w, cin, cout, k, stride = case
    rng = np.random.default_rng(cin + cout + h)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xc, wc, bc = (torch.from_numpy(a).requires_grad_(True) for a in (x, wt, b))
    yc = tf_ops_ref.conv2d_same(xc, wc, stride, bc)
    dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    yc.backward(torch.from_numpy(dy))
    old = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        xg, wg, bg = _t(x, dev, True), _t(wt, dev, True), _t(b, dev, True)
        segs = ops._conv_segs([xg.detach()], wg.detach(), None, [torch.empty_like(_t(dy, dev))], None, None)
        geom = _rn.ConvGeom(k, k, stride, cin, 1)
        assert _rn.lib().rn_conv2d_fwd_workspace(segs, 1, __import__("ctypes").byref(geom)) > 0, "expected a split-K plan"
        yg = ops.conv2d(xg, wg, bg, stride)
        yg.backward(_t(dy, dev))
    finally:
        ops.WINOGRAD = old
    assert_close(yg.detach().cpu().numpy(), yc.detach().numpy(), TOL, "split-K fwd")
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), TOL, "split-K dgrad")
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), TOL, "wgrad")


@pytest.mark.parametrize("case", [((2, 37, 53, 3), (64, 91), "u8"), ((1, 120, 80, 3), (48, 32), "f32"), ((1, 5, 7, 3), (1, 1), "f32"),
                                  ((2, 16, 16, 4), (16, 16), "u8"), ((1, 3, 2, 1), (9, 11), "f32")])
def test_resize_bilinear_normalize_bit_exact(dev, case):
    """rescale_image (align_corners bilinear) + convert_image_dtype + preprocess_image == the oracle, bit for bit."""
    import dataset
    shape, (oh, ow), kind = case
    rng = np.random.default_rng(shape[1] * 7 + oh)
    x = rng.integers(0, 256, shape).astype(np.uint8) if kind == "u8" else rng.random(shape).astype(np.float32)
    ref = dataset_ref.resize_bilinear_align_corners(x, oh, ow)
    got = dataset.rescale_image(_t(x, dev), size=(oh, ow))
    assert np.array_equal(got.cpu().numpy(), ref)
    if shape[3] == 3:
        got = dataset.rescale_image(_t(x, dev), size=(oh, ow), normalize=True)
        assert np.array_equal(got.cpu().numpy(), dataset_ref.preprocess_image(ref))
    assert dataset.rescale_size((480, 640), 256) == dataset_ref.rescale_size((480, 640), 256) == (256, 341)
    assert dataset.rescale_size((500, 375), 300) == (400, 300)          # 1.3333334 * 300 = 400.00003 -> 400


def test_build_dataset_pipeline(dev):
    """data_loaders.shapes -> rescale -> labels -> [sample, hflip] -> normalise, all on the device; labels of the
    flipped half equal labels built from flipped boxes."""
    import dataset, levels as levels_mod
    from data_loaders.shapes import Shapes
    lv = levels_mod.build_levels()
    loader = Shapes(None, num_samples=2, image_size=(160, 128), seed=3)
    assert loader.num_classes == 3 and loader.class_names == ['square', 'triangle', 'circle']
    batches = list(dataset.build_dataset(loader, lv, scale=96, device=dev))
    assert len(batches) == 2
    b = batches[0]
    assert tuple(b['image'].shape) == (2, 120, 96, 3) and b['image'].dtype == torch.float32
    for k in lv:
        assert b['detection']['classifications'][k].shape[0] == 2
        assert b['trainable_masks'][k].dtype in (torch.bool, torch.uint8)
    # the second half is the h-flip of the first
    assert torch.equal(b['image'][1], torch.flip(b['image'][0], [1]))
    flipped_boxes = dataset.flip_boxes(torch.from_numpy(b['boxes']).to(dev))[None]
    ids = torch.from_numpy(np.asarray(b['class_ids'], np.int32)).to(dev)[None]
    c2, r2, m2 = dataset.build_labels(b['image_size'], ids, flipped_boxes, lv, 3)
    for k in lv:
        assert torch.equal(b['trainable_masks'][k][1], m2[k][0])
        assert torch.equal(b['detection']['classifications'][k][1], c2[k][0])


@pytest.mark.parametrize("mode", ["default", "rows", "grid_resident", "three_kernel"])
@pytest.mark.parametrize("c,act,res", [(96, "relu6", False), (256, "elu", True), (24, None, False), (144, "elu", False)])
def test_group_norm_all_kernel_paths(dev, monkeypatch, mode, c, act, res):
    """The GroupNorm implementations (slice-resident; partial-sum rows + merging apply; grid-resident with the in-kernel
    barrier; partial + finalize + apply) on mid-sized multi-segment inputs, each against the oracle; no barrier may time out."""
    import _rn
    import ops
    if mode != "default":
        monkeypatch.setenv("RN_GN_NO_SLICE", "1")
    if mode in ("grid_resident", "three_kernel"):
        monkeypatch.setenv("RN_GN_NO_ROWS", "1")
    if mode == "three_kernel":
        monkeypatch.setenv("RN_GN_NO_COOP", "1")
    rng = np.random.default_rng(c + len(mode))
    shapes = [(2, 40, 36, c), (2, 20, 18, c), (2, 3, 3, c)]
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    xs = [(rng.standard_normal(s) * 2 + 0.5).astype(np.float32) for s in shapes]
    rs = [rng.standard_normal(s).astype(np.float32) for s in shapes] if res else None
    gc, bc = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    xcs = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    rcs = [torch.from_numpy(r).requires_grad_(True) for r in rs] if res else None
    ycs = []
    for i, x in enumerate(xcs):
        y = tf_ops_ref.activation(tf_ops_ref.group_norm(x, gc, bc), act)
        ycs.append(y + rcs[i] if res else y)
    dys = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    torch.autograd.backward(ycs, [torch.from_numpy(d) for d in dys])
    gg, bg = _t(gamma, dev, True), _t(beta, dev, True)
    xgs = [_t(x, dev, True) for x in xs]
    rgs = [_t(r, dev, True) for r in rs] if res else None
    ygs = ops.group_norm_act(xgs, gg, bg, 32, 1e-5, act, rgs)
    torch.autograd.backward(ygs, [_t(d, dev) for d in dys])
    for i in range(len(xs)):
        assert_close(ygs[i].detach().cpu().numpy(), ycs[i].detach().numpy(), TOL, "gn fwd " + mode)
        assert_close(xgs[i].grad.cpu().numpy(), xcs[i].grad.numpy(), TOL, "gn dx " + mode)
        if res:
            assert_close(rgs[i].grad.cpu().numpy(), rcs[i].grad.numpy(), TOL, "gn dres " + mode)
    assert_close(gg.grad.cpu().numpy(), gc.grad.numpy(), TOL, "gn dgamma " + mode)
    assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), TOL, "gn dbeta " + mode)
    assert _rn.barrier_timeouts() == 0


@pytest.mark.parametrize("case", [(2, 32, 32, 96, 24, 1, 1), (2, 16, 16, 144, 32, 1, 1), (1, 20, 20, 64, 128, 3, 1), (2, 8, 8, 320, 1280, 1, 1)])
def test_merged_conv_backward_equals_separate_calls(dev, case):
    """rn_conv2d_bwd (data + weight gradient in one launch) gives the same bits as rn_conv2d_dgrad + rn_conv2d_wgrad."""
    import ctypes as C
    import _rn
    import ops
    n, h, w, cin, cout, k, stride = case
    g = torch.Generator(device=dev).manual_seed(cin + cout)
    x = torch.randn((n, h, w, cin), generator=g, device=dev)
    wt = torch.randn((k, k, cin, cout), generator=g, device=dev) * 0.05
    dy = torch.randn((n, h, w, cout), generator=g, device=dev)
    L = _rn.lib()
    geom = _rn.ConvGeom(k, k, stride, cin, 1)

    def run(merged):
        dx, dw = torch.empty_like(x), torch.empty_like(wt)
        segs = ops._conv_segs([x], wt, None, None, [dy], [dx])
        ws = torch.empty(max(int(L.rn_conv2d_wgrad_workspace(segs, 1, C.byref(geom))), 256), dtype=torch.uint8, device=dev)
        if merged:
            _rn.check(L.rn_conv2d_bwd(segs, 1, C.byref(geom), _rn.f32(dw), ws.data_ptr(), ws.numel(), _rn.stream(), None), "bwd")
        else:
            _rn.check(L.rn_conv2d_dgrad(segs, 1, C.byref(geom), None, 0, _rn.stream()), "dgrad")
            _rn.check(L.rn_conv2d_wgrad(segs, 1, C.byref(geom), _rn.f32(dw), 0, ws.data_ptr(), ws.numel(), _rn.stream(), None), "wgrad")
        torch.cuda.synchronize()
        return dx, dw

    (dx1, dw1), (dx2, dw2) = run(True), run(False)
    assert float(dx1.abs().max()) > 0 and float(dw1.abs().max()) > 0
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2)


@pytest.mark.parametrize("tile", [4, 2])
@pytest.mark.parametrize("keep", [True, False])
def test_winograd_merged_backward_equals_separate_calls(dev, tile, keep):
    """rn_conv3x3_winograd_bwd (three launches of two-kind blocks) == rn_conv3x3_winograd(dgrad) + rn_conv3x3_winograd_wgrad,
    bit for bit, with and without the buffers a forward call keeps; five pyramid levels in one call."""
    import ctypes as C
    import _rn
    import ops
    g = torch.Generator(device=dev).manual_seed(tile + keep)
    cin, cout = 64, 128
    sizes = [(12, 12), (6, 6), (3, 3), (2, 2), (1, 1)]
    xs = [torch.randn((2, h, w, cin), generator=g, device=dev) for h, w in sizes]
    dys = [torch.randn((2, h, w, cout), generator=g, device=dev) for h, w in sizes]
    wt = torch.randn((3, 3, cin, cout), generator=g, device=dev) * 0.05
    L = _rn.lib()
    n = len(xs)
    v = u = None
    if keep:   # what a training-mode forward call leaves behind
        ys = [torch.empty((2, h, w, cout), device=dev) for h, w in sizes]
        fs = ops._conv_segs(xs, wt, None, ys, None, None)
        vb, ub = C.c_size_t(0), C.c_size_t(0)
        _rn.check(L.rn_conv3x3_winograd_keep_bytes(fs, n, cin, cout, tile, C.byref(vb), C.byref(ub)), "keep")
        v = torch.empty(vb.value // 4, device=dev)
        u = torch.empty(ub.value // 4, device=dev)
        ws = torch.empty(L.rn_conv3x3_winograd_workspace(fs, n, cin, cout, tile), dtype=torch.uint8, device=dev)
        _rn.check(L.rn_conv3x3_winograd(fs, n, cin, cout, _rn.f32(wt), None, 0, tile, ws.data_ptr(), ws.numel(), _rn.f32(v),
                                        _rn.f32(u), _rn.stream()), "fwd")

    def separate():
        dxs, dw = [torch.empty_like(x) for x in xs], torch.empty_like(wt)
        segs = ops._conv_segs(xs, wt, None, None, dys, dxs)
        ws = torch.empty(L.rn_conv3x3_winograd_workspace(segs, n, cin, cout, tile), dtype=torch.uint8, device=dev)
        _rn.check(L.rn_conv3x3_winograd(segs, n, cin, cout, _rn.f32(wt), None, 1, tile, ws.data_ptr(), ws.numel(), None,
                                        _rn.f32(u) if keep else None, _rn.stream()), "dgrad")
        ws = torch.empty(L.rn_conv3x3_winograd_wgrad_workspace(segs, n, cin, cout, tile), dtype=torch.uint8, device=dev)
        _rn.check(L.rn_conv3x3_winograd_wgrad(segs, n, cin, cout, _rn.f32(dw), 0, tile, ws.data_ptr(), ws.numel(),
                                              _rn.f32(v) if keep else None, _rn.stream()), "wgrad")
        return dxs, dw

    def merged():
        dxs, dw = [torch.empty_like(x) for x in xs], torch.empty_like(wt)
        segs = ops._conv_segs(xs, wt, None, None, dys, dxs)
        need = L.rn_conv3x3_winograd_bwd_workspace(segs, n, cin, cout, tile, 1 if keep else 0, 1 if keep else 0)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _rn.check(L.rn_conv3x3_winograd_bwd(segs, n, cin, cout, _rn.f32(wt), _rn.f32(dw), 0, tile, ws.data_ptr(), ws.numel(),
                                            _rn.f32(v) if keep else None, _rn.f32(u) if keep else None, _rn.stream()), "bwd")
        return dxs, dw

    (dx1, dw1), (dx2, dw2) = separate(), merged()
    torch.cuda.synchronize()
    assert float(dw1.abs().max()) > 0
    for a, b in zip(dx1, dx2):
        assert torch.equal(a, b)
    assert torch.equal(dw1, dw2)


def test_winograd_forward_keeps_buffers_only_when_training(dev):
    """A training-mode conv2d on the Winograd path leaves the transformed input / rotated kernel on its backward node
    (so the backward pass skips both transforms); under no_grad nothing is kept; gradients agree to rounding either way."""
    import ops
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn((2, 10, 10, 64), generator=g, device=dev, requires_grad=True)
    w = (torch.randn((3, 3, 64, 64), generator=g, device=dev) * 0.05).requires_grad_(True)
    assert ops.WINOGRAD and ops.WINOGRAD_KEEP
    y = ops.conv2d(x, w, None, 1)
    assert y.grad_fn.wino_v is not None and y.grad_fn.wino_urot is not None
    dy = torch.randn(y.shape, generator=g, device=dev)
    dx1, dw1 = torch.autograd.grad(y, (x, w), dy)
    ops.WINOGRAD_KEEP = False
    try:
        y2 = ops.conv2d(x, w, None, 1)
        assert y2.grad_fn.wino_v is None and y2.grad_fn.wino_urot is None
        dx2, dw2 = torch.autograd.grad(y2, (x, w), dy)
    finally:
        ops.WINOGRAD_KEEP = True
    assert torch.equal(y, y2)
    for name, p_, q_ in (("dx", dx1, dx2), ("dw", dw1, dw2)):
        print(name, "kept vs recomputed: max abs diff", float((p_ - q_).abs().max()), "of", float(q_.abs().max()))
        # the kept rotated kernel comes out of the forward's two-output transform, the recomputed one out of the stand-alone
        # kernel: same formula, different fma contraction => a few ulp (observed 2e-6 of the range)
        assert_close(p_.cpu().numpy(), q_.cpu().numpy(), 1e-5, name + " kept vs recomputed transforms")
    xf = x.detach().requires_grad_(True)
    y3 = ops.conv2d(xf, w.detach(), None, 1)      # only the data gradient is wanted: no transformed input kept
    assert y3.grad_fn.wino_v is None and y3.grad_fn.wino_urot is not None
    with torch.no_grad():
        assert torch.equal(ops.conv2d(x, w, None, 1), y)


def test_detect_fp16_maps_and_logits_in_the_scan(dev):
    """BASELINE configs[4]: the class / box maps stay fp16 (2 bytes per element through the scan) and the sigmoid runs
    inside the scan.  fp16 storage == the same values handed over as fp32, bit for bit; logits + fused sigmoid == sigmoid
    kernel then scan, bit for bit; and the result equals the oracle on those probabilities."""
    import levels, ops, utils
    rng = np.random.default_rng(31)
    n, c = 2, 80
    lv = levels.build_levels()
    size = (128, 160)
    logits, regs, anchors = {}, {}, {}
    for k in lv:
        f = 2 ** int(k[1])
        gh, gw = -(-size[0] // f), -(-size[1] // f)
        logits[k] = (rng.standard_normal((n, gh, gw, 9, c)) * 2 - 3.5).astype(np.float16)      # ~1 % of the anchors above 0
        regs[k] = (rng.standard_normal((n, gh, gw, 9, 4)) * 0.3).astype(np.float16)
        anchors[k] = lv[k].normalized_anchor_sizes(size)
    l16 = {k: _t(v, dev) for k, v in logits.items()}
    r16 = {k: _t(v, dev) for k, v in regs.items()}
    l32 = {k: v.float() for k, v in l16.items()}
    r32 = {k: v.float() for k, v in r16.items()}
    p32 = {k: ops.activation(v, 'sigmoid') for k, v in l32.items()}                 # the stand-alone sigmoid kernel
    want = utils.detect_raw(p32, r32, anchors, c)
    for got in (utils.detect_raw(l16, r16, anchors, c, logits=True),                 # fp16 storage, sigmoid in the scan
                utils.detect_raw(l32, r32, anchors, c, logits=True),                 # fp32 storage, sigmoid in the scan
                utils.detect_raw(l16, r32, anchors, c, logits=True)):
        for i in range(n):
            assert torch.equal(got[i].boxes, want[i].boxes) and torch.equal(got[i].scores, want[i].scores)
            assert torch.equal(got[i].class_ids, want[i].class_ids)
    p16 = {k: v.half() for k, v in p32.items()}                                     # fp16 probabilities (no logits)
    a = utils.detect_raw(p16, r16, anchors, c)
    b = utils.detect_raw({k: v.float() for k, v in p16.items()}, r32, anchors, c)
    kept = 0
    for i in range(n):
        assert torch.equal(a[i].boxes, b[i].boxes) and torch.equal(a[i].scores, b[i].scores) and torch.equal(a[i].class_ids, b[i].class_ids)
        dec = {k: utils.regression_postprocess(r32[k], anchors[k])[i].cpu().numpy() for k in lv}
        parts = [utils_ref.boxes_decode(p32[k][i].cpu().numpy(), dec[k]) for k in lv]
        exp = utils_ref.nms_classwise(utils_ref.merge_boxes_decoded(parts), c)
        assert np.array_equal(want[i].boxes.cpu().numpy(), exp.boxes) and np.array_equal(want[i].class_ids.cpu().numpy(), exp.class_ids)
        assert np.array_equal(want[i].scores.cpu().numpy(), exp.scores)
        kept += len(exp.scores)
    assert kept > 20
    # adversarial rows for the max-logit shortcut of the scan: saturated logits (sigmoid == 1.0f for several classes: the
    # FIRST of them must win, not the largest logit), equal logits, logits one ulp apart, huge negative rows
    z = (rng.standard_normal((1, 8, 8, 9, c)) * 2 - 3.5).astype(np.float32)
    rows = z.reshape(-1, c)
    rows[0, [7, 30, 60]] = [18.0, 25.0, 40.0]
    rows[1, [5, 6]] = [2.5, 2.5]
    rows[2, [40, 3]] = [1.0, np.nextafter(np.float32(1.0), np.float32(2.0))]
    rows[3, [70, 11]] = [np.nextafter(np.float32(9.0), np.float32(10.0)), 9.0]
    rows[4, :] = -80.0
    rows[5, [9, 8]] = [16.7, 16.6]
    rows[6, [1, 0]] = [12.0, np.nextafter(np.float32(12.0), np.float32(0.0))]
    adv = {"P5": _t(z, dev)}
    radv = {"P5": _t((rng.standard_normal((1, 8, 8, 9, 4)) * 0.3).astype(np.float32), dev)}
    aadv = {"P5": lv["P5"].normalized_anchor_sizes((256, 256))}
    w1 = utils.detect_raw({"P5": ops.activation(adv["P5"], 'sigmoid')}, radv, aadv, c)[0]
    g1 = utils.detect_raw(adv, radv, aadv, c, logits=True)[0]
    assert torch.equal(g1.boxes, w1.boxes) and torch.equal(g1.scores, w1.scores) and torch.equal(g1.class_ids, w1.class_ids)
    assert 7 in g1.class_ids.tolist() and 5 in g1.class_ids.tolist()
    # a class count that is not a multiple of 8 takes the generic fp16 path
    c2 = 5
    lg = {k: _t((rng.standard_normal((1, 4, 4, 9, c2)) * 2 - 1).astype(np.float16), dev) for k in ("P3",)}
    rg = {k: _t((rng.standard_normal((1, 4, 4, 9, 4)) * 0.3).astype(np.float16), dev) for k in ("P3",)}
    an = {"P3": lv["P3"].normalized_anchor_sizes((32, 32))}
    g1 = utils.detect_raw(lg, rg, an, c2, logits=True)[0]
    g2 = utils.detect_raw({k: ops.activation(v.float(), 'sigmoid') for k, v in lg.items()}, {k: v.float() for k, v in rg.items()}, an, c2)[0]
    assert torch.equal(g1.boxes, g2.boxes) and torch.equal(g1.scores, g2.scores) and len(g1.scores) > 0


def test_segment_sort_long_segment_and_ties(dev):
    """The hand-written segment sort: a single segment longer than its LDS capacity (8192 keys: sorted in place in global
    memory), many equal scores (ties resolve to the lower index), negative scores -- utils.nms vs the oracle."""
    import utils
    rng = np.random.default_rng(41)
    for k, distinct in ((20000, 50), (9000, 0), (300, 3)):
        c = rng.uniform(0.05, 0.95, (k, 2))
        s_ = rng.uniform(0.01, 0.05, (k, 2))
        boxes = np.concatenate([c - s_ / 2, c + s_ / 2], 1).astype(np.float32)
        scores = (rng.integers(0, distinct, k) / max(distinct, 1) - 0.3).astype(np.float32) if distinct else rng.standard_normal(k).astype(np.float32)
        dec = utils.BoxesDecoded(boxes=_t(boxes, dev), scores=_t(scores, dev), class_ids=torch.zeros(k, dtype=torch.int64, device=dev))
        got = utils.nms(dec)
        idx = utils_ref.nms_indices_vectorised(boxes, scores)
        assert np.array_equal(got.boxes.cpu().numpy(), boxes[idx]) and np.array_equal(got.scores.cpu().numpy(), scores[idx])


DWGN_CASES = [  # n, h, w, c, stride, act, drop rate   (c / groups = 6, 12, 18, 30: the widths of MobileNetV2's bottlenecks)
    (2, 32, 32, 192, 1, "elu", 0.0),
    (2, 32, 32, 384, 1, "elu", 0.2),
    (2, 32, 32, 576, 1, "relu6", 0.2),       # input + output slice: the largest that still fits the backward kernel
    (2, 32, 32, 576, 2, "elu", 0.1),
    (2, 16, 16, 960, 1, "elu", 0.2),
    (3, 15, 13, 96, 2, "relu", 0.0),         # odd sizes, stride 2 (asymmetric SAME padding)
    (1, 64, 64, 192, 2, "elu", 0.2),         # the 64 x 64 input slice (98 KB) with a 32 x 32 output
    (2, 5, 7, 32, 1, None, 0.0),             # one channel per group
]


@pytest.mark.parametrize("case", DWGN_CASES, ids=[str(i) for i in range(len(DWGN_CASES))])
def test_fused_groupnorm_depthwise_groupnorm(dev, monkeypatch, case):
    """rn_dwgn_fwd / rn_dwgn_bwd (one kernel per direction) == GroupNorm kernel + depthwise kernel + GroupNorm kernel, forward and
    every gradient, with the same dropout masks; and == the oracle when there is no dropout."""
    import ops
    monkeypatch.setattr(ops, "DW_GN_FUSED", True)
    n, h, w, c, stride, act, rate = case
    rng = np.random.default_rng(sum(case[:5]))
    g = ops.gn_groups(c, 32)
    x = _t((rng.standard_normal((n, h, w, c)) * 1.5 + 0.3).astype(np.float32), dev, True)
    prm = [_t(v.astype(np.float32), dev, True) for v in (1 + 0.3 * rng.standard_normal(c), 0.2 * rng.standard_normal(c),
                                                             rng.standard_normal((3, 3, c, 1)) / 3,
                                                             1 + 0.3 * rng.standard_normal(c), 0.2 * rng.standard_normal(c))]
    g1, b1, wd, g2, b2 = prm
    assert ops.dw_gn_ok((n, h, w, c), wd, stride, g, act, True)
    counter = torch.tensor([12345], dtype=torch.int64, device=dev)
    got = ops.dw_gn_fused(x, g1, b1, wd, g2, b2, stride, g, 1e-5, act, rate, 111, 222, counter)
    a1 = ops.group_norm_act(x, g1, b1, groups=32, eps=1e-5, act=act, drop_rate=rate, seed=111, seed_dev=counter)
    ref = ops.group_norm_act(ops.depthwise_conv2d(a1, wd, stride), g2, b2, groups=32, eps=1e-5, act=act, drop_rate=rate, seed=222,
                             seed_dev=counter)
    assert got.shape == ref.shape
    assert_close(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), 2e-5, "fused forward")
    if rate > 0:      # the same elements are dropped
        assert torch.equal(got == 0, ref == 0) or float(((got == 0) != (ref == 0)).float().mean()) < 1e-5
    dy = _t(rng.standard_normal(tuple(ref.shape)).astype(np.float32), dev)
    leaves = [x] + prm
    gg = torch.autograd.grad(got, leaves, dy)
    gr = torch.autograd.grad(ref, leaves, dy)
    for name, a, b in zip(("dx", "dgamma1", "dbeta1", "dw", "dgamma2", "dbeta2"), gg, gr):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-4, name)
    if rate == 0:
        xc = x.detach().cpu().requires_grad_(True)
        pc = [t.detach().cpu().requires_grad_(True) for t in prm]
        o = tf_ops_ref.activation(tf_ops_ref.group_norm(xc, pc[0], pc[1], 32), act or "none")
        o = tf_ops_ref.activation(tf_ops_ref.group_norm(tf_ops_ref.depthwise_conv2d_same(o, pc[2], stride), pc[3], pc[4], 32), act or "none")
        assert_close(got.detach().cpu().numpy(), o.detach().numpy(), 1e-4, "fused forward vs oracle")
        go = torch.autograd.grad(o, [xc] + pc, dy.cpu())
        for name, a, b in zip(("dx", "dgamma1", "dbeta1", "dw", "dgamma2", "dbeta2"), gg, go):
            assert_close(a.cpu().numpy(), b.numpy(), 1e-4, name + " vs oracle")


def test_fused_dwgn_declines_what_does_not_fit(dev, monkeypatch):
    import ops
    monkeypatch.setattr(ops, "DW_GN_FUSED", True)
    wd = torch.zeros((3, 3, 192, 1), device=dev)
    assert not ops.dw_gn_ok((2, 128, 128, 192), wd, 1, 32, "elu", False)        # 393 KB slice
    assert ops.dw_gn_ok((2, 64, 64, 192), wd, 1, 32, "elu", False) and not ops.dw_gn_ok((2, 64, 64, 192), wd, 1, 32, "elu", True)


# ------------------------------------------------------------------ GroupNorm statistics from the producing kernel
def _moments_ref(y, g, eps):
    n, h, w, c = y.shape
    v = y.detach().double().cpu().reshape(n, h * w, g, c // g)
    mean = v.mean(dim=(1, 3))
    var = v.var(dim=(1, 3), unbiased=False)
    return mean.numpy(), (1.0 / torch.sqrt(var + eps)).numpy()


PRODUCER_STATS_CASES = [
    # kind, x shape, cout, k, stride, act, residual, drop   (MobileNetV2-FPN shapes of BASELINE configs[1], and odd ones)
    ("conv", (2, 32, 32, 64), 384, 1, 1, "elu", False, 0.2),      # expand conv at 1/16 (tile 64x64)
    ("conv", (2, 64, 64, 24), 144, 1, 1, "elu", False, 0.0),      # C = 144: 24 groups of 6, groups straddle tiles
    ("conv", (2, 64, 64, 144), 32, 1, 1, None, False, 0.2),       # linear bottleneck conv, per-channel groups
    ("conv", (2, 64, 64, 192), 32, 1, 1, None, True, 0.0),        # ... with the residual added by the GroupNorm
    ("conv", (2, 128, 128, 3), 32, 3, 2, "relu6", False, 0.0),    # stem: scalar (cin = 3) kernel, 128 x 32 tiles
    ("conv", (3, 16, 16, 160), 960, 1, 1, "relu", False, 0.0),    # three samples, 960 channels
    ("dw", (2, 32, 32, 384), 384, 3, 1, "elu", False, 0.2),
    ("dw", (2, 64, 64, 144), 144, 3, 2, "elu", False, 0.0),       # stride 2, C = 144
    ("dw", (2, 33, 29, 96), 96, 3, 1, "relu6", False, 0.0),       # odd map: ragged last chunk
    ("dw", (1, 128, 128, 32), 32, 3, 1, "elu", False, 0.0),       # 64 pixel lanes per block
]


@pytest.mark.parametrize("case", PRODUCER_STATS_CASES, ids=lambda c: "%s-%s-%d" % (c[0], "x".join(map(str, c[1])), c[2]))
def test_group_norm_statistics_from_the_producer(dev, monkeypatch, case):
    """conv / depthwise forward that also emits partial-sum rows for the GroupNorm that follows (rn_conv2d_fwd_stats,
    rn_depthwise_fwd_stats) + the rows-merging GroupNorm == conv, then the stand-alone GroupNorm (normalization.py:20-35):
    same conv output bits, moments vs an fp64 restatement, same output, dropout mask and gradients."""
    import ops
    kind, xs, cout, k, stride, act, use_res, rate = case
    rng = np.random.default_rng(17)
    n, h, w, cin = xs
    x = _t((rng.standard_normal(xs) * 1.5 + 0.3).astype(np.float32), dev, True)
    if kind == "conv":
        wt = _t((rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32), dev, True)
    else:
        wt = _t((rng.standard_normal((3, 3, cin, 1)) / 3).astype(np.float32), dev, True)
    gamma = _t((1 + 0.2 * rng.standard_normal(cout)).astype(np.float32), dev, True)
    beta = _t((0.1 * rng.standard_normal(cout)).astype(np.float32), dev, True)
    counter = torch.tensor([5], dtype=torch.int64, device=dev)

    def run(producer):
        monkeypatch.setattr(ops, "GN_PRODUCER_STATS", producer)
        gn = (32, 1e-5)
        y = ops.conv2d(x, wt, None, stride, 1, gn=gn) if kind == "conv" else ops.depthwise_conv2d(x, wt, stride, gn=gn)
        res = None
        if use_res:
            res = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(y.shape)).astype(np.float32)).to(dev)
        z = ops.group_norm_act(y, gamma, beta, 32, 1e-5, act, res, rate, 1234, counter)
        dz = torch.from_numpy(np.random.default_rng(4).standard_normal(tuple(z.shape)).astype(np.float32)).to(dev)
        grads = torch.autograd.grad(z, [x, wt, gamma, beta], dz)
        return y, z, grads

    y0, z0, g0 = run(False)
    assert not hasattr(y0, "_gn_rows")
    y1, z1, g1 = run(True)
    assert hasattr(y1, "_gn_rows"), "this shape is expected to take the producer-rows path"
    rows, rows_ps, per_group, g = y1._gn_rows
    assert torch.equal(y0, y1)                                   # the conv output itself: the same kernel arithmetic
    # the rows add up to the moments of y (fp64 restatement of normalization.py:30)
    cpg = cout // g
    r = rows.double().cpu().reshape(n, rows_ps, -1, 2).sum(1)    # [n, width, 2]
    if not per_group:
        r = r.reshape(n, g, cpg, 2).sum(2)
    cnt = y1.shape[1] * y1.shape[2] * cpg
    mean = (r[..., 0] / cnt).numpy()
    var = (r[..., 1] / cnt).numpy() - mean * mean
    mref, rref = _moments_ref(y1, g, 1e-5)
    assert_close(mean, mref, 1e-5, "mean from rows")
    assert np.max(np.abs(1.0 / np.sqrt(np.maximum(var, 0) + 1e-5) / rref - 1)) < 1e-5
    assert_close(z1.detach().cpu().numpy(), z0.detach().cpu().numpy(), 1e-5, "GroupNorm output")
    if rate > 0:
        assert torch.equal(z0 == 0, z1 == 0)                     # the same dropout mask
    for name, a, b in zip(("dx", "dw", "dgamma", "dbeta"), g1, g0):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 2e-5, name)
    y2, z2, _ = run(True)                                        # replay: same bits (fixed-order combination)
    assert torch.equal(z1, z2) and torch.equal(y2._gn_rows[0], rows)


def test_producer_statistics_decline_what_they_cannot_do(dev):
    """Shapes whose conv plan cannot emit statistics (a sample's pixels not whole tiles, split-K plans, a bias) fall back to the
    stand-alone GroupNorm: no statistics on the tensor, same result."""
    import ops
    rng = np.random.default_rng(2)
    for xs, cout, k in (((2, 15, 15, 64), 128, 1), ((2, 16, 16, 960), 160, 1)):
        x = _t(rng.standard_normal(xs).astype(np.float32), dev)
        wt = _t((rng.standard_normal((k, k, xs[3], cout)) / np.sqrt(xs[3])).astype(np.float32), dev)
        y = ops.conv2d(x, wt, None, 1, 1, gn=(32, 1e-5))
        assert not hasattr(y, "_gn_rows")
        assert torch.equal(y, ops.conv2d(x, wt))


def test_zero_kernel_clears_exactly_the_range(dev):
    """rn_zero (the gradient arena before a backward pass, the detector's counters): every float of the range, none beyond."""
    import _rn
    for count in (1, 3, 4, 1023, 1024 * 1024 + 5):
        t = torch.full((count + 8,), 7.0, device=dev)
        _rn.check(_rn.lib().rn_zero(_rn.f32(t), count, _rn.stream()), "rn_zero")
        torch.cuda.synchronize()
        assert float(t[:count].abs().sum()) == 0.0 and bool((t[count:] == 7.0).all())
