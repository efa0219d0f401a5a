"""The batched products of the Winograd convolutions evaluated on the bf16 matrix cores from EXACT three-way bf16 splits of the
fp32 operands (csrc/gemm_x3.hip, rn_set_product_mode(1)) against an fp64 reference, beside the exact fp32 matrix-core kernels
(mode 0): the split products must be as accurate as the fp32 instruction -- error relative to sum |a||b| within 2 x the fp32
kernel's own, and below 3e-7 -- on every operand layout the network uses (retinanet.py:37-62,85-106: forward [K][N] weights,
data gradient [N][K], weight gradient = A^T B with the sum over tiles split across blocks), ragged M / N, short and long K, and
operands spanning 12 orders of magnitude.  Then a whole Winograd conv layer forward + backward in both modes vs the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import tf_ops_ref as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def product_mode():
    import _rn
    L = _rn.lib()
    before = L.rn_get_product_mode()
    yield L.rn_set_product_mode
    L.rn_set_product_mode(before)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _operands(rng, shape, wide):
    x = rng.standard_normal(shape)
    if wide:            # magnitudes over 12 decades, mixed signs: the splits' exponents differ wildly inside one dot product
        x = x * 10.0 ** rng.uniform(-6, 6, shape)
    return x.astype(np.float32)


@pytest.mark.parametrize("M,K,N,nb,b_nk,wide", [(682, 256, 256, 4, 0, False), (682, 256, 256, 3, 1, False), (682, 256, 720, 2, 0, False),
                                                (100, 32, 64, 5, 0, False), (64, 1024, 128, 2, 1, False), (37, 36, 20, 3, 0, False),
                                                (300, 256, 256, 2, 0, True), (300, 256, 256, 2, 1, True)])
def test_batched_products_split_bf16_as_accurate_as_fp32(dev, product_mode, M, K, N, nb, b_nk, wide):
    import _rn
    rng = np.random.default_rng(M + K + N)
    A = _operands(rng, (nb, M, K), wide)
    B = _operands(rng, (nb, N, K) if b_nk else (nb, K, N), wide)
    Bm = np.swapaxes(B, 1, 2) if b_nk else B
    ref = np.einsum("bmk,bkn->bmn", A.astype(np.float64), Bm.astype(np.float64))
    mag = np.einsum("bmk,bkn->bmn", np.abs(A).astype(np.float64), np.abs(Bm).astype(np.float64))
    Ad, Bd = _t(A, dev), _t(B, dev)
    errs = []
    for mode in (0, 1):
        product_mode(mode)
        Cd = torch.full((nb, M, N), float("nan"), device=dev)
        _rn.check(_rn.lib().rn_gemm_batched(_rn.f32(Ad), _rn.f32(Bd), _rn.f32(Cd), M, K, N, nb, b_nk, _rn.stream()), "rn_gemm_batched")
        got = Cd.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all()
        errs.append(float((np.abs(got - ref) / mag).max()))
    print("products %dx%dx%d x%d b_nk=%d%s: error / sum|a||b|: fp32 MFMA %.2e, split bf16 %.2e" % (M, K, N, nb, b_nk, " wide" if wide else "", errs[0], errs[1]))
    assert errs[1] <= max(2.0 * errs[0], 3e-7), errs


@pytest.mark.parametrize("M,Kd,Nd,nb", [(682, 256, 256, 4), (682, 720, 256, 2), (150, 64, 32, 3)])
def test_backward_products_split_bf16(dev, product_mode, M, Kd, Nd, nb):
    """The two products of a Winograd backward pass (rn_winograd_bwd_products): dV = dM Urot^T and the split partial sums of
    dU = V^T dM (contraction over the M tiles, ragged last range); slabs summed in fp64 vs the fp64 reference, both modes."""
    import _rn
    L = _rn.lib()
    rng = np.random.default_rng(M + Kd)
    dM = _operands(rng, (nb, M, Kd), False)            # [T x Cout]
    Urot = _operands(rng, (nb, Nd, Kd), False)         # [Cin x Cout] in the data gradient's [N][K] layout
    V = _operands(rng, (nb, M, Nd), False)             # [T x Cin]
    ref_d = np.einsum("bmk,bnk->bmn", dM.astype(np.float64), Urot.astype(np.float64))
    mag_d = np.einsum("bmk,bnk->bmn", np.abs(dM).astype(np.float64), np.abs(Urot).astype(np.float64))
    ref_w = np.einsum("bmk,bmn->bkn", V.astype(np.float64), dM.astype(np.float64))       # [Cin x Cout]
    mag_w = np.einsum("bmk,bmn->bkn", np.abs(V).astype(np.float64), np.abs(dM).astype(np.float64))
    dMd, Ud, Vd = _t(dM, dev), _t(Urot, dev), _t(V, dev)
    out = []
    for mode in (0, 1):
        product_mode(mode)
        need = L.rn_winograd_bwd_products_workspace(M, Nd, Kd, nb)
        ws = torch.full((need // 4,), float("nan"), device=dev)
        Cd = torch.full((nb, M, Nd), float("nan"), device=dev)
        ns = C.c_int(0)
        _rn.check(L.rn_winograd_bwd_products(_rn.f32(dMd), _rn.f32(Ud), _rn.f32(Cd), M, Kd, Nd, _rn.f32(Vd), _rn.f32(dMd), Nd, Kd, nb,
                                             ws.data_ptr(), need, C.byref(ns), _rn.stream()), "rn_winograd_bwd_products")
        slabs = ws[:ns.value * nb * Nd * Kd].reshape(ns.value, nb, Nd, Kd).cpu().numpy().astype(np.float64)
        assert np.isfinite(slabs).all() and ns.value >= 1
        ed = float((np.abs(Cd.cpu().numpy().astype(np.float64) - ref_d) / mag_d).max())
        ew = float((np.abs(slabs.sum(0) - ref_w) / mag_w).max())
        out.append((ed, ew, ns.value))
    print("backward products M=%d: dgrad error fp32 %.2e / split %.2e; wgrad fp32 %.2e (%d slabs) / split %.2e (%d slabs)" %
          (M, out[0][0], out[1][0], out[0][1], out[0][2], out[1][1], out[1][2]))
    assert out[1][0] <= max(2.0 * out[0][0], 3e-7) and out[1][1] <= max(2.0 * out[0][1], 3e-7), out


@pytest.mark.parametrize("mode", [0, 1])
def test_winograd_layer_both_product_modes_vs_oracle(dev, product_mode, mode):
    """A 3x3 / 256 -> 256 conv over three pyramid levels through the Winograd path (forward, data gradient, weight gradient)
    in both product modes against the oracle's direct convolution: 1e-4 (the bar of tests/test_gpu_ops.py's conv tests)."""
    import ops
    product_mode(mode)
    rng = np.random.default_rng(3)
    xs = [rng.standard_normal((2, s, s, 256)).astype(np.float32) for s in (16, 8, 5)]
    w = (rng.standard_normal((3, 3, 256, 256)) / np.sqrt(9 * 256)).astype(np.float32)
    xc = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    wc = torch.from_numpy(w).requires_grad_(True)
    yc = [T.conv2d_same(x, wc, 1) for x in xc]
    dys = [rng.standard_normal(tuple(y.shape)).astype(np.float32) for y in yc]
    torch.autograd.backward(yc, [torch.from_numpy(d) for d in dys])
    xg = [_t(x, dev).requires_grad_(True) for x in xs]
    wg = _t(w, dev).requires_grad_(True)
    yg = ops.conv2d(xg, wg, None, 1)
    torch.autograd.backward(yg, [_t(d, dev) for d in dys])
    for i in range(3):
        assert_close(yg[i].detach().cpu().numpy(), yc[i].detach().numpy(), 1e-4, "winograd fwd level %d (mode %d)" % (i, mode))
        assert_close(xg[i].grad.cpu().numpy(), xc[i].grad.numpy(), 1e-4, "winograd dx level %d (mode %d)" % (i, mode))
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), 1e-4, "winograd dw (mode %d)" % mode)


# ---- round 6: the dense 1x1 / stride-1 convolutions on the same kernels (resnet.py:38-49,67-69, densenet.py:61-66,137-142)
@pytest.mark.parametrize("n,hw,cin,cout", [(2, 64, 64, 256), (2, 64, 256, 64), (1, 96, 128, 512), (2, 32, 1024, 256), (2, 16, 2048, 512),
                                           (2, 16, 512, 2048), (4, 40, 544, 128), (2, 40, 100, 36)])
def test_conv1x1_split_bf16_fwd_bwd_as_accurate_as_fp32(dev, product_mode, n, hw, cin, cout):
    """ops.conv2d of a 1x1 / stride-1 kernel (forward, data gradient, weight gradient) in both product modes against fp64, at the
    K = 64 .. 2 048 / N = 36 .. 2 048 shapes of the ResNeXt / DenseNet bottlenecks: the split-bf16 path must be as accurate as the
    fp32 matrix-core path (error relative to sum |a||b| within 2 x the fp32 kernel's own, and below 3e-7)."""
    import ops
    rng = np.random.default_rng(n * hw + cin + cout)
    x = rng.standard_normal((n, hw, hw, cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dy = rng.standard_normal((n, hw, hw, cout)).astype(np.float32)
    X, W2, DY = x.reshape(-1, cin).astype(np.float64), w.reshape(cin, cout).astype(np.float64), dy.reshape(-1, cout).astype(np.float64)
    ref = {"y": X @ W2, "dx": DY @ W2.T, "dw": X.T @ DY}
    mag = {"y": np.abs(X) @ np.abs(W2), "dx": np.abs(DY) @ np.abs(W2).T, "dw": np.abs(X).T @ np.abs(DY)}
    errs = []
    for mode in (0, 1):
        product_mode(mode)
        xg, wg = _t(x, dev).requires_grad_(True), _t(w, dev).requires_grad_(True)
        y = ops.conv2d(xg, wg, None, 1)
        y.backward(_t(dy, dev))
        got = {"y": y.detach().cpu().numpy().reshape(-1, cout), "dx": xg.grad.cpu().numpy().reshape(-1, cin), "dw": wg.grad.cpu().numpy().reshape(cin, cout)}
        assert all(np.isfinite(v).all() for v in got.values())
        errs.append({k: float((np.abs(got[k].astype(np.float64) - ref[k]) / mag[k]).max()) for k in ref})
    print("conv 1x1 %d x %d^2 x %d -> %d: error / sum|a||b| fp32 %s, split bf16 %s" % (
        n, hw, cin, cout, {k: "%.1e" % v for k, v in errs[0].items()}, {k: "%.1e" % v for k, v in errs[1].items()}))
    for k in ref:
        assert errs[1][k] <= max(2.0 * errs[0][k], 3e-7), (k, errs)


@pytest.mark.parametrize("mode", [0, 1])
def test_conv1x1_with_group_norm_statistics_both_modes(dev, product_mode, mode):
    """conv 1x1 -> GroupNorm -> ReLU with the GroupNorm's statistics taken from the conv's epilogue (ops.conv2d(gn=...): the rows the
    split-bf16 kernel writes have the layout of the fp32 kernel's), forward and all gradients vs the oracle, both modes."""
    import ops
    product_mode(mode)
    rng = np.random.default_rng(17 + mode)
    n, hw, cin, cout = 2, 64, 128, 256
    x = rng.standard_normal((n, hw, hw, cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    gamma, beta = (1 + 0.2 * rng.standard_normal(cout)).astype(np.float32), (0.1 * rng.standard_normal(cout)).astype(np.float32)
    dz = rng.standard_normal((n, hw, hw, cout)).astype(np.float32)
    xc, wc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(w).requires_grad_(True)
    gc, bc = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    zc = T.activation(T.group_norm(T.conv2d_same(xc, wc, 1), gc, bc), "relu")
    zc.backward(torch.from_numpy(dz))
    xg, wg = _t(x, dev).requires_grad_(True), _t(w, dev).requires_grad_(True)
    gg, bg = _t(gamma, dev).requires_grad_(True), _t(beta, dev).requires_grad_(True)
    y = ops.conv2d(xg, wg, None, 1, gn=(32, 1e-5))
    assert getattr(y, "_gn_rows", None) is not None, "the conv did not emit GroupNorm rows"
    zg = ops.group_norm_act(y, gg, bg, groups=32, act="relu")
    zg.backward(_t(dz, dev))
    assert_close(zg.detach().cpu().numpy(), zc.detach().numpy(), 1e-4, "conv1x1 + GN + relu (mode %d)" % mode)
    assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), 1e-4, "dx (mode %d)" % mode)
    assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), 1e-4, "dw (mode %d)" % mode)
    assert_close(gg.grad.cpu().numpy(), gc.grad.numpy(), 1e-4, "dgamma (mode %d)" % mode)
    assert_close(bg.grad.cpu().numpy(), bc.grad.numpy(), 1e-4, "dbeta (mode %d)" % mode)


@pytest.mark.parametrize("n,h,w,cin,cout,gn", [(2, 64, 64, 128, 512, False), (2, 50, 38, 256, 512, False), (2, 64, 64, 128, 512, True)])
def test_strided_dense_conv_through_the_patch_matrix(dev, product_mode, n, h, w, cin, cout, gn):
    """3x3 / stride-2 dense convs (the ResNeXt "down" convs, resnet.py:60-69) in product mode 1 run as split-bf16 products of an explicit
    patch matrix (csrc/im2col.hip): forward (optionally with the GroupNorm statistic rows of the conv's epilogue), data gradient (product
    + gather) and weight gradient against the oracle's direct convolution, even and odd maps (TF SAME padding 0/1 and 1/1), 1e-4;
    mode 0 (the fp32 implicit GEMM) beside it."""
    import ops
    rng = np.random.default_rng(h * w + cin)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    xc, wc = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(wt).requires_grad_(True)
    yc = T.conv2d_same(xc, wc, 2)
    dy = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    if gn:
        gamma, beta = (1 + 0.2 * rng.standard_normal(cout)).astype(np.float32), (0.1 * rng.standard_normal(cout)).astype(np.float32)
        gc, bc = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
        zc = T.group_norm(yc, gc, bc)
    else:
        zc = yc
    zc.backward(torch.from_numpy(dy))
    for mode in (1, 0):
        product_mode(mode)
        xg, wg = _t(x, dev).requires_grad_(True), _t(wt, dev).requires_grad_(True)
        if gn:
            gg, bg = _t(gamma, dev).requires_grad_(True), _t(beta, dev).requires_grad_(True)
            y = ops.conv2d(xg, wg, None, 2, gn=(32, 1e-5))
            zg = ops.group_norm_act(y, gg, bg, groups=32)
        else:
            zg = ops.conv2d(xg, wg, None, 2)
        zg.backward(_t(dy, dev))
        assert_close(zg.detach().cpu().numpy(), zc.detach().numpy(), 1e-4, "strided conv forward (mode %d)" % mode)
        assert_close(xg.grad.cpu().numpy(), xc.grad.numpy(), 1e-4, "strided conv dx (mode %d)" % mode)
        assert_close(wg.grad.cpu().numpy(), wc.grad.numpy(), 1e-4, "strided conv dw (mode %d)" % mode)
        if gn:
            assert_close(gg.grad.cpu().numpy(), gc.grad.numpy(), 1e-4, "dgamma (mode %d)" % mode)


def test_conv1x1_split_bf16_on_a_channel_slice(dev, product_mode):
    """The 1x1 entry points on a channel SLICE of a wider buffer (rn_conv_seg.x_ld / x_coff: the concat-free DenseNet block reads and
    writes slices of one [n, h, w, c_total] buffer, densenet.py:117-121): forward reads channels [coff, coff + cin) of x, the data
    gradient writes them into dx and leaves the other channels alone, the weight gradient reads them -- product mode 1 vs fp64."""
    import _rn, ops
    L = _rn.lib()
    product_mode(1)
    rng = np.random.default_rng(5)
    n, hw, ld, coff, cin, cout = 2, 64, 160, 32, 96, 128
    xb = rng.standard_normal((n, hw, hw, ld)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dy = rng.standard_normal((n, hw, hw, cout)).astype(np.float32)
    X = xb[..., coff:coff + cin].reshape(-1, cin).astype(np.float64)
    W2, DY = w.reshape(cin, cout).astype(np.float64), dy.reshape(-1, cout).astype(np.float64)
    xd, wd, dyd = _t(xb, dev), _t(w, dev), _t(dy, dev)
    geom = _rn.ConvGeom(1, 1, 1, cin, 1)
    y = torch.full((n, hw, hw, cout), float("nan"), device=dev)
    segs = ops._conv_segs([xd], wd, None, [y], None, None, x_ld=ld, x_coff=coff)
    assert L.rn_conv2d_fwd_workspace(segs, 1, C.byref(geom)) == 0
    _rn.check(L.rn_conv2d_fwd(segs, 1, C.byref(geom), None, 0, _rn.stream()), "rn_conv2d_fwd")
    mag = np.abs(X) @ np.abs(W2)
    assert float((np.abs(y.cpu().numpy().reshape(-1, cout) - X @ W2) / mag).max()) <= 3e-7
    dxb = torch.full((n, hw, hw, ld), 7.0, device=dev)
    segs = ops._conv_segs([xd], wd, None, None, [dyd], [dxb], x_ld=ld, x_coff=coff)
    _rn.check(L.rn_conv2d_dgrad(segs, 1, C.byref(geom), None, 0, _rn.stream()), "rn_conv2d_dgrad")
    got = dxb.cpu().numpy()
    assert np.all(got[..., :coff] == 7.0) and np.all(got[..., coff + cin:] == 7.0)            # the other channels are untouched
    magd = np.abs(DY) @ np.abs(W2).T
    assert float((np.abs(got[..., coff:coff + cin].reshape(-1, cin) - DY @ W2.T) / magd).max()) <= 3e-7
    need = L.rn_conv2d_wgrad_workspace(segs, 1, C.byref(geom))
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
    dw = torch.full((1, 1, cin, cout), float("nan"), device=dev)
    _rn.check(L.rn_conv2d_wgrad(segs, 1, C.byref(geom), _rn.f32(dw), 0, ws.data_ptr(), ws.numel(), _rn.stream(), None), "rn_conv2d_wgrad")
    magw = np.abs(X).T @ np.abs(DY)
    assert float((np.abs(dw.cpu().numpy().reshape(cin, cout) - X.T @ DY) / magw).max()) <= 3e-7


def test_strided_dense_conv_patch_matrix_in_pieces(dev, tmp_path):
    """Forward of a 3x3 / stride-2 dense conv whose patch matrix is built in equal PIECES of the batch (large inference batches: <= 1 GiB per
    piece; here RN_X3_IM2COL_PIECE_MB=40 forces 4 samples into 2 pieces of 2 in a fresh process), with the GroupNorm statistic rows of
    every piece landing in their place: bit-identical to the one-piece run of the same kernels."""
    import os, subprocess, sys
    script = r'''
import os, sys
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(9)
x = torch.from_numpy(rng.standard_normal((4, 64, 64, 128)).astype(np.float32)).to(dev)
w = torch.from_numpy((rng.standard_normal((3, 3, 128, 512)) / 34).astype(np.float32)).to(dev)
g, b = torch.ones(512, device=dev), torch.zeros(512, device=dev)
with torch.no_grad():
    y = ops.conv2d(x, w, None, 2, gn=(32, 1e-5))
    assert getattr(y, "_gn_rows", None) is not None
    z = ops.group_norm_act(y, g, b, groups=32)
np.save(sys.argv[2], np.stack([y.cpu().numpy(), z.cpu().numpy()]))
'''
    outs = []
    for cap in ("4096", "40"):
        out = str(tmp_path / ("y_%s.npy" % cap))
        env = dict(os.environ, RN_X3_IM2COL_PIECE_MB=cap)
        r = subprocess.run([sys.executable, "-c", script, os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])
    assert np.isfinite(outs[0]).all() and abs(float(outs[0][1].mean())) < 1e-3       # (the normalised output: zero mean per group)


# ---- the kernel operand pre-split in fragment order by its producer (gemm_x3_bfrag.hip FragB, include/rn_hip.h rn_x3_*bfrag*) ----
@pytest.fixture()
def bfrag_switch():
    import _rn
    L = _rn.lib()
    before = L.rn_get_x3_bfrag()
    yield L.rn_set_x3_bfrag
    L.rn_set_x3_bfrag(before)


@pytest.mark.parametrize("M,K,N,nb,b_nk,wide", [(682, 256, 256, 4, 0, False), (682, 256, 256, 3, 1, False), (682, 256, 720, 2, 0, False),
                                                (682, 720, 256, 2, 1, False), (682, 256, 36, 3, 0, False), (100, 32, 64, 5, 0, False),
                                                (64, 1024, 128, 2, 1, False), (37, 48, 20, 3, 0, False), (300, 256, 256, 2, 0, True),
                                                (300, 256, 256, 2, 1, True), (5000, 64, 256, 1, 0, False)])
def test_products_with_a_fragment_ordered_operand_are_bit_identical(dev, product_mode, bfrag_switch, M, K, N, nb, b_nk, wide):
    """rn_x3_pack_bfrag + rn_gemm_batched_bfrag against rn_gemm_batched on the same fp32 operands (product mode 1): the same three
    bf16 planes and the same matrix-core instructions in the same order -- every output bit equal; K = 720 ends in a half K-step
    (zeros past K), N = 720 / 36 / 20 end inside a 32-column block (zero padding columns)."""
    import ctypes as C
    import _rn
    L = _rn.lib()
    product_mode(1)
    bfrag_switch(1)
    assert L.rn_x3_bfrag_ok(M, K, N) == 1
    rng = np.random.default_rng(M + K + N + b_nk)
    Ad = _t(_operands(rng, (nb, M, K), wide), dev)
    Bd = _t(_operands(rng, (nb, N, K) if b_nk else (nb, K, N), wide), dev)
    want = torch.full((nb, M, N), float("nan"), device=dev)
    _rn.check(L.rn_gemm_batched(_rn.f32(Ad), _rn.f32(Bd), _rn.f32(want), M, K, N, nb, b_nk, _rn.stream()), "rn_gemm_batched")
    nbytes = L.rn_x3_bfrag_bytes(K, N, nb)
    assert nbytes == nb * ((N + 31) // 32) * 32 * K * 6
    img = torch.full((nbytes,), 0xff, dtype=torch.uint8, device=dev)          # (NaN patterns wherever the pack kernel does not write)
    _rn.check(L.rn_x3_pack_bfrag(_rn.f32(Bd), img.data_ptr(), K, N, nb, b_nk, _rn.stream()), "rn_x3_pack_bfrag")
    for fwd_name in (0, 1):
        got = torch.full((nb, M, N), float("nan"), device=dev)
        _rn.check(L.rn_gemm_batched_bfrag(_rn.f32(Ad), img.data_ptr(), _rn.f32(got), M, K, N, nb, fwd_name, _rn.stream()), "rn_gemm_batched_bfrag")
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), float((got - want).abs().max())
    # switched off (or mode 0, or K % 16 != 0): the entry refuses instead of misreading the operand
    bfrag_switch(0)
    assert L.rn_x3_bfrag_ok(M, K, N) == 0
    assert L.rn_gemm_batched_bfrag(_rn.f32(Ad), img.data_ptr(), _rn.f32(got), M, K, N, nb, 0, _rn.stream()) != 0
    bfrag_switch(1)
    assert L.rn_x3_bfrag_ok(M, 36, N) == 0 and L.rn_x3_bfrag_bytes(36, N, 1) == 0


@pytest.mark.parametrize("shapes,cout", [([(2, 64, 64), (2, 32, 32), (2, 16, 16), (2, 8, 8), (2, 4, 4)], 256), ([(2, 24, 40), (2, 7, 5)], 720),
                                         ([(1, 20, 12)], 36)])
@pytest.mark.parametrize("keep", [True, False])
def test_folded_winograd_layers_fragment_ordered_kernel_is_bit_identical(dev, product_mode, bfrag_switch, shapes, cout, keep, monkeypatch):
    """GroupNorm-folded head-tower layers (ops.wino_tower: the kernel transform writes U as the forward product's fragment image)
    with the switch on against off: outputs, input gradients and weight gradients bit-equal -- the two kernel-transform bodies
    compute the same bits (Wino<M>::g compiles without contraction), the product is bit-identical on equal operands (above).
    cout = 720 / 36: the class / box output convs (columns padded inside the last 32-block).  keep = False: the backward pass
    rebuilds V and Urot itself."""
    import _rn
    import ops
    L = _rn.lib()
    product_mode(1)
    monkeypatch.setattr(ops, "WINOGRAD_KEEP", keep)
    torch.manual_seed(3)
    cin = 256
    xs0 = [torch.randn(n, h, w, cin, device=dev) for n, h, w in shapes]
    tower = [(torch.randn(3, 3, cin, cin, device=dev) * 0.03, torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1) for _ in range(2)]
    w_out = torch.randn(3, 3, cin, cout, device=dev) * 0.03
    b_out = torch.randn(cout, device=dev) * 0.1
    res = []
    for on in (0, 1):
        bfrag_switch(on)
        assert L.rn_x3_bfrag_ok(100, cin, cin) == on
        xs = [x.clone().requires_grad_(True) for x in xs0]
        tw = [tuple(t.clone().requires_grad_(True) for t in layer) for layer in tower]
        wo, bo = w_out.clone().requires_grad_(True), b_out.clone().requires_grad_(True)
        assert ops.wino_tower_ok(xs, tw, wo, 32)
        ys = ops.wino_tower(xs, tw, wo, bo, 32, 1e-5, "elu")
        loss = sum((y * torch.cos(torch.arange(y.numel(), device=dev, dtype=torch.float32).reshape(y.shape) * 0.37)).sum() for y in ys)
        loss.backward()
        torch.cuda.synchronize()
        res.append([y.detach() for y in ys] + [x.grad for x in xs] + [t.grad for layer in tw for t in layer] + [wo.grad, bo.grad])
    for a, b in zip(*res):
        assert a is not None and b is not None
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), float((a - b).abs().max())
