"""fp16 inference path (BASELINE configs[4]: ResNeXt-50-FPN, fp16): f16 matrix-core convs with fp32 accumulation,
fp16-storage GroupNorm / pool / upsample.  Tolerances: a conv on fp16-rounded operands must match the fp32 oracle
on the SAME rounded operands to 1e-3 (fp32 accumulate + one fp16 rounding of the output); a whole network in
fp16 storage is compared with the fp32 oracle at 3e-2 of each tensor's max."""
import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import backbones_ref, tf_ops_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


CASES = [
    # n, h, w, cin, cout, k, stride, groups, bias, out_f32
    (2, 16, 16, 256, 256, 3, 1, 1, False, False),
    (2, 8, 8, 256, 720, 3, 1, 1, True, True),
    (1, 9, 7, 96, 256, 1, 1, 1, False, False),
    (2, 17, 15, 64, 128, 3, 2, 1, False, False),
    (2, 12, 12, 128, 128, 3, 1, 32, False, False),     # ResNeXt stage 2: 4 channels per group (8-byte gathers)
    (2, 10, 10, 256, 256, 3, 2, 32, False, False),     # 8 per group, stride 2
    (1, 6, 6, 1024, 1024, 3, 1, 32, False, False),     # 32 per group
    (2, 16, 32, 128, 128, 3, 1, 32, False, False),     # maps of whole 16 x 16 tiles: the super-group kernel (conv3x3_sg32_f16_kernel), 4 per group
    (1, 32, 16, 256, 256, 3, 1, 32, False, False),     # 8 per group
    (1, 16, 16, 512, 512, 3, 1, 32, False, False),     # 16 per group
    (2, 16, 16, 1024, 1024, 3, 1, 32, False, False),   # 32 per group: one group per super-group
    (2, 32, 64, 256, 256, 3, 2, 32, False, False),     # stride 2 on maps of whole 32 x 32 input tiles: the same kernel, 8 x 16 output tiles
    (1, 64, 32, 512, 512, 3, 2, 32, False, False),
    (1, 5, 5, 2048, 256, 1, 1, 1, False, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_f16(dev, case):
    import ops_f16
    import torch.nn.functional as F
    n, h, w, cin, cout, k, stride, groups, use_bias, out_f32 = case
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float16)
    wt = (rng.standard_normal((k, k, cin // groups, cout)) / np.sqrt(k * k * cin / groups)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if use_bias else None
    xf = torch.from_numpy(x.astype(np.float32))
    _, pt, pb = tf_ops_ref.same_pad_1d(h, k, stride)
    _, pl, pr = tf_ops_ref.same_pad_1d(w, k, stride)
    ref = F.conv2d(F.pad(xf.permute(0, 3, 1, 2), (pl, pr, pt, pb)), torch.from_numpy(wt).permute(3, 2, 0, 1),
                   torch.from_numpy(b) if use_bias else None, stride=stride, groups=groups).permute(0, 2, 3, 1).numpy()
    got = ops_f16.conv2d(torch.from_numpy(x).to(dev), torch.from_numpy(wt).to(dev),
                         torch.from_numpy(b).to(dev) if use_bias else None, stride, groups, out_f32=out_f32)
    assert got.dtype == (torch.float32 if out_f32 else torch.float16) and tuple(got.shape) == ref.shape
    assert_close(got.float().cpu().numpy(), ref, 2e-5 if out_f32 else 1e-3, "conv f16")


@pytest.mark.parametrize("cfg", ["4", "5"])
@pytest.mark.parametrize("case", [(2, 40, 38, 256, 384, 3, 1, 1, True, False), (1, 67, 33, 128, 264, 1, 1, 1, False, False),
                                  (2, 30, 30, 64, 256, 3, 2, 1, False, True),
                                  (1, 20, 20, 64, 256, 1, 1, 1, False, False)],       # ONE K-tile: the pipelined loop's prologue only
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_f16_large_tiles(dev, monkeypatch, case, cfg):
    """The 256x128 / 256x256 tile shapes (chosen automatically only for the 3x3 head convs of large batches): forced
    here on moderate shapes with ragged M / N edges; same checks as test_conv_f16.  cfg 5 with dense taps is the
    software-pipelined kernel (double-buffered LDS, one memory operation behind each MFMA): 1, 2, 9 and 36 K-tiles."""
    monkeypatch.setenv("RN_CONV_CFG", cfg)
    test_conv_f16(dev, case)


def test_packed_kernel_layout(dev):
    """rn_pack_weights_f16 (rn_hip.h): Wt[cout][K] and, 256-byte aligned behind it when K % 16 == 0, the same values in
    matrix-core fragment order Wf[ceil(cout/32)][K/16][64 lanes][8] -- lane l of a fragment holds output channel (l & 31) of
    the 32-channel block and k = 16 step + 8 (l >> 5) .. + 7; channels past cout are zeros."""
    import ctypes as C
    import _rn
    L = _rn.lib()
    rng = np.random.default_rng(11)
    # (3, 3, 4, 24): K = 36, no fragment copy; the last two: 3 x 3 kernels of 8 / 32 channels per group with cout % 32 == 0 carry a
    # third copy, Ws[cout / 32][9 taps][2][64 lanes][8]: the block-diagonal 32 x 32 kernel of every 32-channel super-group
    for kh, kw, cin_g, cout in ((3, 3, 32, 40), (1, 1, 64, 256), (3, 3, 4, 24), (3, 3, 8, 64), (3, 3, 32, 64)):
        w = rng.standard_normal((kh, kw, cin_g, cout)).astype(np.float32)
        K = kh * kw * cin_g
        nbytes = int(L.rn_pack_weights_f16_bytes(kh, kw, cin_g, cout))
        frag = K % 16 == 0
        off = (cout * K * 2 + 255) // 256 * 256 // 2
        rows = (cout + 31) // 32 * 32
        end = (off + rows * K) if frag else cout * K
        sg = kh == 3 and kw == 3 and cin_g in (4, 8, 16, 32) and cout % 32 == 0
        sg_off = (end * 2 + 255) // 256 * 256 // 2
        assert nbytes == ((sg_off + cout // 32 * 18 * 64 * 8) * 2 if sg else end * 2)
        buf = torch.zeros(nbytes // 2, dtype=torch.float16, device=dev)
        _rn.check(L.rn_pack_weights_f16(_rn.f32(torch.from_numpy(w).to(dev)), _rn.f16(buf), kh, kw, cin_g, cout, _rn.stream()),
                  "rn_pack_weights_f16")
        got = buf.cpu().numpy()
        wk = w.reshape(K, cout).astype(np.float16)                                       # [k][co]
        np.testing.assert_array_equal(got[:cout * K].reshape(cout, K), wk.T)
        if frag:
            wf = got[off:off + rows * K].reshape(rows // 32, K // 16, 64, 8)
            lane = np.arange(64)
            n = np.arange(rows // 32)[:, None, None, None] * 32 + (lane & 31)[None, None, :, None]
            k = np.arange(K // 16)[None, :, None, None] * 16 + ((lane >> 5) * 8)[None, None, :, None] + np.arange(8)[None, None, None, :]
            want = np.where(n < cout, wk[k, np.minimum(n, cout - 1)], np.float16(0))
            np.testing.assert_array_equal(wf, want)
        if sg:
            ws = got[sg_off:sg_off + cout // 32 * 18 * 64 * 8].reshape(cout // 32, 9, 2, 64, 8)
            lane = np.arange(64)
            sgi = np.arange(cout // 32)[:, None, None, None, None]
            tap = np.arange(9)[None, :, None, None, None]
            kk = np.arange(2)[None, None, :, None, None] * 16 + ((lane >> 5) * 8)[None, None, None, :, None] + np.arange(8)[None, None, None, None, :]
            ci, co = sgi * 32 + kk, sgi * 32 + (lane & 31)[None, None, None, :, None]
            w16 = w.reshape(9, cin_g, cout).astype(np.float16)
            want = np.where(ci // cin_g == co // cin_g, w16[tap, ci % cin_g, co], np.float16(0))
            np.testing.assert_array_equal(ws, want)


def test_group_norm_pool_upsample_f16(dev):
    import ops_f16
    rng = np.random.default_rng(3)
    for c, groups, act, aar, in_half in ((256, 32, "elu", False, True), (64, 32, "relu", False, False),
                                         (2048, 32, "relu", True, True), (256, 256, "relu", False, True)):
        x = (rng.standard_normal((2, 6, 5, c)) * 2 + 0.3).astype(np.float16 if in_half else np.float32)
        r = rng.standard_normal((2, 6, 5, c)).astype(np.float16)
        gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
        beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
        z = tf_ops_ref.group_norm(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(gamma), torch.from_numpy(beta), groups)
        rt = torch.from_numpy(r.astype(np.float32))
        ref = tf_ops_ref.activation(z + rt, act) if aar else tf_ops_ref.activation(z, act) + rt
        got = ops_f16.group_norm_act(torch.from_numpy(x).to(dev), torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev),
                                     groups, 1e-5, act, torch.from_numpy(r).to(dev), aar)
        assert got.dtype == torch.float16
        assert_close(got.float().cpu().numpy(), ref.numpy(), 2e-3, "gn f16 c=%d" % c)
    x = rng.standard_normal((2, 9, 9, 64)).astype(np.float16)
    got = ops_f16.max_pool(torch.from_numpy(x).to(dev))
    assert np.array_equal(got.float().cpu().numpy(), tf_ops_ref.max_pool_same(torch.from_numpy(x.astype(np.float32))).numpy())
    lat, top = rng.standard_normal((1, 8, 8, 16)).astype(np.float16), rng.standard_normal((1, 4, 4, 16)).astype(np.float16)
    got = ops_f16.upsample_add(torch.from_numpy(lat).to(dev), torch.from_numpy(top).to(dev))
    ref = torch.from_numpy(lat.astype(np.float32)) + tf_ops_ref.upsample_nearest_align_corners(torch.from_numpy(top.astype(np.float32)), 8, 8)
    assert_close(got.float().cpu().numpy(), ref.numpy(), 1e-3, "upsample_add f16")


def test_resnext_fpn_fp16_inference_matches_fp32(dev):
    """Whole RetinaNet(resnet_50) forward in fp16 storage vs its own fp32 forward and vs the fp32 oracle backbone."""
    import layers, levels, retinanet
    torch.manual_seed(5)
    net = retinanet.RetinaNet('resnet_50', levels.build_levels(), 80, layers.elu, 0.0)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("beta"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    params = {k[len("base."):]: v.detach().clone() for k, v in net.named_parameters() if k.startswith("base.backbone")}
    net.to(dev)
    x = torch.randn(2, 256, 256, 3)      # C5 = 8x8: the per-channel norms of the deep stages still average 64 values
    with torch.no_grad():
        ref32 = net(x.to(dev), training=False)
        feats32 = backbones_ref.resnext50_forward(params, x)
        layers.set_inference_dtype('f16')
        try:
            out16 = net(x.to(dev), training=False)
            feats16 = net.base.backbone(x.to(dev), training=False)
            layers.set_inference_dtype('f16', outputs='f32')
            assert net(x.to(dev), training=False)["classifications"]["P5"].dtype == torch.float32
        finally:
            layers.set_inference_dtype('f32')
    def rel_l2(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / b.norm())

    worst = 0.0
    for k in ("C3", "C4", "C5"):
        assert feats16[k].dtype == torch.float16
        worst = max(worst, rel_l2(feats16[k].float(), feats32[k]))
        print(k, "rel L2", rel_l2(feats16[k].float(), feats32[k]))
    for k in ("P3", "P4", "P5", "P6", "P7"):
        a, b = out16["classifications"][k], ref32["classifications"][k]
        assert a.dtype == torch.float16 and a.shape == b.shape          # logits leave the net in fp16 (configs[4]); fp32 on request
        a = a.float()
        e = max(rel_l2(a - a.mean(), b - b.mean()), rel_l2(out16["regressions"][k].float(), ref32["regressions"][k]))
        print(k, "rel L2", e)
        if k in ("P6", "P7"):
            # 2x2 / 1x1 maps at this input size: GroupNorm over a handful of values amplifies the fp16 rounding
            assert e <= 1e-1, (k, e)
        else:
            worst = max(worst, e)
    print("fp16 vs fp32 worst relative L2 error", worst)
    # fp16 storage (2^-11 per rounding) through ~75 conv + GroupNorm layers of a randomly initialised ReLU net with
    # per-channel norms: this net amplifies a single rounding ~60x (its fp32 forward differs from the fp32 oracle by
    # ~1e-5..1e-4 for 6e-8 roundings, test_gpu_backbones), so 2^-11 * 60 ~ 3e-2 is what fp16 storage costs here
    assert worst <= 6e-2


@pytest.mark.parametrize("project,size,cin,f", [(True, 32, 64, 64), ("down", 32, 256, 128), (False, 16, 512, 128), (False, 32, 1024, 256)])
def test_resnext_bottleneck_fp16_folded_equals_layer_by_layer(dev, project, size, cin, f):
    """The ResNeXt bottleneck with its GroupNorms folded into the convs (rn_conv2d_fwd_f16_fold: statistics from the conv
    epilogue, GroupNorm + ReLU applied to the next conv's operand on load) vs conv -> three-kernel GroupNorm -> conv: the same
    fp32 arithmetic on the same fp16 values, so they agree to an fp16 rounding or two (2e-3 of the range); and the folded
    block vs the fp32 oracle block at the network-level fp16 tolerance."""
    import layers, ops_f16, resnet
    torch.manual_seed(11)
    blk = resnet.ResNeXt_Bottleneck(f, project=project, kernel_initializer=layers.VarianceScaling(factor=2.0),
                                    kernel_regularizer=None, in_channels=cin)
    blk.build(cin)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if name.endswith("gamma"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("beta"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    blk.to(dev)
    x = torch.randn(2, size, size, cin, generator=g).to(dev)
    xh = ops_f16.to_half(x)
    with torch.no_grad():
        ref32 = blk(x, training=False)
        layers.set_inference_dtype('f16')
        try:
            assert ops_f16.FOLD
            folded = blk._call_f16_folded(xh)
            assert folded is not None, "this shape must fold"
            ops_f16.FOLD = False
            plain = blk(xh, training=False)
        finally:
            ops_f16.FOLD = True
            layers.set_inference_dtype('f32')
    assert folded.dtype == torch.float16 and folded.shape == plain.shape
    assert_close(folded.float().cpu().numpy(), plain.float().cpu().numpy(), 2e-3, "folded vs layer-by-layer fp16 bottleneck")
    err = float((folded.float() - ref32).norm() / ref32.norm())
    assert err < 1e-2, "folded fp16 bottleneck vs fp32: rel L2 %.3e" % err


@pytest.mark.parametrize("n,rows,c,groups", [(3, 37, 256, 32), (2, 256, 128, 32), (2, 5, 2048, 32), (2, 9, 64, 32),
                                              (2, 33, 96, 32), (1, 4, 320, 32)])
def test_group_norm_finalize_matches_fp64(dev, n, rows, c, groups):
    """rn_group_norm_finalize (the merge of a folded conv's per-tile statistic rows into mean / rstd) against fp64 numpy: the
    64-channel-slab kernel (c % 64 == 0, whole groups per slab) and the per-group kernel (every other shape)."""
    import _rn
    rng = np.random.default_rng(5)
    hw = rows * 64
    part = rng.normal(0.3, 1.0, size=(2, n * rows, c)).astype(np.float32)
    part[1] = np.abs(part[1]) * 40 + 20            # sums of squares: large enough for a positive variance
    p = torch.from_numpy(part).to(dev)
    mean = torch.empty((n, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    _rn.check(_rn.lib().rn_group_norm_finalize(_rn.f32(p), n, rows, hw, c, groups, 1e-5, _rn.f32(mean), _rn.f32(rstd), _rn.stream()),
              "rn_group_norm_finalize")
    cpg = c // groups
    s = part.astype(np.float64).reshape(2, n, rows, groups, cpg).sum(axis=(2, 4))
    m = s[0] / (hw * cpg)
    var = np.maximum(s[1] / (hw * cpg) - m * m, 0.0)
    assert_close(mean.cpu().numpy(), m.astype(np.float32), 1e-6, "mean")
    assert_close(rstd.cpu().numpy(), (1.0 / np.sqrt(var + 1e-5)).astype(np.float32), 1e-6, "rstd")


@pytest.mark.parametrize("size,cmid,cout", [(32, 128, 256), (16, 64, 512), (32, 320, 256)])
def test_pipelined_conv_with_groupnorm_on_load(dev, monkeypatch, size, cmid, cout):
    """ResNeXt's conv 3 on the software-pipelined 256 x 256 kernel (forced onto a small shape): ReLU(GN(y2)) applied between the
    staging registers and LDS (one mixed-precision fma per element) against the same conv on the MATERIALISED ReLU(GN(y2)) -- the
    same fp32 arithmetic on the same fp16 values: equal to an fp16 rounding (1e-3 of the range); its output statistics against
    fp64 sums of its own output.  2, 1 and 5 K-tiles."""
    import layers, ops_f16
    monkeypatch.setenv("RN_CONV_CFG", "5")
    g = torch.Generator().manual_seed(cmid + cout)
    x = (torch.randn(2, size, size, cmid, generator=g) * 1.5 + 0.3).to(dev).half()
    w2 = (torch.randn(1, 1, cmid, cmid, generator=g) / cmid ** 0.5).to(dev)
    w3 = (torch.randn(1, 1, cmid, cout, generator=g) / cmid ** 0.5).to(dev)

    class Norm(object):
        def __init__(self, c):
            self.groups, self.eps = 32, 1e-5
            self.gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
            self.beta = (0.1 * torch.randn(c, generator=g)).to(dev)
        def build(self, c, d):
            pass
    n2, n3 = Norm(cmid), Norm(cout)
    p2 = ops_f16.conv2d_norm(x, w2, n2, act='relu')
    assert p2 is not None
    folded = ops_f16.conv2d_norm(p2, w3, n3, act='relu')            # GroupNorm + ReLU on load
    plain = ops_f16.conv2d_norm(p2.materialise(), w3, n3, act='relu')
    assert folded is not None and plain is not None
    assert_close(folded.y.float().cpu().numpy(), plain.y.float().cpu().numpy(), 1e-3, "conv 3: GroupNorm on load vs materialised input")
    y = folded.y.float().cpu().numpy().astype(np.float64).reshape(2, size * size, 32, cout // 32)
    mean = y.mean(axis=(1, 3))
    rstd = 1.0 / np.sqrt(y.var(axis=(1, 3)) + 1e-5)
    assert_close(folded.mean.cpu().numpy(), mean.astype(np.float32), 1e-5, "statistics of the folded conv: mean")
    assert_close(folded.rstd.cpu().numpy(), rstd.astype(np.float32), 1e-4, "statistics of the folded conv: rstd")


def test_stem_conv_statistics_and_fused_norm_pool(dev):
    """ResNeXt's stem in fp16 inference: the 7x7/2 conv of the 4-channel-padded image with the GroupNorm statistics from its
    epilogue (4-wide operand vectors), then GroupNorm + ReLU + 3x3/2 max pool in one pass (rn_maxpool_gn_fwd_f16) -- bit-equal to
    the apply pass followed by the plain max pool; statistics against fp64 sums of the conv's own output."""
    import layers, ops_f16
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 64, 96, 3, generator=g).to(dev)
    w = (torch.randn(7, 7, 3, 64, generator=g) / (147 ** 0.5)).to(dev)

    class Norm(object):
        def __init__(self, c):
            self.groups, self.eps = 32, 1e-5
            self.gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
            self.beta = (0.1 * torch.randn(c, generator=g)).to(dev)
        def build(self, c, d):
            pass
    n = Norm(64)
    x4 = ops_f16.image_to_half4(img)
    p = ops_f16.conv2d_norm(x4, w, n, 'relu', 2, 1)
    assert p is not None, "the stem conv must emit its statistics"
    plain = ops_f16.conv2d(x4, w, None, 2, 1)
    assert torch.equal(p.y, plain)
    y = p.y.float().cpu().numpy().astype(np.float64).reshape(2, 32 * 48, 32, 2)
    assert_close(p.mean.cpu().numpy(), y.mean(axis=(1, 3)).astype(np.float32), 1e-5, "stem statistics: mean")
    assert_close(p.rstd.cpu().numpy(), (1.0 / np.sqrt(y.var(axis=(1, 3)) + 1e-5)).astype(np.float32), 1e-4, "stem statistics: rstd")
    fused = ops_f16.max_pool_norm(p, 3, 2)
    two = ops_f16.max_pool(p.materialise(), 3, 2)
    assert fused.shape == two.shape == (2, 16, 24, 64)
    assert torch.equal(fused, two)


@pytest.mark.parametrize("act,stride", [("elu", 1), ("relu", 2), (None, 1), ("elu", 2)])
def test_super_group_conv_with_pending_groupnorm_equals_materialised(dev, act, stride):
    """The grouped 3 x 3 kernel on LDS-resident patches (conv3x3_sg32_f16_kernel) applying a PENDING GroupNorm + activation while
    the patch goes to LDS (FIN = 2: ReLU through v_fma_mixlo/hi_f16 + packed max; FIN = 1: any activation in fp32) against the
    same kernel on the materialised tensor: the same fp32 arithmetic on the same fp16 values -- equal bits for FIN = 1, an fp16 ulp
    of the activation for FIN = 2; zero padding stays zero AFTER the activation (beta != 0 would show at the borders)."""
    import ops_f16
    g = torch.Generator().manual_seed(31 + stride)
    c = 256
    x = (torch.randn(2, 32, 64, c, generator=g) * 1.3 + 0.2).to(dev).half()
    w1 = (torch.randn(1, 1, c, c, generator=g) / c ** 0.5).to(dev)
    w2 = (torch.randn(3, 3, c // 32, c, generator=g) / (9 * c / 32) ** 0.5).to(dev)

    class Norm(object):
        def __init__(self):
            self.groups, self.eps = 32, 1e-5
            self.gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
            self.beta = (0.5 + 0.1 * torch.randn(c, generator=g)).to(dev)
        def build(self, c_, d):
            pass
    n1, n2 = Norm(), Norm()
    p1 = ops_f16.conv2d_norm(x, w1, n1, act=act)
    assert ops_f16.sg_kernel_takes(p1.y.shape, w2, stride, 32)
    folded = ops_f16.conv2d_norm(p1, w2, n2, act='relu', stride=stride, groups=32)
    plain = ops_f16.conv2d_norm(p1.materialise(), w2, n2, act='relu', stride=stride, groups=32)
    assert folded is not None and plain is not None and folded.y.shape == (2, 32 // stride, 64 // stride, c)
    if act == "relu":
        # v_fma_mixlo/hi_f16 rounds the exact fma ONCE, to fp16; the apply pass rounds it to fp32 first: the rare double-rounding
        # cases differ by one fp16 ulp of the activation (the mixed-precision instruction is the more exact one)
        assert_close(folded.y.float().cpu().numpy(), plain.y.float().cpu().numpy(), 1e-3, "GroupNorm + ReLU on the patch load vs materialised")
        assert_close(folded.mean.cpu().numpy(), plain.mean.cpu().numpy(), 1e-4, "mean")
        assert_close(folded.rstd.cpu().numpy(), plain.rstd.cpu().numpy(), 1e-4, "rstd")
    else:
        assert torch.equal(folded.y, plain.y), float((folded.y.float() - plain.y.float()).abs().max())
        assert torch.equal(folded.mean, plain.mean) and torch.equal(folded.rstd, plain.rstd)


@pytest.mark.parametrize("act,after", [("relu", True), (None, False), ("elu", True)])
def test_apply_with_pending_residual_norm(dev, act, after):
    """rn_group_norm_apply_res_f16 (the residual is a raw conv output with its own activation-free GroupNorm) against the apply pass
    on the MATERIALISED residual: the fused pass skips one fp16 rounding of the residual, so the two agree to an fp16 rounding of it
    (1e-3 of the range), and the fused one is the closer of the two to the fp32 result."""
    import ops_f16
    g = torch.Generator().manual_seed(5)
    n, h, w, c = 2, 24, 20, 256
    mk = lambda: (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.3).to(dev).half()
    y, r = mk(), mk()
    stat = lambda t: (t.float().reshape(n, h * w, 32, c // 32).mean(dim=(1, 3)), 1.0 / torch.sqrt(t.float().reshape(n, h * w, 32, c // 32).var(dim=(1, 3), unbiased=False) + 1e-5))
    gam = lambda: (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
    bet = lambda: (0.1 * torch.randn(c, generator=g)).to(dev)
    my, ry = stat(y); mr, rr = stat(r)
    py = ops_f16.Pending(y, my.contiguous(), ry.contiguous(), gam(), bet(), 32, act)
    pr = ops_f16.Pending(r, mr.contiguous(), rr.contiguous(), gam(), bet(), 32, None)
    fused = py.materialise(residual=pr, act_after_residual=after)
    two = py.materialise(residual=pr.materialise(), act_after_residual=after)
    assert_close(fused.float().cpu().numpy(), two.float().cpu().numpy(), 1e-3, "fused residual norm vs materialised residual")
    # fp32 reference of the same expression
    def gn(t, p):
        xh = (t.float().reshape(n, h * w, 32, c // 32) - p.mean.reshape(n, 1, 32, 1)) * p.rstd.reshape(n, 1, 32, 1)
        return xh.reshape(n, h, w, c) * p.gamma + p.beta
    f = {"relu": torch.relu, "elu": torch.nn.functional.elu, None: (lambda t: t)}[act]
    ref = f(gn(y, py) + gn(r, pr)) if after else f(gn(y, py)) + gn(r, pr)
    e_f = float((fused.float() - ref).abs().max()); e_t = float((two.float() - ref).abs().max())
    assert e_f <= e_t * 1.05 + 1e-6, (e_f, e_t)


@pytest.mark.parametrize("act", ["elu", "relu"])
@pytest.mark.parametrize("shapes", [[(3, 32, 32), (3, 16, 16), (3, 8, 8), (3, 4, 4)],        # P7-like last level: 16 pixels per sample, no rows
                                    [(2, 64, 48), (2, 16, 16), (2, 2, 2)],
                                    [(1, 16, 16)]])
def test_head_tower_block_with_epilogue_statistics_on_all_levels(dev, monkeypatch, act, shapes):
    """ops_f16.conv_norm_act_levels (round 6: the tower conv of every pyramid level in one launch with the GroupNorm statistics of its
    output from the epilogue, finalise, apply; a level whose conv tiles straddle samples is summed from its tensor by the finalise
    blocks) against conv2d + group_norm_act: the conv outputs are the same kernel's (bit-equal), the statistics are fp64 sums of
    the same fp16 values in another order, so the normalised tensors agree to an fp16 rounding (1e-3 of the range); the statistics
    themselves against fp64 sums of the conv output."""
    import ops_f16
    monkeypatch.setenv("RN_CONV_CFG", "5")          # the 256 x 256 tile (what the cfg-5 heads run) on these small maps
    g = torch.Generator().manual_seed(len(shapes) * 7 + len(act))
    c = 256
    xs = [(torch.randn(n, h, w, c, generator=g) * 1.2 + 0.2).to(dev).half() for n, h, w in shapes]
    wt = (torch.randn(3, 3, c, c, generator=g) / (9 * c) ** 0.5).to(dev)

    class Norm(object):
        def __init__(self):
            self.groups, self.eps = 32, 1e-5
            self.gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
            self.beta = (0.1 * torch.randn(c, generator=g)).to(dev)
        def build(self, c_, d):
            pass
    norm = Norm()
    got = ops_f16.conv_norm_act_levels(xs, wt, norm, act)
    assert got is not None and len(got) == len(xs)
    raw = ops_f16.conv2d(xs, wt)
    want = ops_f16.group_norm_act(raw, norm.gamma, norm.beta, groups=32, eps=1e-5, act=act)
    for i, (a, b, r) in enumerate(zip(got, want, raw)):
        assert a.shape == b.shape and a.dtype == torch.float16
        assert_close(a.float().cpu().numpy(), b.float().cpu().numpy(), 1e-3, "level %d" % i)
        # ... and against an fp64 GroupNorm of the conv output itself
        y = r.float().cpu().numpy().astype(np.float64)
        n, h, w, _ = y.shape
        yg = y.reshape(n, h * w, 32, c // 32)
        mean, var = yg.mean(axis=(1, 3), keepdims=True), yg.var(axis=(1, 3), keepdims=True)
        z = ((yg - mean) / np.sqrt(var + 1e-5)).reshape(n, h, w, c) * norm.gamma.cpu().numpy().astype(np.float64) + norm.beta.cpu().numpy().astype(np.float64)
        z = np.where(z > 0, z, np.expm1(np.minimum(z, 0))) if act == "elu" else np.maximum(z, 0)
        assert_close(a.float().cpu().numpy(), z.astype(np.float32), 2e-3, "level %d vs fp64" % i)
    # switched off: the caller's fallback
    monkeypatch.setattr(ops_f16, "HEAD_EPILOGUE_STATS", False)
    assert ops_f16.conv_norm_act_levels(xs, wt, norm, act) is None
