"""The CPU oracle against every still-valid known-answer vector of the reference's own
tests and against the fixture made from the reference's levels.py."""
import os

import numpy as np
import torch

import reference_kats as K
from oracle import dataset_ref, levels_ref, losses_ref, tf_ops_ref, utils_ref


def test_levels_match_reference_module(golden_dir):
    ref = np.load(os.path.join(golden_dir, "levels_reference.npz"))
    pyr = levels_ref.pyramid()
    assert list(pyr.keys()) == [str(k) for k in ref["keys"]] == ["P3", "P4", "P5", "P6", "P7"]
    assert levels_ref.num_anchors() == int(ref["num_anchors"]) == 9
    for k in pyr:
        assert np.array_equal(pyr[k], ref["anchor_sizes_" + k])       # fp64, exact
    assert np.array_equal(levels_ref.box_size(32, (1, 2), 1), ref["box_size_32_1x2_1"])
    assert np.array_equal(levels_ref.anchor_table(32, [(1, 4)], [1, 2]), ref["level_32_1x4"])


def test_levels_kats():
    c = K.BOX_SIZE_CASE
    b = levels_ref.box_size(c["base"], c["aspect"], c["scale"])
    assert len(b) == 2 and np.isclose(b.prod(), c["area"]) and b[1] / b[0] == c["ratio"]
    lc = K.LEVEL_CASE
    assert np.array_equal(levels_ref.anchor_table(lc["base"], lc["aspects"], lc["scales"]), lc["expected"])


def test_anchor_size_modes():
    px = levels_ref.pyramid()["P3"]
    t = levels_ref.normalized_anchor_sizes(px, (512, 512), "trunc_int")
    f = levels_ref.normalized_anchor_sizes(px, (512, 512), "float")
    assert t.dtype == f.dtype == np.float32
    assert np.array_equal(t[0], np.float32([22 / 512, 45 / 512]))          # 22.63 -> 22, 45.25 -> 45
    assert np.allclose(f[0] * 512, [22.627417, 45.254834])
    assert np.array_equal(t[3], f[3])                                       # 32 x 32 is integral


def test_grid_and_decode_kats():
    a = utils_ref.anchor_relative_to_image_relative(K.ANCHOR_REL_INPUT)
    assert a.shape == (1, 3, 4, 1, 4) and np.allclose(a, K.ANCHOR_REL_EXPECTED)
    b = utils_ref.anchor_boxmap(K.ANCHOR_BOXMAP_GRID, K.ANCHOR_BOXMAP_ANCHORS)
    assert b.shape == (1, 3, 4, 1, 4) and np.allclose(b, K.ANCHOR_BOXMAP_EXPECTED)
    c = utils_ref.center_to_corner(K.CENTER_CORNER_INPUT)
    assert np.array_equal(c, K.CENTER_CORNER_EXPECTED)
    s = utils_ref.scale_regression(K.SCALE_REGR_INPUT, K.SCALE_REGR_ANCHORS)
    assert np.array_equal(s, K.SCALE_REGR_EXPECTED)


def test_iou_kat():
    v = utils_ref.iou(K.IOU_A, K.IOU_B)
    assert v.shape == (4,) and np.allclose(v, K.IOU_EXPECTED)


def test_classmap_and_merge_kats():
    assert np.array_equal(utils_ref.classmap_decode(K.CLASSMAP), K.CLASSMAP_FG_EXPECTED)
    assert np.array_equal(utils_ref.compact_trainable(K.MERGE_OUTPUTS, K.MERGE_MASKS), K.MERGE_EXPECTED)


def test_huber_kat():
    v = losses_ref.regression_loss(torch.from_numpy(K.HUBER_LABELS), torch.from_numpy(K.HUBER_LOGITS),
                                   torch.from_numpy(K.HUBER_FG))
    assert float(v) == K.HUBER_EXPECTED


def test_assignment_kat():
    lv = K.ASSIGN_LEVEL
    sizes = levels_ref.anchor_table(lv["base"], lv["aspects"], lv["scales"])
    for mode in ("trunc_int", "float"):
        cls, reg, trainable, _ = dataset_ref.level_labels(
            K.ASSIGN_IMAGE_SIZE, K.ASSIGN_CLASS_IDS, K.ASSIGN_BOXES, sizes, K.ASSIGN_FACTOR, 401, mode)
        ids = np.where(cls.max(-1) > 0, cls.argmax(-1), 0)
        assert np.array_equal(ids, K.ASSIGN_CLASSMAP_EXPECTED)
        for pos in ((0, 0, 0), (0, 0, 1), (1, 1, 0)):
            assert np.allclose(reg[pos], K.ASSIGN_MATCHED_REGRESSION, atol=1e-6)
        assert trainable.shape == (2, 2, 2)


def test_flip_kat():
    r = {"P3": K.FLIP_REGR_INPUT}
    m = {"P3": np.ones(K.FLIP_REGR_INPUT.shape[:3], dtype=bool)}
    c = {"P3": K.FLIP_REGR_INPUT[..., :1]}
    _, fr, _, _ = dataset_ref.flip(c, r, m)
    exp = K.FLIP_REGR_INPUT[:, ::-1].copy()
    exp[..., 1] *= -1
    assert np.array_equal(fr["P3"], exp)


def test_same_padding_rule():
    assert tf_ops_ref.same_pad_1d(512, 3, 2) == (256, 0, 1)      # even n, k=3 -> (0, 1)
    assert tf_ops_ref.same_pad_1d(25, 3, 2) == (13, 1, 1)        # odd n -> (1, 1)
    assert tf_ops_ref.same_pad_1d(800, 7, 2) == (400, 2, 3)
    assert tf_ops_ref.same_pad_1d(64, 3, 1) == (64, 1, 1)


def test_nn_resize_rule():
    for n in (2, 4, 8, 16, 32, 50):
        assert np.array_equal(tf_ops_ref.nn_resize_index(2 * n, n), np.arange(2 * n) // 2)
    # odd sizes (scale 600 -> 75/38/19) are NOT floor(dst/2)
    idx = tf_ops_ref.nn_resize_index(75, 38)
    assert idx[0] == 0 and idx[-1] == 37 and np.all(np.diff(idx) >= 0)


def test_conv_oracle_vs_naive_loops():
    rng = np.random.default_rng(1)
    for (h, w, ci, co, k, s) in ((7, 6, 5, 4, 3, 1), (8, 8, 3, 6, 3, 2), (9, 7, 4, 4, 3, 2), (5, 5, 8, 3, 1, 1),
                                 (10, 9, 3, 4, 7, 2)):
        x = rng.standard_normal((2, h, w, ci)).astype(np.float32)
        wt = rng.standard_normal((k, k, ci, co)).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        got = tf_ops_ref.conv2d_same(torch.from_numpy(x), torch.from_numpy(wt), s, torch.from_numpy(b)).numpy()
        ref = tf_ops_ref.conv2d_same_naive(x, wt, s, b)
        assert got.shape == ref.shape
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_group_norm_oracle_vs_numpy():
    rng = np.random.default_rng(2)
    for c in (16, 24, 144, 256):
        x = rng.standard_normal((2, 5, 4, c)).astype(np.float32) * 3 + 1
        g = rng.standard_normal(c).astype(np.float32)
        b = rng.standard_normal(c).astype(np.float32)
        got = tf_ops_ref.group_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b)).numpy()
        G = tf_ops_ref.gn_groups(c)
        xg = x.astype(np.float64).reshape(2, 5, 4, G, c // G)
        mu = xg.mean(axis=(1, 2, 4), keepdims=True)
        var = xg.var(axis=(1, 2, 4), keepdims=True)
        ref = ((xg - mu) / np.sqrt(var + 1e-5)).reshape(x.shape) * g + b
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-5)
    assert tf_ops_ref.gn_groups(144) == 24 and tf_ops_ref.gn_groups(256) == 32 and tf_ops_ref.gn_groups(16) == 16


def test_nms_reference_vs_vectorised():
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 200):
        c = rng.uniform(0.2, 0.8, (n, 2)); s = rng.uniform(0.02, 0.3, (n, 2))
        boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
        scores = rng.uniform(0.5, 1, n).astype(np.float32)
        if n > 4:
            scores[3] = scores[1]                       # tie -> lower index first
        a = utils_ref.nms_indices(boxes, scores)
        b = utils_ref.nms_indices_vectorised(boxes, scores)
        assert np.array_equal(a, b)
    # tie-break + threshold semantics on a hand case: identical boxes, equal scores
    boxes = np.float32([[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 1, 0.5]])
    scores = np.float32([0.9, 0.9, 0.8])
    assert list(utils_ref.nms_indices(boxes, scores)) == [0, 2]    # IoU(0,2)=0.5 is NOT > 0.5


def test_oracle_e2e_fixture_is_stable(golden_dir):
    from oracle import model_ref, train_ref
    fx = np.load(os.path.join(golden_dir, "oracle_e2e_tiny.npz"))
    params = model_ref.init_params("mobilenet_v2", num_classes=3, seed=0)
    image = torch.from_numpy(fx["image"])
    with torch.no_grad():
        out = model_ref.retinanet_forward(params, image, 3)
    assert np.allclose(out["classifications"]["P5"].numpy(), fx["cls_P5"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out["regressions"]["P7"].numpy(), fx["reg_P7"], rtol=1e-4, atol=1e-5)
    sizes = K.pyramid_sizes(64)
    for k, s in zip(("P3", "P4", "P5", "P6", "P7"), sizes):
        assert out["classifications"][k].shape == (2, s, s, 9, 3)
        assert out["regressions"][k].shape == (2, s, s, 9, 4)


def test_dropout_mask_restatement_vector_equals_scalar_and_sites_cover_the_reference():
    """oracle/dropout_ref.py: the vectorised numpy form of the mask function == the plain-integer form element by element
    (small / 64-bit seeds, indices beyond 2^32); keep fraction and tf.nn.dropout scaling; and the hook visits exactly the
    reference's dropout sites (mobilenet_v2.py:62,71,79,117,184: stem + 17 x 3 + output conv = 53)."""
    import torch
    from oracle import dropout_ref, model_ref
    idx = np.array([0, 1, 2, 3, 1000, 2 ** 31 + 5, 2 ** 32 - 1, 2 ** 32 + 7, 2 ** 40 + 3], dtype=np.uint64)
    for seed in (0, 0x5EED + 0x9E3779B1, (1 << 40) + 12345, 0x632BE59BD9B4E019, 2 ** 64 - 1):
        v = dropout_ref.uniform01(seed, idx)
        w = np.array([dropout_ref.uniform01_scalar(seed, int(i)) for i in idx], dtype=np.float32)
        assert np.array_equal(v, w) and (v >= 0).all() and (v < 1).all()
    keep = dropout_ref.keep_mask(123, (4, 32, 32, 24), 0.2)
    assert abs(keep.mean() - 0.8) < 0.01
    assert not np.array_equal(keep, dropout_ref.keep_mask(124, (4, 32, 32, 24), 0.2))
    x = torch.full((4, 32, 32, 24), 2.0)
    y = dropout_ref.apply(x, keep, 0.2)
    assert set(np.unique(y.numpy()).tolist()) == {0.0, float(np.float32(2.0) / (np.float32(1) - np.float32(0.2)))}
    seen = []

    def hook(site, t):
        seen.append(site)
        return t
    p = model_ref.init_params("mobilenet_v2", num_classes=3)
    model_ref.retinanet_forward(p, torch.zeros(1, 64, 64, 3), 3, dropout=hook)
    assert len(seen) == 53 and len(set(seen)) == 53 and all(s.startswith("backbone.") and s.endswith(".dropout") for s in seen)


# ---- TensorFlow's own published unit-test vectors (tests/golden/reference_kats.py, "TF_*"): the [TF-sem] half of the oracle
def test_tf_published_nms_vectors():
    """NonMaxSuppressionOpTest.* -> utils_ref.nms_indices / nms_indices_vectorised (reference utils.py:213-220)."""
    for name, boxes, scores, max_out, want in K.TF_NMS_CASES:
        for fn in (utils_ref.nms_indices, utils_ref.nms_indices_vectorised):
            got = fn(boxes, scores, max_output_size=max_out, iou_threshold=0.5)
            assert list(got) == want, (name, fn.__name__, list(got))


def test_tf_published_resize_align_corners_vectors():
    """ResizeImagesTest.testResizeUpAlignCornersTrue -> the nearest-neighbour up-sample of the FPN (retinanet.py:153-155) and
    the bilinear rescale of the input pipeline (dataset.py:145-151)."""
    x = torch.from_numpy(K.TF_RESIZE_ALIGN_CORNERS_INPUT)
    oh, ow = K.TF_RESIZE_ALIGN_CORNERS_SIZE
    assert np.array_equal(tf_ops_ref.upsample_nearest_align_corners(x, oh, ow).numpy(), K.TF_RESIZE_ALIGN_CORNERS_NEAREST)
    assert list(tf_ops_ref.nn_resize_index(5, 3)) == [0, 1, 1, 2, 2] and list(tf_ops_ref.nn_resize_index(4, 2)) == [0, 0, 1, 1]
    got = dataset_ref.resize_bilinear_align_corners(K.TF_RESIZE_ALIGN_CORNERS_INPUT, oh, ow)
    assert np.allclose(np.asarray(got), K.TF_RESIZE_ALIGN_CORNERS_BILINEAR, rtol=0, atol=1e-6)


def test_tf_published_huber_vectors():
    """HuberLossTest.* (delta 1) -> losses_ref.regression_loss with every row foreground (losses.py:144-152)."""
    for name, labels, preds, want in K.TF_HUBER_CASES:
        lab = torch.from_numpy(np.atleast_2d(labels))
        pre = torch.from_numpy(np.atleast_2d(preds))
        fg = torch.ones(lab.shape[0], dtype=torch.bool)
        got = float(losses_ref.regression_loss(lab, pre, fg))
        assert abs(got - want) <= 1e-6, (name, got, want)


def test_tf_published_sigmoid_cross_entropy_vectors():
    """SigmoidCrossEntropyLossTest.testAllCorrectSigmoid / testAllWrongSigmoid -> losses_ref.sigmoid_bce_with_logits (losses.py:124)."""
    z = torch.from_numpy(K.TF_BCE_LOGITS)
    for name, labels, want in K.TF_BCE_CASES:
        got = float(losses_ref.sigmoid_bce_with_logits(torch.from_numpy(labels), z).mean())
        assert abs(got - want) <= 1e-3, (name, got, want)
