"""The reference's own known-answer vectors (tests/golden/reference_kats.py, each citing the reference test it is
transcribed from) pushed through the PRODUCT path -- the HIP kernels behind the reference-shaped Python API -- not only
through the oracle (tests/test_oracle_kats.py does that on CPU)."""
import numpy as np
import pytest
import torch

import reference_kats as K
from helpers import assert_close
from oracle import utils_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import _rn
    _rn.lib()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_iou_reference_kat_through_kernel(dev):
    """utils_test.py:99-118 -> [0.25, 0.25, 0, 0] (overlap, scale invariance, disjoint, degenerate)."""
    import utils
    got = utils.iou(_t(K.IOU_A, dev), _t(K.IOU_B, dev)).cpu().numpy()
    assert got.shape == (4,)
    assert np.allclose(got, K.IOU_EXPECTED, rtol=1e-6, atol=0), got
    assert got[2] == 0.0 and got[3] == 0.0


def test_iou_broadcast_shapes_bit_exact_vs_oracle(dev):
    """The call shape of dataset.py:57-60 ([O,1,1,1,4] x [1,H,W,A,4] -> [O,H,W,A]), equal shapes and a mixed broadcast;
    bit-exact against the oracle's float32 restatement."""
    import utils
    rng = np.random.default_rng(5)

    def boxes(*shape):
        c = rng.uniform(0.1, 0.9, shape + (2,))
        s = rng.uniform(0.0, 0.5, shape + (2,))
        return np.concatenate([c - s / 2, c + s / 2], -1).astype(np.float32)

    a, b = boxes(7, 1, 1, 1), boxes(1, 5, 6, 9)
    got = utils.iou(_t(a, dev), _t(b, dev)).cpu().numpy()
    exp = utils_ref.iou(a, b)
    assert got.shape == (7, 5, 6, 9) and np.array_equal(got, exp.astype(np.float32))
    a, b = boxes(33, 3), boxes(33, 3)
    assert np.array_equal(utils.iou(_t(a, dev), _t(b, dev)).cpu().numpy(), utils_ref.iou(a, b).astype(np.float32))
    a, b = boxes(4, 1, 3), boxes(1, 5, 3)                      # general broadcast (expanded on the host side)
    got = utils.iou(_t(a, dev), _t(b, dev)).cpu().numpy()
    assert got.shape == (4, 5, 3) and np.array_equal(got, utils_ref.iou(a, b).astype(np.float32))
    # empty
    assert utils.iou(_t(boxes(0), dev), _t(boxes(0), dev)).shape == (0,)
    # the reference's tf.assert_* (utils.py:65-68): malformed boxes raise
    bad = np.array([[0.5, 0.5, 0.4, 0.6]], np.float32)
    with pytest.raises(AssertionError):
        utils.iou(_t(bad, dev), _t(bad, dev))


def test_decode_transform_kats_through_kernel(dev):
    """utils_test.py:7-42 (cell centres added to the shifts), :76-97 (centre -> corner), retinanet_old_test.py:15-37
    (scale by the anchor) -- the three stages of utils.regression_postprocess -- each isolated by the choice of inputs
    to the ONE decode kernel the product has (rn_decode_boxes)."""
    import utils
    # (1) anchor-relative -> image-relative: anchors (1, 1), regression [dy, dx, log h, log w]
    r = K.ANCHOR_REL_INPUT.copy()
    r[..., 2:] = np.log(r[..., 2:])
    out = utils.regression_postprocess(_t(r, dev), np.ones((1, 2), np.float32)).cpu().numpy().astype(np.float64)
    centre = (out[..., :2] + out[..., 2:]) / 2
    size = out[..., 2:] - out[..., :2]
    assert np.allclose(np.concatenate([centre, size], -1), K.ANCHOR_REL_EXPECTED, atol=1e-6)
    # (2) centre -> corner on a 1x1 grid (cell centre 0.5): centre (0.5, 1.0), size (0.2, 0.4)
    c = K.CENTER_CORNER_INPUT[:, :1, :1]
    r = np.concatenate([c[..., :2] - 0.5, np.log(c[..., 2:])], -1).astype(np.float32)
    out = utils.regression_postprocess(_t(r, dev), np.ones((1, 2), np.float32)).cpu().numpy()
    assert np.allclose(out, K.CENTER_CORNER_EXPECTED[:, :1, :1], atol=1e-6)
    # (3) scale_regression: [shift, exp(log size)] * [ah, aw, ah, aw]
    r = K.SCALE_REGR_INPUT.copy()
    r[..., 2:] = np.log(r[..., 2:])
    out = utils.regression_postprocess(_t(r, dev), K.SCALE_REGR_ANCHORS).cpu().numpy().astype(np.float64)
    centre = (out[..., :2] + out[..., 2:]) / 2 - 0.5              # 1x1 grid: cell centre 0.5
    size = out[..., 2:] - out[..., :2]
    assert np.allclose(np.concatenate([centre, size], -1), K.SCALE_REGR_EXPECTED, atol=1e-6)
    # and the composition against the oracle, element-wise (boxes are the quantity north_star's 1e-4 is stated on)
    rng = np.random.default_rng(3)
    reg = (rng.standard_normal((2, 5, 7, 9, 4)) * 0.5).astype(np.float32)
    import levels
    anchors = levels.build_levels()["P5"].normalized_anchor_sizes((160, 224))
    got = utils.regression_postprocess(_t(reg, dev), anchors).cpu().numpy()
    assert_close(got, utils_ref.regression_postprocess(reg, anchors), 1e-5, "decode", elementwise_tol=1e-4)


def test_classmap_decode_kat_through_product(dev):
    """utils_test.py:120-138 in its current form (fg mask)."""
    import utils
    got = utils.classmap_decode(_t(K.CLASSMAP, dev)).fg_mask.cpu().numpy()
    assert np.array_equal(got, K.CLASSMAP_FG_EXPECTED)


# ---- TensorFlow's own published unit-test vectors (tests/golden/reference_kats.py "TF_*") through the kernels
@pytest.mark.parametrize("case", range(len(K.TF_NMS_CASES)), ids=[c[0] for c in K.TF_NMS_CASES])
def test_tf_published_nms_vectors_through_kernel(dev, case):
    """NonMaxSuppressionOpTest.* through rn_nms_classwise (one class): the indices TF's own test expects (utils.py:213-220)."""
    import utils
    name, boxes, scores, max_out, want = K.TF_NMS_CASES[case]
    cls = torch.zeros((boxes.shape[0],), dtype=torch.int64, device=dev)
    _, _, _, idx = utils._nms_arrays(_t(boxes, dev), _t(scores, dev), cls, 1, max_out)
    assert idx.cpu().tolist() == want, (name, idx.cpu().tolist())
    if boxes.shape[0]:
        got = utils.nms(utils.BoxesDecoded(_t(boxes, dev), _t(scores, dev), cls), max_output_size=max_out)
        assert np.array_equal(got.boxes.cpu().numpy(), boxes[want]) and np.array_equal(got.scores.cpu().numpy(), scores[want])


def test_tf_published_resize_vectors_through_kernels(dev):
    """ResizeImagesTest.testResizeUpAlignCornersTrue: the FPN's nearest-neighbour up-sample (rn_upsample_add with a zero lateral,
    retinanet.py:153-157) and the input pipeline's bilinear rescale (rn_resize_bilinear_normalize, dataset.py:145-151)."""
    import dataset
    import ops
    oh, ow = K.TF_RESIZE_ALIGN_CORNERS_SIZE
    for ch in (4, 8):
        top = np.repeat(K.TF_RESIZE_ALIGN_CORNERS_INPUT, ch, axis=3)
        got = ops.upsample_add(torch.zeros((1, oh, ow, ch), device=dev), _t(top, dev)).cpu().numpy()
        assert np.array_equal(got, np.repeat(K.TF_RESIZE_ALIGN_CORNERS_NEAREST, ch, axis=3))
    img = np.repeat(K.TF_RESIZE_ALIGN_CORNERS_INPUT, 3, axis=3)
    got = dataset.rescale_image(_t(img, dev), size=(oh, ow)).cpu().numpy()
    assert np.allclose(got, np.repeat(K.TF_RESIZE_ALIGN_CORNERS_BILINEAR, 3, axis=3), rtol=0, atol=1e-6)


def test_tf_published_huber_and_sigmoid_cross_entropy_vectors_through_loss_kernel(dev):
    """HuberLossTest.* (losses.py:144-152) and SigmoidCrossEntropyLossTest.testAllCorrect / testAllWrongSigmoid (losses.py:124)
    through rn_detection_loss (ops.detection_loss, mode bce_dice = the reference's live loss)."""
    import ops
    for name, labels, preds, want in K.TF_HUBER_CASES:
        lab, pre = np.atleast_2d(labels), np.atleast_2d(preds)
        m = lab.shape[0]
        one = np.ones((m, 1), np.float32)                      # one class, label 1 => every row is foreground
        _, reg, _ = ops.detection_loss([_t(np.zeros((m, 1), np.float32), dev)], [_t(pre, dev)], [_t(one, dev)], [_t(lab, dev)],
                                       [_t(np.ones(m, np.uint8), dev)], 1, "bce_dice")
        assert abs(reg.item() - want) <= 1e-6, (name, reg.item(), want)
    z = K.TF_BCE_LOGITS
    zero4 = np.zeros((3, 4), np.float32)
    for name, labels, want in K.TF_BCE_CASES:
        cl, _, _ = ops.detection_loss([_t(z, dev)], [_t(zero4, dev)], [_t(labels, dev)], [_t(zero4, dev)],
                                      [_t(np.ones(3, np.uint8), dev)], 3, "bce_dice")
        # class loss = mean BCE + mean_c dice_c (losses.py:124-139); dice_c = 1 - 2 sum(l p) / (sum l + sum p) with p = sigmoid(+-100)
        # in {0, 1}: 0 for every class when all are right, 1 when all are wrong
        dice = 0.0 if name == "testAllCorrectSigmoid" else 1.0
        assert abs(cl.item() - dice - want) <= 1e-3, (name, cl.item(), want)
