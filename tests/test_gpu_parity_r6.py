"""Parity at the operating points the earlier full-size tests left out (VERDICT r5 "next" item 4), all against the CPU oracle:

  (a) a whole-net train step at the reference's own default scale (train.py:94 `--scale 600`) on a NON-square image, 600 x 800:
      pyramid 75x100 / 38x50 / 19x25 / 10x13 / 5x7 -- odd maps, stride-2 convs on odd sizes, and the non-2x `align_corners`
      nearest-neighbour up-sample (retinanet.py:153-155: 19 -> 38, 13 -> 25, 7 -> 13 ...) INSIDE the FPN, forward and backward --
      MobileNetV2-FPN and ResNeXt-50-FPN;
  (b) BASELINE configs[3] AS BENCHMARKED: DenseNet-121-FPN 640 x 640, batch 4, dropout 0.2 (densenet.py:44,67,77,143), the masks the
      kernels draw injected into the oracle at its 119 sites;
  (c) the cfg-3 size (ResNeXt-50-FPN, 800 x 800, batch 2) on a CONDITIONED net -- trained by the product's own loop at that size --
      judged against the oracle in fp32 and fp64: measured, the trained net is MORE sensitive to ReLU / max-pool decisions than the
      random-init one (the fp32 oracle is a median 1.2e-1 from its own fp64 evaluation), and the product is ~100 x closer to the
      fp64 gradient than the fp32 oracle is (the random-init variant in test_gpu_fullsize.py stays as the stress case).
Runs on the MI355X box; the oracle legs take tens of seconds each on its host cores."""
import numpy as np
import pytest
import torch

from helpers import assert_close, coco_like_objects, dropout_sites, load_oracle_params, to_oracle_name
from oracle import dataset_ref, model_ref, train_ref
from test_gpu_fullsize import LEVELS, _oracle_losses_and_grads, _randomize_norms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pair_batch(rng, hw, pairs, classes=80):
    h, w = hw
    imgs = []
    boxes, cids, nobj = np.zeros((pairs, 32, 4), np.float32), np.zeros((pairs, 32), np.int32), np.zeros(pairs, np.int32)
    for i in range(pairs):
        im = rng.standard_normal((h, w, 3)).astype(np.float32)
        imgs += [im, im[:, ::-1].copy()]
        b, k = coco_like_objects(rng, min(h, w))
        nobj[i] = len(b)
        boxes[i, :len(b)], cids[i, :len(b)] = b, k % classes
    return torch.from_numpy(np.stack(imgs)), boxes, cids, nobj


def _grad_errors(grads_hip, gref):
    scale = max(float(np.abs(v).max()) for v in gref.values())

    def err(a, ref):
        return float(np.abs(a - ref).max()) / max(float(np.abs(ref).max()), 1e-3 * scale)

    return sorted(((err(grads_hip[n], gref[n]), n) for n in gref), reverse=True), err


def _product_step(dev, net, lv, image, boxes, cids, nobj, hw, classes, mode="focal"):
    """Labels from the product's own assignment (flip pair), one forward + focal / smooth-L1 loss + backward on the device."""
    import dataset, train
    pc, pr, pm = dataset.build_labels(hw, torch.from_numpy(cids).to(dev), torch.from_numpy(boxes).to(dev), lv, classes,
                                      num_obj=torch.from_numpy(nobj).to(dev), flip_pair=True)
    feats = {"image": image.to(dev), "detection": {"classifications": pc, "regressions": pr}, "trainable_masks": pm}
    trainer = train.Trainer(net, lv, optimizer="momentum", learning_rate=1e-2, loss_mode=mode, device=dev)
    cl, rl = trainer.forward_backward(feats)
    grads = {n: p.grad.detach().cpu().double().numpy() for n, p in net.named_parameters()}
    return cl.item(), rl.item(), grads, (pc, pr, pm)


@pytest.mark.parametrize("backbone", ["mobilenet_v2", "resnet_50"])
def test_scale600_non_square_train_step_matches_oracle(dev, backbone):
    """(a): 600 x 800, batch [image, hflip(image)], 80 classes.  Assignment maps bit-exact against the oracle's own assignment
    (both slots); losses <= 1e-4; every parameter gradient <= 5e-4 of max(|gradient|, 1e-3 x the largest gradient) -- for ResNeXt
    (ReLU gates, a max pool, random init) a tensor that misses the fp32 oracle is judged against the fp64 oracle as in
    test_gpu_fullsize.py: not further from the fp64 gradient than 3 x the fp32 oracle itself; MobileNetV2 (ELU) takes no clause."""
    import layers, levels as levels_mod, retinanet
    hw, classes = (600, 800), 80
    rng = np.random.default_rng(600 + len(backbone))
    lv = levels_mod.build_levels()
    torch.manual_seed(61)
    net = retinanet.RetinaNet(backbone, lv, classes, layers.elu, 0.0)
    _randomize_norms(net, 62)
    image, boxes, cids, nobj = _pair_batch(rng, hw, 1)
    weights = {k: v.detach().clone() for k, v in net.named_parameters()}
    net.to(dev)
    cl, rl, grads_hip, (pc, pr, pm) = _product_step(dev, net, lv, image, boxes, cids, nobj, hw, classes)
    sizes = [tuple(pc[k].shape[1:3]) for k in LEVELS]
    assert sizes == [(75, 100), (38, 50), (19, 25), (10, 13), (5, 7)], sizes
    # the oracle's own assignment of the sample and its mirror (dataset.py:126-142, augmentation.py:5-22): bit-exact maps
    o = int(nobj[0])
    c, r, m = dataset_ref.build_labels(hw, cids[0, :o], boxes[0, :o], classes)
    fc, fr, fm, _ = dataset_ref.flip(c, r, m)
    for k in LEVELS:
        for slot, (oc, orr, om) in enumerate(((c[k], r[k], m[k]), (fc[k], fr[k], fm[k]))):
            assert np.array_equal(pm[k][slot].cpu().numpy().astype(bool), om), "trainable mask %s[%d]" % (k, slot)
            assert np.array_equal(pc[k][slot].cpu().numpy(), oc), "class map %s[%d]" % (k, slot)
            assert_close(pr[k][slot].cpu().numpy(), orr, 1e-6, "regression targets %s[%d]" % (k, slot))
    masks = {k: pm[k].cpu().bool() for k in LEVELS}
    lab_c, lab_r = {k: pc[k].cpu() for k in LEVELS}, {k: pr[k].cpu() for k in LEVELS}
    if backbone == "mobilenet_v2":
        leaves = {to_oracle_name(k): v.clone().requires_grad_(True) for k, v in weights.items()}
        labels = {"classifications": lab_c, "regressions": lab_r, "trainable_masks": masks}
        _, ocl_t, orl_t, _ = train_ref.total_loss(leaves, image, labels, classes, "focal")
        g = torch.autograd.grad(ocl_t + orl_t, list(leaves.values()))
        ocl, orl = float(ocl_t), float(orl_t)
        inv = {to_oracle_name(k): k for k in weights}
        g32 = {inv[n]: t.double().numpy() for n, t in zip(leaves.keys(), g)}
    else:
        ocl, orl, g32 = _oracle_losses_and_grads(backbone, weights, image, lab_c, lab_r, masks, classes, torch.float32)
    assert_close(cl, ocl, 1e-4, backbone + " class loss (focal)")
    assert_close(rl, orl, 1e-4, backbone + " regression loss (smooth-L1)")
    errs, err = _grad_errors(grads_hip, g32)
    loose = [(e, n) for e, n in errs if e > 5e-4]
    note = ""
    if loose:
        assert backbone != "mobilenet_v2", "MobileNetV2-FPN: %d gradients off: %s" % (len(loose), ", ".join("%s %.2e" % (n, e) for e, n in loose[:8]))
        _, _, g64 = _oracle_losses_and_grads(backbone, weights, image, lab_c, lab_r, masks, classes, torch.float64)
        bad = [(n, e, err(grads_hip[n], g64[n]), err(g32[n], g64[n])) for e, n in loose
               if err(grads_hip[n], g64[n]) > max(5e-4, 3.0 * err(g32[n], g64[n]))]
        assert not bad, "%s: %d gradients off: %s" % (backbone, len(bad), "; ".join("%s %.2e / %.2e / %.2e" % b for b in bad[:8]))
        ratios = np.array([err(grads_hip[n], g64[n]) / max(err(g32[n], g64[n]), 1e-9) for _, n in loose])
        geo = float(np.exp(np.log(np.maximum(ratios, 1e-9)).mean()))
        assert geo <= 1.5, "%s: on average %.2f x further from the fp64 gradient than the fp32 oracle is" % (backbone, geo)
        note = "; %d of %d tensors judged against the fp64 oracle (geometric-mean distance ratio %.2f)" % (len(loose), len(errs), geo)
    print("%s 600x800 (pyramid 75x100 .. 5x7): class loss %.6f (oracle %.6f), regr loss %.6f (oracle %.6f), worst gradient error vs "
          "the fp32 oracle among the other tensors %.2e%s" % (backbone, cl, ocl, rl, orl, max([e for e, _ in errs if e <= 5e-4] or [0.0]), note))


def test_cfg4_as_benchmarked_dropout_02_matches_oracle(dev):
    """(b): DenseNet-121-FPN 640 x 640, batch 4 ([image, hflip] x 2), 80 classes, dropout 0.2 -- the step bench.py times as cfg 4 --
    with the product's counter-based masks (step counter 0) injected at the oracle's 119 dropout sites.  Losses 1e-4, every
    parameter gradient 5e-4 (a few may take the fp64 clause: one max pool switches)."""
    import layers, levels as levels_mod, retinanet
    backbone, hw, classes, rate = "densenet_121", (640, 640), 80, 0.2
    rng = np.random.default_rng(640)
    lv = levels_mod.build_levels()
    torch.manual_seed(41)
    net = retinanet.RetinaNet(backbone, lv, classes, layers.elu, rate)
    _randomize_norms(net, 42)
    image, boxes, cids, nobj = _pair_batch(rng, hw, 2)
    weights = {k: v.detach().clone() for k, v in net.named_parameters()}
    hook = dropout_sites(net, rate)
    assert len(hook.seeds) == 2 * (6 + 12 + 24 + 16) + 3
    net.to(dev)
    cl, rl, grads_hip, (pc, pr, pm) = _product_step(dev, net, lv, image, boxes, cids, nobj, hw, classes)
    masks = {k: pm[k].cpu().bool() for k in LEVELS}
    lab_c, lab_r = {k: pc[k].cpu() for k in LEVELS}, {k: pr[k].cpu() for k in LEVELS}
    ocl, orl, g32 = _oracle_losses_and_grads(backbone, weights, image, lab_c, lab_r, masks, classes, torch.float32, dropout=hook)
    assert sorted(hook.seen) == sorted(hook.seeds)
    assert_close(cl, ocl, 1e-4, "class loss (focal)")
    assert_close(rl, orl, 1e-4, "regression loss (smooth-L1)")
    errs, err = _grad_errors(grads_hip, g32)
    loose = [(e, n) for e, n in errs if e > 5e-4]
    if loose:
        assert len(loose) <= 8, "%d tensors miss 5e-4: %s" % (len(loose), ", ".join("%s %.2e" % (n, e) for e, n in loose[:8]))
        hook64 = dropout_sites(net, rate)
        _, _, g64 = _oracle_losses_and_grads(backbone, weights, image, lab_c, lab_r, masks, classes, torch.float64, dropout=hook64)
        bad = [(n, e) for e, n in loose if err(grads_hip[n], g64[n]) > max(5e-4, 3.0 * err(g32[n], g64[n]))]
        assert not bad, bad
    print("cfg 4 as benchmarked (640x640 x 4, dropout 0.2, 119 injected mask sites): class loss %.6f (oracle %.6f), regr loss %.6f "
          "(oracle %.6f), worst gradient error %.2e (%s), %d tensor(s) judged against the fp64 oracle"
          % (cl, ocl, rl, orl, errs[0][0], errs[0][1], len(loose)))


def test_cfg3_full_size_gradients_on_a_conditioned_net(dev):
    """(c): ResNeXt-50-FPN trained by the product's own loop AT the cfg-3 size -- DeviceFeed + one-graph step on the seeded shapes
    stream rendered at 800 x 800, batch [sample, hflip], BCE + dice + Huber (the reference's live loss, losses.py:124-152; focal
    collapses at this learning rate), momentum, lr 1e-2, 600 steps -- THEN one more 800 x 800 batch-2 step on a
    held-out sample against the composed oracle evaluated in fp32 AND in fp64.  Asserted: losses within 1e-5 of the fp64 oracle's;
    no gradient tensor further from the fp64 gradient than the fp32 oracle is, on geometric average >= 10 x closer; median / worst
    distance to the fp64 gradient <= 6e-3 / 6e-2 (see the comment at the asserts for what was measured and why the 5e-4 bar against
    the FP32 oracle cannot be the yardstick here)."""
    import dataset, layers, levels as levels_mod, retinanet, train
    from data_loaders.shapes import Shapes
    steps, hw, mode = 600, (800, 800), "bce_dice"
    lv = levels_mod.build_levels()
    loader = Shapes(None, image_size=hw, seed=0)
    classes = loader.num_classes
    torch.manual_seed(0)
    net = retinanet.RetinaNet('resnet_50', lv, classes, layers.elu, 0.0).to(dev)
    feed = dataset.DeviceFeed(loader, lv, scale=800, device=dev)
    tr = train.Trainer(net, lv, learning_rate=1e-2, loss_mode=mode, device=dev, use_graph=True, input_fn=feed)
    try:
        first = [tr.step()["class_loss"].item() for _ in range(20)]
        for _ in range(steps - 120):
            tr.step()
        last = [tr.step()["class_loss"].item() for _ in range(100)]
    finally:
        feed.close()
    tr.check_device_errors()
    # (every step is a new sample: single losses scatter between 0.6 and 1.0 late in this short run, and any change of a kernel's
    # summation order changes the trajectory -- the check is on the late AVERAGE, and only says "the weights have moved")
    assert float(np.mean(last)) < 0.95 * float(np.mean(first)), (float(np.mean(last)), float(np.mean(first)))
    del tr
    torch.cuda.empty_cache()
    weights = {k: v.detach().cpu().clone() for k, v in net.named_parameters()}
    for p in net.parameters():                                   # (the trainer's arena is gone: fresh gradient slots)
        p.grad = None
    sample = next(dataset.build_dataset(Shapes(None, image_size=hw, seed=777), lv, scale=800, device=dev, normalize=True))
    image = sample['image'].cpu()
    assert tuple(image.shape) == (2, 800, 800, 3)
    o = len(sample['boxes'])
    boxes, cids, nobj = np.zeros((1, 32, 4), np.float32), np.zeros((1, 32), np.int32), np.array([o], np.int32)
    boxes[0, :o], cids[0, :o] = np.asarray(sample['boxes'], np.float32), np.asarray(sample['class_ids'], np.int32)
    cl, rl, grads_hip, (pc, pr, pm) = _product_step(dev, net, lv, image, boxes, cids, nobj, hw, classes, mode=mode)
    masks = {k: pm[k].cpu().bool() for k in LEVELS}
    lab_c, lab_r = {k: pc[k].cpu() for k in LEVELS}, {k: pr[k].cpu() for k in LEVELS}
    ocl, orl, g32 = _oracle_losses_and_grads('resnet_50', weights, image, lab_c, lab_r, masks, classes, torch.float32, mode=mode)
    ocl64, orl64, g64 = _oracle_losses_and_grads('resnet_50', weights, image, lab_c, lab_r, masks, classes, torch.float64, mode=mode)

    def loss_ok(got, o32, o64, what):
        """1e-4 against the fp32 oracle -- or, where two fp32 evaluations of this net themselves differ by more than that (measured
        on the trained net: the regression loss, a mean over a few dozen foreground anchors, 5e-4), not further from the fp64 oracle
        than 3 x the fp32 oracle is."""
        e32, e64, o = abs(got - o32) / abs(o32), abs(got - o64) / abs(o64), abs(o32 - o64) / abs(o64)
        assert e32 <= 1e-4 or e64 <= max(1e-4, 3.0 * o), "%s: %.3e from the fp32 oracle, %.3e from the fp64 oracle (fp32 oracle: %.3e)" % (what, e32, e64, o)
        return "%.1e / %.1e (fp32 oracle %.1e)" % (e32, e64, o)

    n1 = loss_ok(cl, ocl, ocl64, "class loss (BCE + dice)")
    n2 = loss_ok(rl, orl, orl64, "regression loss (smooth-L1)")
    errs, err = _grad_errors(grads_hip, g32)
    loose = [(e, n) for e, n in errs if e > 5e-4]
    frac = 1.0 - len(loose) / len(errs)
    bad = [(n, e) for e, n in loose if err(grads_hip[n], g64[n]) > max(5e-4, 3.0 * err(g32[n], g64[n]))]
    worst64 = max(err(g32[n], g64[n]) for n in g32)
    ratios = np.array([err(grads_hip[n], g64[n]) / max(err(g32[n], g64[n]), 1e-9) for n in g32])
    geo = float(np.exp(np.log(np.maximum(ratios, 1e-9)).mean()))
    e64 = sorted((err(grads_hip[n], g64[n]), n) for n in g64)
    in64 = sum(1 for e, _ in e64 if e <= 5e-4)
    med64, worst = e64[len(e64) // 2][0], e64[-1]
    print("cfg-3 size on a conditioned ResNeXt-50-FPN (600 steps at 800 x 800): class loss %.6f (oracle %.6f; distance to the fp32 / fp64 oracle "
          "%s), regr loss %.6f (oracle %.6f; %s); against the fp32 oracle %d of %d tensors (%.1f %%) inside 5e-4 (fp32 vs fp64 ORACLE: median %.1e, "
          "up to %.1e); against the fp64 oracle directly: %d inside 5e-4, median %.1e, 90th percentile %.1e, worst %.1e (%s); product / fp32-oracle "
          "distance to the fp64 gradient: geometric mean %.3f, max %.2f"
          % (cl, ocl, n1, rl, orl, n2, len(errs) - len(loose), len(errs), 100 * frac, float(np.median([err(g32[n], g64[n]) for n in g32])), worst64,
             in64, med64, e64[int(0.9 * len(e64))][0], worst[0], worst[1], geo, float(ratios.max())))
    # MEASURED (round 6, twice): the hypothesis behind this test -- "on a conditioned net two fp32 evaluations rarely take different ReLU /
    # max-pool branches" -- is FALSE at this size: training sharpens the net, and the fp32 CPU oracle is then a median 1.2e-1 (up to
    # 3.4e-1) away from its own fp64 evaluation, so "5e-4 against the fp32 oracle" holds for 4 of 208 tensors.  The product sits a
    # median 2.1e-3 (90th percentile 4.7e-3, worst 2.1e-2) from the fp64 gradient -- 60 x closer than the fp32 oracle on median,
    # 100 x on geometric average -- and its losses 5e-8 / 4e-7 from the fp64 losses (the fp32 oracle's: 9e-5 / 4e-4).  The bars
    # below are those measurements x ~3; what they protect is "the product is far MORE accurate than an fp32 reference evaluation".
    assert not bad, bad
    assert abs(cl - ocl64) <= 1e-5 * abs(ocl64) and abs(rl - orl64) <= 1e-5 * abs(orl64), (cl, ocl64, rl, orl64)
    assert geo <= 0.1 and float(ratios.max()) <= 1.0, (geo, float(ratios.max()))
    assert med64 <= 6e-3 and worst[0] <= 6e-2, (med64, worst)
