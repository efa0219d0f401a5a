"""End-to-end parity of the RetinaNet hot path (model -> loss -> gradients -> optimizer step)
with the CPU oracle, through the reference-shaped Python API.  Runs on the MI355X box."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, load_oracle_params, to_oracle_name
from oracle import dataset_ref, model_ref, train_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4
LEVELS = ("P3", "P4", "P5", "P6", "P7")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _tiny_problem(dev, size=64, classes=3, seed=0):
    import layers, levels, retinanet
    rng = np.random.default_rng(seed)
    params = model_ref.init_params("mobilenet_v2", num_classes=classes, seed=seed)
    # non-trivial gamma/beta so their gradients are exercised
    g = torch.Generator().manual_seed(seed + 1)
    for k in params:
        if k.endswith(".gamma"):
            params[k] = 1 + 0.2 * torch.randn(params[k].shape, generator=g)
        elif k.endswith(".beta"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    net = retinanet.RetinaNet('mobilenet_v2', levels.build_levels(), classes, layers.elu, 0.0).to(dev)
    load_oracle_params(net, params)
    image = rng.standard_normal((2, size, size, 3)).astype(np.float32)
    boxes = np.array([[0.1, 0.15, 0.6, 0.7], [0.5, 0.4, 0.95, 0.9], [0.05, 0.5, 0.4, 0.98]], dtype=np.float32)
    cids = np.array([0, 2, 1])
    cls, reg, msk = dataset_ref.build_labels((size, size), cids, boxes, classes)
    fc, fr, fm, _ = dataset_ref.flip(cls, reg, msk)
    labels = {"classifications": {}, "regressions": {}, "trainable_masks": {}}
    for k in cls:
        labels["classifications"][k] = torch.from_numpy(np.stack([cls[k], fc[k]]))
        labels["regressions"][k] = torch.from_numpy(np.stack([reg[k], fr[k]]))
        labels["trainable_masks"][k] = torch.from_numpy(np.stack([msk[k], fm[k]]))
    return net, params, torch.from_numpy(image), labels


def _features(image, labels, dev):
    return {
        "image": image.to(dev),
        "detection": {"classifications": {k: v.to(dev) for k, v in labels["classifications"].items()},
                      "regressions": {k: v.to(dev) for k, v in labels["regressions"].items()}},
        "trainable_masks": {k: v.to(torch.uint8).to(dev) for k, v in labels["trainable_masks"].items()},
    }


def test_forward_matches_oracle_and_golden(dev, golden_dir):
    import layers, levels, retinanet
    fx = np.load(os.path.join(golden_dir, "oracle_e2e_tiny.npz"))
    params = model_ref.init_params("mobilenet_v2", num_classes=3, seed=0)
    net = retinanet.RetinaNet('mobilenet_v2', levels.build_levels(), 3, layers.elu, 0.2).to(dev)
    load_oracle_params(net, params)
    image = torch.from_numpy(fx["image"])
    with torch.no_grad():
        out = net(image.to(dev), training=False)
        ref = model_ref.retinanet_forward(params, image, 3)
    assert list(out["classifications"].keys()) == list(LEVELS)
    for k in LEVELS:
        assert out["classifications"][k].shape == ref["classifications"][k].shape
        assert_close(out["classifications"][k].cpu().numpy(), ref["classifications"][k].numpy(), TOL, "cls " + k)
        assert_close(out["regressions"][k].cpu().numpy(), ref["regressions"][k].numpy(), TOL, "reg " + k)
    assert_close(out["classifications"]["P5"].cpu().numpy(), fx["cls_P5"], TOL, "golden cls P5")
    assert_close(out["regressions"]["P7"].cpu().numpy(), fx["reg_P7"], TOL, "golden reg P7")


@pytest.mark.parametrize("mode", ["bce_dice", "focal"])
def test_loss_and_gradients_match_oracle(dev, mode):
    import levels as levels_mod, losses, utils
    net, params, image, labels = _tiny_problem(dev)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    tot, cl, rl, reg = train_ref.total_loss(leaves, image, labels, 3, mode)
    grads = dict(zip(leaves.keys(), torch.autograd.grad(cl + rl, list(leaves.values()))))

    feats = _features(image, labels, dev)
    lv = levels_mod.build_levels()
    logits = {"detection": net(feats["image"], training=True)}
    inp, logits = utils.process_labels_and_logits(labels=feats, logits=logits, levels=lv)
    gcl, grl = losses.loss(labels=inp["detection_trainable"], logits=logits["detection_trainable"], mode=mode)
    (gcl + grl).backward()
    # `regression_postprocessed` (only summaries / metrics read it) is decoded lazily, per level, on first access
    rp = logits["detection"].regression_postprocessed
    assert list(rp.keys()) == list(lv) and len(dict.keys(rp)) == 0
    p3 = rp["P3"]
    size = tuple(feats["image"].shape[1:3])
    want = utils.regression_postprocess(logits["detection"].regression["P3"], lv["P3"].normalized_anchor_sizes(size, utils.ANCHOR_SIZE_MODE))
    assert torch.equal(p3, want) and list(dict.keys(rp)) == ["P3"] and len(rp.items()) == len(list(lv))
    # Classification.prob = sigmoid(unscaled) (reference utils.py:245-247), also on first access only
    cls = logits["detection"].classification
    assert len(dict.keys(cls.prob)) == 0
    assert_close(cls.prob["P4"].cpu().numpy(), torch.sigmoid(cls.unscaled["P4"].detach()).cpu().numpy(), 1e-6, "prob")
    assert_close(gcl.item(), cl.item(), TOL, "class loss")
    assert_close(grl.item(), rl.item(), TOL, "regr loss")
    # Error is measured against max(|grad of this tensor|, 1e-3 * largest gradient in the net): some
    # gradients are analytically zero (a per-channel shift feeding a 1x1 conv + per-channel GroupNorm,
    # e.g. bottleneck_7_1.linear_conv.norm.beta) and hold only rounding noise on both sides.
    scale = max(float(g.abs().max()) for g in grads.values())
    worst = 0.0
    for name, p in net.named_parameters():
        o = to_oracle_name(name)
        ref = grads[o].numpy()
        err = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-3 * scale)
        assert err <= 5e-4, "grad %s: relative error %.3e" % (o, err)
        worst = max(worst, err)
    print("worst gradient error", worst)


def test_adam_under_graph_replay_equals_eager(dev):
    """Adam's bias-corrected learning rate changes every step; the optimizer kernel is launched per step outside the
    captured segments, so three steps through the replayed hipGraphs give the weights of three eager steps (the update
    rule itself is checked against the oracle in test_gpu_ops.py::test_optimizer_matches_tf_semantics)."""
    import levels as levels_mod, train
    weights = []
    for use_graph in (False, True):
        net, params, image, labels = _tiny_problem(dev, seed=3)
        trainer = train.Trainer(net, levels_mod.build_levels(), optimizer="adam", learning_rate=1e-3, grad_clip_norm=5.0,
                                loss_mode="bce_dice", device=dev, use_graph=use_graph)
        feats = _features(image, labels, dev)
        for _ in range(3):
            trainer.step(feats)
        torch.cuda.synchronize()
        weights.append(trainer.arena.weights.clone())
    assert torch.equal(weights[0], weights[1])


@pytest.mark.parametrize("use_graph,optimizer,lr", [(False, "momentum", 1e-2), (True, "momentum", 1e-2)])
def test_train_steps_match_oracle(dev, use_graph, optimizer, lr):
    """Three optimizer steps (momentum, L2 regulariser, global-norm clip) == the oracle's (reference train.py:111-134),
    eager and through the replayed hipGraph segments."""
    import levels as levels_mod, train
    net, params, image, labels = _tiny_problem(dev, seed=3)
    trainer = train.Trainer(net, levels_mod.build_levels(), optimizer=optimizer, learning_rate=lr,
                            grad_clip_norm=5.0, loss_mode="bce_dice", device=dev, use_graph=use_graph)
    feats = _features(image, labels, dev)
    state = {}
    for step in range(1, 4):
        out = trainer.step(feats)
        first, _ = train_ref.train_step(params, image, labels, 3, state, lr=lr, optimizer=optimizer, step=step,
                                        loss_mode="bce_dice", grad_clip_norm=5.0)
        torch.cuda.synchronize()
        assert_close(out["class_loss"].item(), first[1], 2e-4, "class loss step %d" % step)
        assert_close(out["regr_loss"].item(), first[2], 2e-4, "regr loss step %d" % step)
        assert_close(out["regularization_loss"].item(), first[3], 2e-4, "reg loss step %d" % step)
    worst = 0.0
    for name, p in net.named_parameters():
        worst = max(worst, assert_close(p.detach().cpu().numpy(), params[to_oracle_name(name)].numpy(), 5e-4, name))
    print("worst weight error after 3 steps", worst)


def test_fused_head_towers_equal_separate_subnets(dev):
    """retinanet.FUSE_HEAD_TOWERS (one 512-channel tower for both subnets) gives the same outputs."""
    import retinanet
    net, params, image, labels = _tiny_problem(dev, seed=7)
    with torch.no_grad():
        a = net(image.to(dev), training=False)
        retinanet.FUSE_HEAD_TOWERS = True
        try:
            b = net(image.to(dev), training=False)
        finally:
            retinanet.FUSE_HEAD_TOWERS = False
    for k in LEVELS:
        # (the default path folds the GroupNorms into the Winograd transforms: statistics from chunk rows, box output conv as
        # a Winograd layer -- rounding differs from the stand-alone kernels by a few 1e-6 of the output range)
        assert_close(b["classifications"][k].cpu().numpy(), a["classifications"][k].cpu().numpy(), 5e-5, "cls " + k)
        assert_close(b["regressions"][k].cpu().numpy(), a["regressions"][k].cpu().numpy(), 5e-5, "reg " + k)


def test_gradient_average_equals_bigger_batch(dev):
    """MirroredStrategy semantics: mean of two replicas' gradients == what the
    GradientAllReduce + grad_scale path applies; checked single-process by accumulation."""
    import levels as levels_mod, train
    net, params, image, labels = _tiny_problem(dev, seed=5)
    trainer = train.Trainer(net, levels_mod.build_levels(), device=dev)
    feats = _features(image, labels, dev)
    feats2 = _features(torch.flip(image, [0]), {k: {kk: torch.flip(v, [0]) for kk, v in d.items()}
                                                for k, d in labels.items()}, dev)
    trainer.forward_backward(feats)
    g1 = trainer.arena.grads.clone()
    trainer.forward_backward(feats2)
    g2 = trainer.arena.grads.clone()
    trainer.arena.grads.copy_(g1 + g2)          # what an all-reduce(sum) over 2 ranks leaves behind
    w0 = trainer.arena.weights.clone()
    trainer.opt.step(grad_scale=0.5)
    expect = w0 - 1e-2 * (0.5 * (g1 + g2) + trainer.arena.wd_per_block.repeat_interleave(train.OPT_BLOCK) * w0)
    assert_close(trainer.arena.weights.cpu().numpy(), expect.cpu().numpy(), 1e-6, "averaged step")


def test_no_group_norm_barrier_timeouts(dev):
    """The grid-resident GroupNorm kernels meet at a bounded in-kernel barrier; after everything this test module ran
    (eager, hipGraph, two head streams) no barrier may have given up."""
    import _rn
    assert _rn.barrier_timeouts() == 0


def test_plain_autograd_after_a_trainer_exists(dev):
    """The trainer's backward cut (detached leaves at the backbone taps) is installed only while one of ITS segments runs:
    a plain `net(x, training=True)` + backward on the same net -- before, between and after trainer steps, and with a
    second trainer built on it -- still reaches the backbone."""
    import losses, train, utils, levels as levels_mod
    net, params, image, labels = _tiny_problem(dev)
    lv = levels_mod.build_levels()
    feats = _features(image, labels, dev)

    def plain_backward():
        for p in net.parameters():
            p.grad = None
        out = {'detection': net(feats['image'], training=True)}
        inp, logits = utils.process_labels_and_logits(labels=feats, logits=out, levels=lv)
        cl, rl = losses.loss(labels=inp['detection_trainable'], logits=logits['detection_trainable'], mode='focal')
        (cl + rl).backward()
        g = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        assert all(n in g for n, _ in net.named_parameters()), "a parameter got no gradient"
        return g

    before = plain_backward()
    assert float(before['base.backbone.input_conv._mods.0.weight'].abs().max()) > 0
    tr1 = train.Trainer(net, lv, loss_mode='focal', device=dev)
    assert net.base.backward_cut is None
    tr1.forward_backward(feats)
    with_trainer = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    tr2 = train.Trainer(net, lv, loss_mode='focal', device=dev)        # re-points the parameters into ITS arena
    tr2.forward_backward(feats)
    after = plain_backward()
    for n in before:
        assert_close(after[n].cpu().numpy(), before[n].cpu().numpy(), 1e-5, "plain backward after trainers: " + n)
        assert_close(with_trainer[n].cpu().numpy(), before[n].cpu().numpy(), 1e-5, "trainer segments vs plain backward: " + n)


@pytest.mark.parametrize("backbone,use_graph", [("resnet_50", False), ("resnet_50", True), ("densenet_121", False)])
def test_stage_cut_backward_parts_bit_equal_to_single_segment(dev, backbone, use_graph):
    """ResNeXt / DenseNet: the backbone's backward pass in one part per stage (Trainer stage cuts: each stage's slice of the
    gradient arena can be all-reduced underneath the stages below it, reference train.py:261-267) changes no bit: same losses
    and weights after two steps as the single-segment step; the parts' arena slices tile [0, cut_offset) from the top down."""
    import dataset, layers, levels as levels_mod, retinanet, train
    lv = levels_mod.build_levels()

    def build():        # (twice from the same seeds; copy.deepcopy would drop the kernels' l2_scale attribute)
        layers.Dropout._next_seed[0] = 0x5EED
        torch.manual_seed(3)
        return retinanet.RetinaNet(backbone, lv, 4, layers.elu, 0.1).to(dev)

    net_a, net_b = build(), build()
    rng = np.random.default_rng(1)
    size = 96
    image = torch.from_numpy(rng.standard_normal((2, size, size, 3)).astype(np.float32)).to(dev)
    boxes = torch.tensor([[[0.1, 0.2, 0.7, 0.8], [0.4, 0.1, 0.9, 0.5]]], device=dev)
    cids = torch.tensor([[1, 3]], dtype=torch.int32, device=dev)
    c, r, m = dataset.build_labels((size, size), cids, boxes, lv, 4, flip_pair=True)
    feats = {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}
    ta = train.Trainer(net_a, lv, loss_mode="focal", device=dev, use_graph=use_graph)
    tb = train.Trainer(net_b, lv, loss_mode="focal", device=dev, use_graph=use_graph, overlap=False)
    assert ta._stage_bb is not None and tb.cut_offset == 0
    # (`ta` keeps the segments several ranks replay -- their schedule is what is checked here; `tb`, one rank's default, captures
    # the whole step, update included, as ONE graph: the bit-equal weights below cover that path too)
    ta.whole_step_graph = False
    seen = []
    orig = ta.allreduce.launch
    ta.allreduce.launch = lambda start=0, end=None: (seen.append((start, ta.arena.count if end is None else end)), orig(start, end))[1]
    for _ in range(2):
        del seen[:]
        oa, ob = ta.step(feats), tb.step(feats)
        assert oa["class_loss"].item() == ob["class_loss"].item() and oa["regr_loss"].item() == ob["regr_loss"].item()
        # heads + FPN first, then the backbone's parts from the last stage down to the stem, tiling the arena exactly
        assert seen[0] == (ta.cut_offset, ta.arena.count) and len(seen) >= 4, seen
        assert all(a[0] == b[1] for a, b in zip(seen, seen[1:])) and seen[-1][0] == 0, seen
    assert torch.equal(ta.arena.weights, tb.arena.weights)


def test_twenty_step_loss_curve_matches_oracle(dev):
    """Loss-CURVE parity: twenty momentum steps at 64x64 on a stream of different batches, the product
    through its replayed hipGraph segments, the oracle through train_ref.train_step -- every step's class loss, regression
    loss and regulariser within 1e-3 relative (the bar loosens from the one-step 1e-4 because twenty updates compound
    rounding differences through the weights), the weights after twenty steps within 2e-3 of their range."""
    import levels as levels_mod, train
    net, params, image0, labels0 = _tiny_problem(dev, seed=5)
    rng = np.random.default_rng(55)
    box_sets = [np.array([[0.1, 0.15, 0.6, 0.7], [0.5, 0.4, 0.95, 0.9], [0.05, 0.5, 0.4, 0.98]], np.float32),
                np.array([[0.2, 0.2, 0.8, 0.8]], np.float32),
                np.array([[0.0, 0.0, 0.5, 0.45], [0.45, 0.5, 1.0, 1.0]], np.float32),
                np.array([[0.3, 0.1, 0.7, 0.5], [0.1, 0.55, 0.35, 0.95], [0.6, 0.6, 0.9, 0.85], [0.05, 0.05, 0.3, 0.3]], np.float32)]
    batches = []
    for b in box_sets:
        cids = rng.integers(0, 3, len(b))
        cls, reg, msk = dataset_ref.build_labels((64, 64), cids, b, 3)
        fc, fr, fm, _ = dataset_ref.flip(cls, reg, msk)
        img = rng.standard_normal((1, 64, 64, 3)).astype(np.float32)
        image = torch.from_numpy(np.concatenate([img, img[:, :, ::-1]], 0).copy())
        labels = {"classifications": {k: torch.from_numpy(np.stack([cls[k], fc[k]])) for k in cls},
                  "regressions": {k: torch.from_numpy(np.stack([reg[k], fr[k]])) for k in cls},
                  "trainable_masks": {k: torch.from_numpy(np.stack([msk[k], fm[k]])) for k in cls}}
        batches.append((image, labels, _features(image, labels, dev)))
    trainer = train.Trainer(net, levels_mod.build_levels(), optimizer="momentum", learning_rate=1e-2, loss_mode="bce_dice",
                            device=dev, use_graph=True)
    state, worst, curve = {}, 0.0, []
    for step in range(1, 21):
        image, labels, feats = batches[(step - 1) % len(batches)]
        out = trainer.step(feats)
        first, _ = train_ref.train_step(params, image, labels, 3, state, lr=1e-2, optimizer="momentum", step=step,
                                        loss_mode="bce_dice")
        torch.cuda.synchronize()
        got = (out["class_loss"].item(), out["regr_loss"].item(), out["regularization_loss"].item())
        curve.append(got[:2])
        for g, w, what in zip(got, first[1:4], ("class", "regr", "regulariser")):
            worst = max(worst, assert_close(g, w, 1e-3, "%s loss at step %d" % (what, step)))
    assert curve[-4][0] < curve[0][0], curve       # same batch, 16 steps later: the class loss came down (product and oracle alike)
    for name, p in net.named_parameters():
        assert_close(p.detach().cpu().numpy(), params[to_oracle_name(name)].numpy(), 2e-3, name + " after 20 steps")
    print("worst loss error over 20 steps %.2e" % worst)


@pytest.mark.parametrize("use_graph", [False, True])
def test_deferred_tower_weight_gradients_change_no_bit(dev, use_graph, monkeypatch):
    """The head towers' weight-gradient halves run on a side stream beside the backbone's backward pass (Trainer.defer_wgrad,
    rn_conv3x3_winograd_gn_bwd_wgrad) instead of inside the heads' backward pass: the same kernels on the same operands, so
    losses and weights after three steps are bit-identical to the undeferred schedule, eager and under graph replay."""
    import dataset, layers, levels as levels_mod, retinanet, train
    lv = levels_mod.build_levels()

    def build(defer):
        monkeypatch.setenv("RN_DEFER_WGRAD", "1" if defer else "0")
        layers.Dropout._next_seed[0] = 0x5EED
        torch.manual_seed(4)
        net = retinanet.RetinaNet('mobilenet_v2', lv, 4, layers.elu, 0.2).to(dev)
        return train.Trainer(net, lv, loss_mode="focal", device=dev, use_graph=use_graph)

    ta, tb = build(True), build(False)
    assert ta.defer_wgrad and not tb.defer_wgrad and ta.cut_offset > 0
    rng = np.random.default_rng(2)
    size = 256
    image = torch.from_numpy(rng.standard_normal((2, size, size, 3)).astype(np.float32)).to(dev)
    boxes = torch.tensor([[[0.1, 0.2, 0.7, 0.8], [0.4, 0.1, 0.9, 0.5]]], device=dev)
    cids = torch.tensor([[1, 3]], dtype=torch.int32, device=dev)
    c, r, m = dataset.build_labels((size, size), cids, boxes, lv, 4, flip_pair=True)
    feats = {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}
    for _ in range(3):
        oa, ob = ta.step(feats), tb.step(feats)
        assert oa["class_loss"].item() == ob["class_loss"].item() and oa["regr_loss"].item() == ob["regr_loss"].item()
    torch.cuda.synchronize()
    assert torch.equal(ta.arena.weights, tb.arena.weights)


@pytest.mark.parametrize("use_graph", [False, True])
def test_kernel_transforms_ahead_of_the_layers_change_no_bit(dev, use_graph, monkeypatch):
    """RN_WINO_PRE=1 (opt-in: measured slower, DESIGN_EXPERIMENTS): the head towers' Winograd kernel transforms once per step on a side
    stream in front of the forward pass (rn_conv3x3_winograd_gn_weights -> rn_wino_gn.u_ready / urot_ready) instead of inside every
    layer's first launch: the same kernels on the same weights, so losses and weights after three steps are bit-identical."""
    import dataset, layers, levels as levels_mod, retinanet, train
    lv = levels_mod.build_levels()

    def build(pre):
        monkeypatch.setenv("RN_WINO_PRE", "1" if pre else "0")
        layers.Dropout._next_seed[0] = 0x5EED
        torch.manual_seed(4)
        net = retinanet.RetinaNet('mobilenet_v2', lv, 4, layers.elu, 0.2).to(dev)
        return train.Trainer(net, lv, loss_mode="focal", device=dev, use_graph=use_graph)

    ta, tb = build(True), build(False)
    assert ta.wino_pre and not tb.wino_pre
    rng = np.random.default_rng(2)
    size = 256
    image = torch.from_numpy(rng.standard_normal((2, size, size, 3)).astype(np.float32)).to(dev)
    boxes = torch.tensor([[[0.1, 0.2, 0.7, 0.8], [0.4, 0.1, 0.9, 0.5]]], device=dev)
    cids = torch.tensor([[1, 3]], dtype=torch.int32, device=dev)
    c, r, m = dataset.build_labels((size, size), cids, boxes, lv, 4, flip_pair=True)
    feats = {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}
    for _ in range(3):
        oa, ob = ta.step(feats), tb.step(feats)
        assert oa["class_loss"].item() == ob["class_loss"].item() and oa["regr_loss"].item() == ob["regr_loss"].item()
    torch.cuda.synchronize()
    assert ta._wino_pre is not None and len(ta._wino_pre.items) == 10 and tb._wino_pre is None
    assert torch.equal(ta.arena.weights, tb.arena.weights)


def test_whole_step_graph_equals_segments_and_eager(dev):
    """One rank: segment A, segment B and the optimizer's update captured as ONE graph (Trainer.whole_step_graph) against the
    segments several ranks replay (two graphs + the eagerly launched update) and against eager launches: the same kernels on the
    same operands -- losses, regulariser, weights, optimizer slots and the dropout counter bit-identical after four steps at
    dropout 0.2; a changed learning rate captures again instead of replaying the old constant."""
    import dataset, layers, levels as levels_mod, ops, retinanet, train
    lv = levels_mod.build_levels()

    def build(use_graph, whole):
        layers.Dropout._next_seed[0] = 0x5EED
        torch.manual_seed(4)
        net = retinanet.RetinaNet('mobilenet_v2', lv, 4, layers.elu, 0.2).to(dev)
        t = train.Trainer(net, lv, loss_mode="focal", device=dev, use_graph=use_graph)
        t.whole_step_graph = whole
        return t

    tw, ts, te = build(True, True), build(True, False), build(False, False)
    rng = np.random.default_rng(2)
    size = 256
    image = torch.from_numpy(rng.standard_normal((2, size, size, 3)).astype(np.float32)).to(dev)
    boxes = torch.tensor([[[0.1, 0.2, 0.7, 0.8], [0.4, 0.1, 0.9, 0.5]]], device=dev)
    cids = torch.tensor([[1, 3]], dtype=torch.int32, device=dev)
    c, r, m = dataset.build_labels((size, size), cids, boxes, lv, 4, flip_pair=True)
    feats = {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}
    for i in range(4):
        if i == 2:
            for t in (tw, ts, te):
                t.opt.lr = 0.5 * t.opt.lr
        ow, os_, oe = tw.step(feats), ts.step(feats), te.step(feats)
        for k in ("class_loss", "regr_loss", "regularization_loss"):
            assert ow[k].item() == os_[k].item() == oe[k].item(), (i, k, ow[k].item(), os_[k].item(), oe[k].item())
    torch.cuda.synchronize()
    assert tw._graphs[5] and not ts._graphs[5] and len(tw._graph_cache) == 2
    assert torch.equal(tw.arena.weights, ts.arena.weights) and torch.equal(tw.arena.weights, te.arena.weights)
    assert torch.equal(tw.opt.state1, ts.opt.state1) and torch.equal(tw.opt.state1, te.opt.state1)
    assert tw.drop_counter.item() == ts.drop_counter.item() == te.drop_counter.item() == 4 * ops.DROPOUT_COUNTER_STEP
    assert tw.opt.step_count == ts.opt.step_count == te.opt.step_count == 4 and tw.steps_done == 4
    # a learning-rate SCHEDULE (a new rate every step) must not capture a new graph set every step: after the third distinct rate the
    # update leaves the captured step (one graph per part + the eagerly launched update, whose rate is a launch argument) -- and the
    # results stay those of eager launches
    for i in range(5):
        for t in (tw, te):
            t.opt.lr = 0.9 * t.opt.lr
        ow, oe = tw.step(feats), te.step(feats)
        assert ow["class_loss"].item() == oe["class_loss"].item(), (i, ow["class_loss"].item(), oe["class_loss"].item())
    torch.cuda.synchronize()
    assert not tw._graphs[5] and tw._lr_changes >= 5
    caches = len(tw._graph_cache)
    for t in (tw, te):
        t.opt.lr = 0.9 * t.opt.lr
    tw.step(feats); te.step(feats)
    torch.cuda.synchronize()
    assert len(tw._graph_cache) == caches                       # (no further capture: the rate is no longer part of the key)
    assert torch.equal(tw.arena.weights, te.arena.weights)
