"""The minimal CLI driver (reference train.py flags) trains on the synthetic shapes loader, saves and resumes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_trains_and_resumes(tmp_path, capsys):
    import train
    exp = str(tmp_path / "exp")
    argv = ["--dataset", "shapes", "--epochs", "1", "--steps-per-epoch", "60", "--scale", "128", "--experiment", exp,
            "--backbone", "mobilenet_v2", "--optimizer", "momentum", "--dropout", "0.1"]
    assert train.main(argv) == 60
    out = capsys.readouterr().out
    losses = [float(l.split("class_loss ")[1].split()[0]) for l in out.splitlines() if "class_loss" in l]
    regr = [float(l.split("regr_loss ")[1].split()[0]) for l in out.splitlines() if "regr_loss" in l]
    assert len(losses) == 3 and losses[-1] < losses[0] and regr[-1] < regr[0]      # it learns
    assert os.path.exists(os.path.join(exp, "model.safetensors"))
    assert train.main(argv + ["--eval-images", "4"]) == 120                           # resumed from step 60
    out = capsys.readouterr().out
    assert "restored step 60" in out
    ev = [l for l in out.splitlines() if l.startswith("eval:")]
    assert len(ev) == 1 and "over 4 images" in ev[0]                                  # mAP / class_iou / regr_iou ran
    vals = dict(zip(ev[0].split()[1::2], ev[0].split()[2::2]))
    assert 0.0 <= float(vals["class_iou"]) <= 1.0


def _shapes_trainer(dev, use_graph, feed, dropout=0.2, seed=11, scale=128, loss="bce_dice", backbone='mobilenet_v2'):
    import dataset, layers, levels as levels_mod, retinanet, train
    from data_loaders.shapes import Shapes
    lv = levels_mod.build_levels()
    loader = Shapes(None, image_size=(scale + scale // 4, scale), seed=seed)
    torch.manual_seed(0)
    layers.Dropout._next_seed[0] = 0x5EED             # the same dropout streams for every net built in this process
    net = retinanet.RetinaNet(backbone, lv, loader.num_classes, layers.elu, dropout).to(dev)
    if feed:
        src = dataset.DeviceFeed(loader, lv, scale=scale, device=dev)
        tr = train.Trainer(net, lv, learning_rate=1e-2, loss_mode=loss, device=dev, use_graph=use_graph, input_fn=src)
        return net, tr, src, lv
    tr = train.Trainer(net, lv, learning_rate=1e-2, loss_mode=loss, device=dev, use_graph=use_graph)
    return net, tr, dataset.build_dataset(loader, lv, scale=scale, device=dev), lv


def test_graph_with_fresh_data_equals_eager_on_the_same_sample_stream():
    """train_input_fn (train.py:190-202, dataset.py:182-204: a NEW sample every step) on the hipGraph path: the loader
    thread -> pinned memory -> async upload -> static buffers -> rescale / flip / assignment INSIDE the captured segment
    (dataset.DeviceFeed) gives, after five steps with dropout 0.2, bit for bit the weights of five eager steps fed by
    dataset.build_dataset from the same loader seed -- and the eager feed path too."""
    dev = torch.device("cuda:0")
    _, eager, it, _ = _shapes_trainer(dev, False, False)
    losses = []
    for _ in range(5):
        losses.append(eager.step(next(it))["class_loss"].item())
    torch.cuda.synchronize()
    want = eager.arena.weights.clone()
    assert len(set(losses)) == 5                                   # five different samples
    for use_graph in (True, False):
        _, tr, feed, _ = _shapes_trainer(dev, use_graph, True)
        try:
            got = [tr.step()["class_loss"].item() for _ in range(5)]
        finally:
            feed.close()
        torch.cuda.synchronize()
        assert got == losses, (use_graph, got, losses)
        assert torch.equal(tr.arena.weights, want), use_graph
        assert feed.samples_staged == 5


def test_feed_recaptures_on_a_new_input_shape():
    """A sample of another size gets new static buffers and its own captured segments (one hipGraph set per shape key);
    going back to the first shape replays the first capture."""
    import dataset, layers, levels as levels_mod, retinanet, train
    from data_loaders.shapes import Shapes
    dev = torch.device("cuda:0")
    lv = levels_mod.build_levels()

    class TwoSizes(object):
        class_names = ['square', 'triangle', 'circle']
        num_classes = 3

        def __iter__(self):
            a, b = iter(Shapes(None, image_size=(160, 128), seed=1)), iter(Shapes(None, image_size=(128, 160), seed=2))
            for i in range(6):
                yield next(a if i % 3 != 1 else b)

    torch.manual_seed(0)
    net = retinanet.RetinaNet('mobilenet_v2', lv, 3, layers.elu, 0.0).to(dev)
    feed = dataset.DeviceFeed(TwoSizes(), lv, scale=128, device=dev)
    tr = train.Trainer(net, lv, device=dev, use_graph=True, input_fn=feed)
    try:
        out = [tr.step()["class_loss"].item() for _ in range(6)]
        with pytest.raises(StopIteration):
            tr.step()
    finally:
        feed.close()
    assert len(tr._graph_cache) == 2 and all(np.isfinite(out))


def test_shapes_training_reaches_a_pinned_map():
    """Training-QUALITY evidence (reference train.py:137-161 metrics, README.md:4-8): the product's own
    loop -- DeviceFeed + hipGraph step, BCE + dice + Huber (the reference's live loss), momentum 0.9, lr 1e-2, dropout 0.2 --
    on the synthetic shapes stream at 256x256 for a fixed, seeded number of steps, then train.evaluate on 32 seeded images
    of a held-out stream.  The run is deterministic (fixed-order reductions, counter-based dropout), so the metrics are
    pinned to the values measured on MI355X (tolerance for rounding-order differences between builds) and must clear a
    floor that an untrained net (mAP 0) cannot."""
    import train
    from data_loaders.shapes import Shapes
    dev = torch.device("cuda:0")
    net, tr, feed, lv = _shapes_trainer(dev, True, True, dropout=0.2, seed=0, scale=256)
    try:
        first = [tr.step() for _ in range(20)]
        c0 = float(np.mean([o["class_loss"].item() for o in first]))
        for _ in range(TRAIN_STEPS - 20):
            out = tr.step()
    finally:
        feed.close()
    tr.check_device_errors()
    assert out["class_loss"].item() < 0.5 * c0
    res = train.evaluate(net, Shapes(None, image_size=(320, 256), seed=12345), lv, 32, scale=256, device=dev)
    print("shapes 256^2 after %d steps: mAP %.4f AP50 %.4f AP75 %.4f class_iou %.4f regr_iou %.4f" %
          (TRAIN_STEPS, res["mAP"], res["AP50"], res["AP75"], res["class_iou"], res["regr_iou"]))
    assert res["images"] == 32
    assert res["AP50"] >= AP50_FLOOR and res["mAP"] >= MAP_FLOOR
    assert abs(res["mAP"] - MAP_PINNED) <= MAP_TOL and abs(res["AP50"] - AP50_PINNED) <= MAP_TOL


# measured on MI355X (gpurun, round 4) with the committed seeds; see DESIGN.md section 4 "Training quality"
# 2500 steps (9.5 s): mAP 0.6269, AP50 0.9106, AP75 0.7882 (tools/map_probe.py: 0.14 / 0.29 / 0.39 / 0.48 / 0.63 / 0.66 mAP after
# 300 / 600 / 1000 / 1500 / 2500 / 4000 steps).  The tolerance allows for other rounding orders in later builds (2500 SGD steps
# amplify them); the floors are what "it learned to detect shapes" means here.
TRAIN_STEPS = 2500
MAP_PINNED, AP50_PINNED, MAP_TOL = 0.6269, 0.9106, 0.08
MAP_FLOOR, AP50_FLOOR = 0.45, 0.75
