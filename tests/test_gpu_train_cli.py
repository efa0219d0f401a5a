"""The minimal CLI driver (reference train.py flags) trains on the synthetic shapes loader, saves and resumes."""
import os

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
