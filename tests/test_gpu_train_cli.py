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
    assert train.main(argv) == 120                                                    # resumed from step 60
    assert "restored step 60" in capsys.readouterr().out
