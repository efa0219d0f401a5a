"""Checkpoint I/O: weights + optimizer slots in one safetensors file, keyed by the
model's parameter names (the reference delegates this to tf.estimator, train.py:263-273: model_dir +
save_checkpoints_steps; no custom format exists to be compatible with)."""
import json
import os

import torch
from safetensors.torch import load_file, save_file


def save(path, net, trainer=None, step=0, extra=None):
    tensors = {"model/" + k: v.detach().cpu().contiguous().clone() for k, v in net.named_parameters()}
    meta = {"step": str(int(step)), "format": "retinanet-amd-v1"}
    if trainer is not None:
        tensors["optimizer/state1"] = trainer.opt.state1.detach().cpu().clone()
        if trainer.opt.state2 is not None:
            tensors["optimizer/state2"] = trainer.opt.state2.detach().cpu().clone()
        meta.update(optimizer=trainer.opt.kind, step_count=str(trainer.opt.step_count))
    if extra:
        meta["extra"] = json.dumps(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    save_file(tensors, tmp, metadata=meta)
    os.replace(tmp, path)


def load(path, net, trainer=None):
    """Restores in place (the parameters keep pointing into the trainer's arena).  Returns the saved step."""
    from safetensors import safe_open
    tensors = load_file(path)
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.copy_(tensors["model/" + k].to(p.device))
        if trainer is not None and "optimizer/state1" in tensors:
            assert meta.get("optimizer") == trainer.opt.kind, "checkpoint optimizer %s != %s" % (meta.get("optimizer"), trainer.opt.kind)
            trainer.opt.state1.copy_(tensors["optimizer/state1"].to(trainer.opt.state1.device))
            if trainer.opt.state2 is not None:
                trainer.opt.state2.copy_(tensors["optimizer/state2"].to(trainer.opt.state2.device))
            trainer.opt.step_count = int(meta.get("step_count", 0))
    return int(meta.get("step", 0))
