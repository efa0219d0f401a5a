"""Checkpoint I/O: weights + optimizer slots in one safetensors file, keyed by the
model's parameter names (the reference delegates this to tf.estimator, train.py:263-273: model_dir +
save_checkpoints_steps; no custom format exists to be compatible with)."""
import json
import os

import torch
from safetensors import safe_open
from safetensors.torch import save_file


def save(path, net, trainer=None, step=0, extra=None):
    """Weights under "model/<name>"; optimizer slots PER PARAMETER under "optimizer/state{1,2}/<name>" (independent of the
    arena's layout); the dropout step counter, so that a resumed run does not replay the masks of step 0."""
    tensors = {"model/" + k: v.detach().cpu().contiguous().clone() for k, v in net.named_parameters()}
    meta = {"step": str(int(step)), "format": "retinanet-amd-v2"}
    if trainer is not None:
        names = {id(p): k for k, p in net.named_parameters()}
        for p, (off, size) in zip(trainer.arena.params, trainer.arena.offsets):
            name = names[id(p)]
            tensors["optimizer/state1/" + name] = trainer.opt.state1[off:off + size].detach().cpu().clone().view(p.shape)
            if trainer.opt.state2 is not None:
                tensors["optimizer/state2/" + name] = trainer.opt.state2[off:off + size].detach().cpu().clone().view(p.shape)
        # (stored without this replica's offset: every rank adds its own back on load, replicas keep distinct dropout streams)
        tensors["trainer/drop_counter"] = trainer.drop_counter.detach().cpu().clone() - int(getattr(trainer, "drop_rank_offset", 0))
        meta.update(optimizer=trainer.opt.kind, step_count=str(trainer.opt.step_count))
    if extra:
        meta["extra"] = json.dumps(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    save_file(tensors, tmp, metadata=meta)
    os.replace(tmp, path)


def load_extra(path):
    """The `extra` dict a checkpoint was saved with (train.py main(): epochs done, samples drawn per rank); {} if none."""
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    return json.loads(meta["extra"]) if "extra" in meta else {}


def load(path, net, trainer=None):
    """Restores in place (the parameters keep pointing into the trainer's arena).  Returns the saved step.
    Missing keys and shape mismatches raise a ValueError that names the parameter."""
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
        keys = set(f.keys())
        fmt = meta.get("format")
        if fmt != "retinanet-amd-v2":       # checked BEFORE any weight is overwritten in place
            raise ValueError("checkpoint %s has format %r; this build reads 'retinanet-amd-v2' (per-parameter optimizer slots)" % (path, fmt))
        if trainer is not None and meta.get("optimizer") is not None:
            if meta.get("optimizer") != trainer.opt.kind:
                raise ValueError("checkpoint optimizer %s != %s" % (meta.get("optimizer"), trainer.opt.kind))
            need = ["optimizer/state1/"] + (["optimizer/state2/"] if trainer.opt.state2 is not None else [])
            for k, _ in net.named_parameters():
                for pre in need:
                    if pre + k not in keys:
                        raise ValueError("checkpoint %s has no tensor %r" % (path, pre + k))
        for k, p in net.named_parameters():
            if "model/" + k not in keys:
                raise ValueError("checkpoint %s has no tensor %r" % (path, "model/" + k))
            shape = tuple(f.get_slice("model/" + k).get_shape())
            if shape != tuple(p.shape):
                raise ValueError("checkpoint %s: %r has shape %s, the model expects %s" % (path, "model/" + k, shape, tuple(p.shape)))

        def get(key, like):
            if key not in keys:
                raise ValueError("checkpoint %s has no tensor %r" % (path, key))
            t = f.get_tensor(key)
            if tuple(t.shape) != tuple(like.shape):
                raise ValueError("checkpoint %s: %r has shape %s, the model expects %s" % (path, key, tuple(t.shape), tuple(like.shape)))
            return t

        with torch.no_grad():
            for k, p in net.named_parameters():
                p.copy_(get("model/" + k, p).to(p.device))
            if trainer is not None and meta.get("optimizer") is not None:
                if meta.get("optimizer") != trainer.opt.kind:
                    raise ValueError("checkpoint optimizer %s != %s" % (meta.get("optimizer"), trainer.opt.kind))
                names = {id(p): k for k, p in net.named_parameters()}
                for p, (off, size) in zip(trainer.arena.params, trainer.arena.offsets):
                    name = names[id(p)]
                    trainer.opt.state1[off:off + size].copy_(get("optimizer/state1/" + name, p).reshape(-1).to(trainer.opt.state1.device))
                    if trainer.opt.state2 is not None:
                        trainer.opt.state2[off:off + size].copy_(get("optimizer/state2/" + name, p).reshape(-1).to(trainer.opt.state2.device))
                if "trainer/drop_counter" in keys:
                    trainer.drop_counter.copy_((f.get_tensor("trainer/drop_counter") + int(getattr(trainer, "drop_rank_offset", 0)))
                                               .to(trainer.drop_counter.device))
                trainer.opt.step_count = int(meta.get("step_count", 0))
    import ops_f16
    ops_f16.weights_changed()
    return int(meta.get("step", 0))
