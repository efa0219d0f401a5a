"""Oracle: one training step (forward, loss, L2, backward, optimizer) on torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference ``train.py``:
  * model_fn composition            train.py:206-243 -> total_loss()
  * build_train_step                train.py:111-134 -> apply_optimizer(), clip_by_global_norm()
  * MirroredStrategy semantics      train.py:261-267 -> mean of per-replica gradients
and ``utils.process_labels_and_logits`` / ``postprocess_and_mask`` (utils.py:240-284) for
the trainable-row compaction (concat P3..P7 of boolean_mask).
"""
import torch

from . import losses_ref, model_ref

LEVELS = ("P3", "P4", "P5", "P6", "P7")


def compact(per_level, masks):
    """concat_{k=P3..P7} boolean_mask(x_k, mask_k) (utils.py:270-278, dict order Q16)."""
    return torch.cat([per_level[k][masks[k]] for k in LEVELS], 0)


def total_loss(params, image, labels, num_classes, loss_mode="bce_dice", act="elu",
               backbone="mobilenet_v2", dropout=None):
    """labels: dict(classifications, regressions, trainable_masks) of dicts P3..P7 with a
    leading batch axis.  Returns (total, class_loss, regr_loss, reg_loss).  `dropout`: optional hook with the masks of
    this step (oracle/dropout_ref.Sites); None = dropout_rate 0."""
    out = model_ref.retinanet_forward(params, image, num_classes, act=act, backbone=backbone, dropout=dropout)
    masks = {k: labels["trainable_masks"][k].bool() for k in LEVELS}
    cls_loss, regr_loss = losses_ref.loss(
        compact(labels["classifications"], masks), compact(labels["regressions"], masks),
        compact(out["classifications"], masks), compact(out["regressions"], masks), loss_mode)
    reg = model_ref.l2_regularization(params)
    return cls_loss + regr_loss + reg, cls_loss, regr_loss, reg


def clip_by_global_norm(grads, clip_norm):
    """[TF-sem] tf.clip_by_global_norm: g * clip / max(global_norm, clip)."""
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    scale = clip_norm / torch.clamp(gn, min=clip_norm)
    return [g * scale for g in grads], gn


def apply_optimizer(kind, params, grads, state, lr, step):
    """[TF-sem] train.py:114-119.
    momentum: tf.train.MomentumOptimizer(lr, 0.9): acc = 0.9*acc + g; w -= lr*acc.
    rmsprop:  tf.train.RMSPropOptimizer(lr, decay 0.9, momentum 0.9, eps 1e-10):
              ms = 0.9*ms + 0.1*g^2 (ms starts at 1); mom = 0.9*mom + lr*g/sqrt(ms+eps); w -= mom.
    adam:     tf.train.AdamOptimizer(lr): b1 .9, b2 .999, eps 1e-8,
              lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps).
    `step` is 1-based.  Updates in place."""
    for name, g in grads.items():
        w = params[name]
        st = state.setdefault(name, {})
        if kind == "momentum":
            acc = st.setdefault("acc", torch.zeros_like(w))
            acc.mul_(0.9).add_(g)
            w.sub_(lr * acc)
        elif kind == "rmsprop":
            ms = st.setdefault("ms", torch.ones_like(w))
            mom = st.setdefault("mom", torch.zeros_like(w))
            ms.mul_(0.9).add_(0.1 * g * g)
            mom.mul_(0.9).add_(lr * g / torch.sqrt(ms + 1e-10))
            w.sub_(mom)
        elif kind == "adam":
            m = st.setdefault("m", torch.zeros_like(w))
            v = st.setdefault("v", torch.zeros_like(w))
            m.mul_(0.9).add_(0.1 * g)
            v.mul_(0.999).add_(0.001 * g * g)
            lr_t = lr * (1 - 0.999 ** step) ** 0.5 / (1 - 0.9 ** step)
            w.sub_(lr_t * m / (torch.sqrt(v) + 1e-8))
        else:
            raise ValueError(kind)


def train_step(params, image, labels, num_classes, state, lr=1e-2, optimizer="momentum",
               step=1, loss_mode="bce_dice", grad_clip_norm=None, replicas=None, dropout=None):
    """One optimizer step.  `replicas`: optional list of (image, labels) per replica; the
    gradient is then the mean over replicas (MirroredStrategy).  Returns the
    loss tuple of the first replica and the dict of applied gradients."""
    batches = replicas if replicas is not None else [(image, labels)]
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    sums, first = None, None
    for img, lab in batches:
        losses = total_loss(leaves, img, lab, num_classes, loss_mode, dropout=dropout)
        gs = torch.autograd.grad(losses[0], list(leaves.values()))
        sums = list(gs) if sums is None else [a + b for a, b in zip(sums, gs)]
        if first is None:
            first = tuple(float(x) for x in losses)
    grads = [g / len(batches) for g in sums]
    if grad_clip_norm is not None:
        grads, _ = clip_by_global_norm(grads, grad_clip_norm)
    named = dict(zip(leaves.keys(), grads))
    with torch.no_grad():
        apply_optimizer(optimizer, params, named, state, lr, step)
    return first, named
