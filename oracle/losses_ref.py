"""Oracle: classification / regression losses on torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference ``losses.py``:
  * focal_sigmoid_cross_entropy_with_logits  losses.py:6-15 (+ normaliser :119-122)
  * dice_loss                                losses.py:50-60
  * classification_loss (live: BCE + dice)   losses.py:115-141
  * regression_loss (Huber, fg weighted)     losses.py:144-152
  * loss                                     losses.py:155-175
Inputs are the *compacted* trainable rows ([M, C], [M, 4]) exactly as the reference's
``detection_trainable`` tensors.
"""
import torch


def sigmoid_bce_with_logits(labels, logits):
    """[TF-sem] tf.nn.sigmoid_cross_entropy_with_logits:
    max(x,0) - x*z + log(1+exp(-|x|))."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def focal_sigmoid(labels, logits, focus=2.0, alpha=0.25, eps=1e-7):
    """losses.py:6-15."""
    prob = torch.sigmoid(logits)
    is_pos = labels == 1
    prob_true = torch.where(is_pos, prob, 1 - prob)
    a = torch.where(is_pos, torch.full_like(prob, alpha), torch.full_like(prob, 1 - alpha))
    return -a * (1 - prob_true) ** focus * torch.log(prob_true + eps)


def dice(labels, logits, smooth=0.0, axis=0):
    """losses.py:50-60 with the live call's smooth=0, axis=0 (:132)."""
    prob = torch.sigmoid(logits)
    inter = (labels * prob).sum(axis)
    union = labels.sum(axis) + prob.sum(axis)
    return 1 - (2 * inter + smooth) / (union + smooth)


def fg_mask_of(label_prob):
    """utils.classmap_decode utils.py:171-179: max over classes > 0.5."""
    return label_prob.max(-1).values > 0.5


def classification_loss(labels, logits, fg_mask, mode="bce_dice"):
    """mode='bce_dice': the live path losses.py:124-139 = mean(BCE) + mean(dice(axis 0)).
    mode='focal': the commented-out path losses.py:119-122 = sum(focal)/max(#fg, 1)
    (north_star asks for focal)."""
    if mode == "bce_dice":
        return sigmoid_bce_with_logits(labels, logits).mean() + dice(labels, logits).mean()
    if mode == "focal":
        num_fg = fg_mask.float().sum()
        return focal_sigmoid(labels, logits).sum() / torch.clamp(num_fg, min=1.0)
    raise ValueError(mode)


def huber(err, delta=1.0):
    a = err.abs()
    quad = torch.clamp(a, max=delta)
    return 0.5 * quad * quad + delta * (a - quad)


def regression_loss(labels, logits, fg_mask):
    """losses.py:144-152.  [TF-sem] tf.losses.huber_loss(delta=1) with weights
    fg[:, None] broadcast over the 4 coordinates and Reduction.SUM_BY_NONZERO_WEIGHTS:
    sum(w*huber) / (number of non-zero broadcast weights), 0 when there are none.
    Pinned by losses_test.py:17-27 (= 2.0)."""
    w = fg_mask.to(labels.dtype).unsqueeze(-1).expand_as(labels)
    total = (huber(labels - logits) * w).sum()
    nz = (w != 0).to(labels.dtype).sum()
    return torch.where(nz > 0, total / torch.clamp(nz, min=1.0), torch.zeros_like(total))


def loss(label_prob, label_regr, logit_cls, logit_regr, mode="bce_dice"):
    """losses.py:155-175 -> (class_loss, regr_loss)."""
    fg = fg_mask_of(label_prob)
    return (classification_loss(label_prob, logit_cls, fg, mode),
            regression_loss(label_regr, logit_regr, fg))
