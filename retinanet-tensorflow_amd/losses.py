"""Detection losses (drop-in for the live path of reference losses.py:115-175 plus the focal
variant :6-15 that BASELINE's north_star names).

    class_loss, regr_loss = losses.loss(labels=labels['detection_trainable'],
                                        logits=logits['detection_trainable'])

``labels`` / ``logits`` are ``utils.Detection`` tuples of per-level dicts (P3..P7) as produced by
``utils.process_labels_and_logits``; the trainable masks ride along on the module-level
``set_trainable_masks`` / the ``trainable_masks`` argument.  One fused HIP kernel pair
(csrc/loss.hip) computes BCE + dice (or focal) and the Huber box loss over every anchor of
every level, forward and backward.
"""
import ops

LOSS_MODE = 'bce_dice'        # 'bce_dice' = live reference path; 'focal' = losses.py:6-15 + :119-122


def loss(labels, logits, trainable_masks=None, mode=None, name='loss', return_stats=False):
    """(class_loss, regr_loss) as 0-dim tensors (losses.py:155-175)."""
    mode = mode or LOSS_MODE
    keys = list(logits.regression.keys())
    if trainable_masks is None:
        trainable_masks = getattr(labels, 'trainable_masks', None)
    if trainable_masks is None:
        raise AssertionError('losses.loss needs trainable_masks (dict P3..P7 of [N,H,W,A] uint8/bool)')
    cls_logits = [logits.classification.unscaled[k] for k in keys]
    num_classes = cls_logits[0].shape[-1]
    class_loss, regr_loss, stats = ops.detection_loss(
        cls_logits, [logits.regression[k] for k in keys],
        [labels.classification.prob[k] for k in keys], [labels.regression[k] for k in keys],
        [trainable_masks[k] for k in keys], num_classes, mode)
    if return_stats:
        return class_loss, regr_loss, stats
    return class_loss, regr_loss
