"""Decode / NMS / label-processing API (drop-in for the hot-path part of reference utils.py).

Same names and argument meaning as the reference: ``Detection`` / ``Classification`` /
``BoxesDecoded`` namedtuples (utils.py:10-13), ``regression_postprocess`` (:108-117),
``boxes_decode`` (:183-195), ``nms`` / ``nms_classwise`` / ``merge_boxes_decoded`` (:198-227),
``classmap_decode`` (:171-179), ``dict_map`` / ``dict_starmap`` / ``dict_update`` (:160-167,
:230-237), ``process_labels_and_logits`` / ``postprocess_and_mask`` (:240-284), ``iou`` (:62-97).
Tensors are NHWC fp32 torch tensors on the GPU; the arithmetic runs in csrc/decode_nms.hip
(compiled without fp contraction so NMS indices are bit-identical to the oracle).
``detect`` is the batched composition the reference only performs inside its summary builder
(train.py:68-85): all images, all levels, all classes in one pass.
"""
import ctypes as C
from collections import namedtuple
from typing import List

import numpy as np
import torch

import _rn

NMS_MAX_OUTPUT_SIZE = 1000
BoxesDecoded = namedtuple('BoxesDecoded', ['boxes', 'scores', 'class_ids'])
ClassmapDecoded = namedtuple('ClassmapDecoded', ['fg_mask'])
Detection = namedtuple('Detection', ['classification', 'regression', 'regression_postprocessed'])
Classification = namedtuple('Classification', ['unscaled', 'prob'])
# `detection_trainable` of the reference is the boolean_mask-compacted copy; here it is the same
# per-level tensors plus the masks that select the rows (the loss kernel applies them as weights)
DetectionTrainable = namedtuple('DetectionTrainable', Detection._fields + ('trainable_masks',))

ANCHOR_SIZE_MODE = 'trunc_int'     # how `anchor_sizes / image_size` is evaluated


def all_same(items):
    return all(x == items[0] for x in items)


def dict_map(f, dict):
    return {k: f(dict[k]) for k in dict}


def dict_starmap(f, dicts):
    assert all_same([list(d.keys()) for d in dicts])
    return {k: f(*[d[k] for d in dicts]) for k in dicts[0].keys()}


def dict_update(dict, keys, f):
    if len(keys) == 0:
        return f(dict)
    return {**dict, keys[0]: dict_update(dict[keys[0]], keys[1:], f)}


def merge_outputs(dict, name='merge_outputs'):
    return torch.cat(list(dict.values()), 0)


_ANCHOR_CACHE = {}


def _anchor_tensor(anchor_boxes, device):
    """[A,2] anchor table on the device.  Host tables are uploaded once and cached (an H2D copy
    cannot be captured into a hipGraph, and the tables never change)."""
    if torch.is_tensor(anchor_boxes):
        return anchor_boxes.to(device=device, dtype=torch.float32).contiguous()
    a = np.ascontiguousarray(anchor_boxes, dtype=np.float32)
    key = (a.tobytes(), a.shape, str(device))
    t = _ANCHOR_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(a).to(device).contiguous()
        _ANCHOR_CACHE[key] = t
    return t


def regression_postprocess(regression, anchor_boxes, name='regression_postprocess'):
    """[N,H,W,A,4] (dy, dx, log h, log w) anchor-relative -> normalised corners [y1,x1,y2,x2].
    `anchor_boxes`: [A,2] anchor (h, w) already divided by the image size."""
    n, h, w, a, _ = regression.shape
    regression = regression.contiguous()
    anchors = _anchor_tensor(anchor_boxes, regression.device)
    assert anchors.shape == (a, 2)
    out = torch.empty_like(regression)
    _rn.check(_rn.lib().rn_decode_boxes(_rn.f32(regression), _rn.f32(anchors), _rn.f32(out), n, h, w, a,
                                        _rn.stream()), 'rn_decode_boxes')
    return out


def classmap_decode(classmap, name='classmap_decoder'):
    return ClassmapDecoded(fg_mask=classmap.max(-1).values > 0.5)


def _fp_ptr(t):
    """(pointer, is_fp16) of an fp32 / fp16 device tensor."""
    if t.dtype not in (torch.float32, torch.float16):
        raise _rn.RnError("detect: fp32 or fp16 maps (got %s)" % t.dtype)
    return _rn.ptr(t), 1 if t.dtype == torch.float16 else 0


def _det_call(fn_name, probs, boxes, num_classes, n, score_threshold, iou_threshold, max_per_class, capacity, raw=None,
              logits=False):
    """probs/boxes: lists over levels of [n, rows, C] / [n, rows, 4] tensors; or boxes=None and raw = list over levels of
    (regression [n,H,W,A,4], anchor tensor [A,2]): candidates are decoded on the fly (see rn_det_level).  probs and raw
    regressions may be fp16 (read as stored); logits=True: `probs` are class logits, the sigmoid runs inside the scan."""
    L = _rn.lib()
    dev = probs[0].device
    levels = (_rn.DetLevel * len(probs))()
    total_rows = 0
    for i, p in enumerate(probs):
        assert p.shape[0] == n and p.shape[2] == num_classes
        levels[i].prob, levels[i].prob_f16 = _fp_ptr(p)
        levels[i].prob_is_logit = 1 if logits else 0
        levels[i].rows_per_image = p.shape[1]
        if raw is None:
            assert boxes[i].shape[0] == n and p.shape[1] == boxes[i].shape[1]
            levels[i].boxes = _rn.f32(boxes[i])
        else:
            reg, anc = raw[i]
            assert reg.dim() == 5 and reg.shape[0] == n and reg.shape[4] == 4 and reg.shape[3] == anc.shape[0]
            assert reg.shape[1] * reg.shape[2] * reg.shape[3] == p.shape[1]
            levels[i].boxes = None
            levels[i].regression, levels[i].regression_f16 = _fp_ptr(reg)
            levels[i].anchor_sizes = _rn.f32(anc)
            levels[i].grid_h, levels[i].grid_w, levels[i].num_anchors = reg.shape[1], reg.shape[2], reg.shape[3]
        total_rows += p.shape[1]
    cap = int(capacity) if capacity is not None else n * total_rows
    params = _rn.DetParams(n, num_classes, max_per_class, score_threshold, iou_threshold, cap)
    need = L.rn_detect_workspace(levels, len(probs), C.byref(params))
    ws = _rn.workspace(need, dev)
    out_boxes = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((cap,), dtype=torch.float32, device=dev)
    out_class = torch.empty((cap,), dtype=torch.int32, device=dev)
    out_image = torch.empty((cap,), dtype=torch.int32, device=dev)
    out_anchor = torch.empty((cap,), dtype=torch.int64, device=dev)
    counts = torch.empty((2 + n,), dtype=torch.int64, device=dev)     # (every entry is written by the pipeline itself)
    _rn.check(getattr(L, fn_name)(levels, len(probs), C.byref(params), _rn.f32(out_boxes), _rn.f32(out_scores),
                                  _rn.ptr(out_class), _rn.ptr(out_image), _rn.ptr(out_anchor), _rn.ptr(counts),
                                  ws.data_ptr(), ws.numel(), _rn.stream()), fn_name)
    return out_boxes, out_scores, out_class, out_image, out_anchor, counts, cap


def boxes_decode(classifications, regressions, name='boxes_decode'):
    """ONE image / level: `classifications` [..., C] probabilities, `regressions` [..., 4] decoded
    boxes -> BoxesDecoded of the rows whose max class prob > 0.5, in row-major order."""
    c = classifications.shape[-1]
    p = classifications.reshape(1, -1, c).contiguous()
    b = regressions.reshape(1, -1, 4).contiguous()
    ob, os_, oc, _, _, counts, cap = _det_call('rn_boxes_decode', [p], [b], c, 1, 0.5, 0.5, NMS_MAX_OUTPUT_SIZE, None)
    k = int(counts[0].item())
    assert k <= cap
    return BoxesDecoded(boxes=ob[:k], scores=os_[:k], class_ids=oc[:k].long())


def merge_boxes_decoded(decoded: List[BoxesDecoded]):
    return BoxesDecoded(boxes=torch.cat([d.boxes for d in decoded], 0),
                        scores=torch.cat([d.scores for d in decoded], 0),
                        class_ids=torch.cat([d.class_ids for d in decoded], 0))


def _nms_arrays(boxes, scores, class_ids, num_classes, max_output_size):
    L = _rn.lib()
    dev = boxes.device
    k = boxes.shape[0]
    cap = max(k, 1)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    cls32 = class_ids.to(torch.int32).contiguous()
    img32 = torch.zeros((cap,), dtype=torch.int32, device=dev)
    count = torch.tensor([k], dtype=torch.int64, device=dev)
    params = _rn.DetParams(1, num_classes, max_output_size, 0.5, 0.5, cap)
    need = L.rn_nms_classwise_workspace(C.byref(params))
    ws = _rn.workspace(need, dev)
    ob = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((cap,), dtype=torch.float32, device=dev)
    oc = torch.empty((cap,), dtype=torch.int32, device=dev)
    oi = torch.empty((cap,), dtype=torch.int32, device=dev)
    oidx = torch.empty((cap,), dtype=torch.int64, device=dev)
    counts = torch.empty((3,), dtype=torch.int64, device=dev)
    if k == 0:
        return ob[:0], os_[:0], oc[:0].long(), oidx[:0]
    _rn.check(L.rn_nms_classwise(_rn.f32(boxes), _rn.f32(scores), _rn.ptr(cls32), _rn.ptr(img32), _rn.ptr(count),
                                 C.byref(params), _rn.f32(ob), _rn.f32(os_), _rn.ptr(oc), _rn.ptr(oi), _rn.ptr(oidx),
                                 _rn.ptr(counts), ws.data_ptr(), ws.numel(), _rn.stream()), 'rn_nms_classwise')
    m = int(counts[1].item())
    return ob[:m], os_[:m], oc[:m].long(), oidx[:m]


def nms(decoded: BoxesDecoded, max_output_size=NMS_MAX_OUTPUT_SIZE, name='nms'):
    """tf.image.non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5) + gather."""
    zeros = torch.zeros_like(decoded.class_ids)
    _, _, _, idx = _nms_arrays(decoded.boxes, decoded.scores, zeros, 1, max_output_size)
    return BoxesDecoded(boxes=decoded.boxes[idx], scores=decoded.scores[idx], class_ids=decoded.class_ids[idx])


def nms_classwise(decoded: BoxesDecoded, num_classes, name='nms_classwise'):
    """Per-class NMS, output class-major then score-descending (utils.py:198-210)."""
    ob, os_, oc, _ = _nms_arrays(decoded.boxes, decoded.scores, decoded.class_ids, num_classes, NMS_MAX_OUTPUT_SIZE)
    return BoxesDecoded(boxes=ob, scores=os_, class_ids=oc)


def detect(class_probs, regressions_postprocessed, num_classes, score_threshold=0.5, iou_threshold=0.5,
           max_per_class=NMS_MAX_OUTPUT_SIZE, capacity=None, return_raw=False):
    """Batched train.py:68-85: dicts P3..P7 of [N,H,W,A,C] probabilities and [N,H,W,A,4] decoded
    boxes -> list (one per image) of BoxesDecoded after class-wise NMS.  One pass on the device."""
    keys = list(class_probs.keys())
    n = class_probs[keys[0]].shape[0]
    probs = [class_probs[k].reshape(n, -1, num_classes).contiguous() for k in keys]
    boxes = [regressions_postprocessed[k].reshape(n, -1, 4).contiguous() for k in keys]
    ob, os_, oc, oi, oa, counts, cap = _det_call('rn_detect', probs, boxes, num_classes, n, score_threshold,
                                                 iou_threshold, max_per_class, capacity)
    if return_raw:
        return ob, os_, oc, oi, oa, counts
    cnt = counts.cpu().tolist()
    if cnt[0] > cap:
        raise _rn.RnError('detect: %d candidates exceed capacity %d' % (cnt[0], cap))
    out, off = [], 0
    for i in range(n):
        m = cnt[2 + i]
        out.append(BoxesDecoded(boxes=ob[off:off + m], scores=os_[off:off + m], class_ids=oc[off:off + m].long()))
        off += m
    return out


def detect_raw(class_probs, regressions, anchor_sizes, num_classes, score_threshold=0.5, iou_threshold=0.5,
               max_per_class=NMS_MAX_OUTPUT_SIZE, capacity=None, return_raw=False, logits=False):
    """`detect` without the full-map decode: dicts P3..P7 of [N,H,W,A,C] probabilities, RAW regressions [N,H,W,A,4] and
    normalised anchor sizes (`Level.normalized_anchor_sizes`).  Only the rows that pass the score threshold are decoded
    (same arithmetic as `regression_postprocess`, same results as `detect(probs, regression_postprocess(...))`).
    The maps may be fp16 (BASELINE configs[4]: read as stored, 2 bytes per element).  logits=True: `class_probs` holds the
    net's class LOGITS and the sigmoid (train.py:74) runs inside the scan -- no probability map is written."""
    keys = list(class_probs.keys())
    n = class_probs[keys[0]].shape[0]
    probs = [class_probs[k].reshape(n, -1, num_classes).contiguous() for k in keys]
    dev = probs[0].device
    raw = [(regressions[k].contiguous(), _anchor_tensor(anchor_sizes[k], dev)) for k in keys]
    ob, os_, oc, oi, oa, counts, cap = _det_call('rn_detect', probs, None, num_classes, n, score_threshold, iou_threshold,
                                                 max_per_class, capacity, raw=raw, logits=logits)
    if return_raw:
        return ob, os_, oc, oi, oa, counts
    cnt = counts.cpu().tolist()
    if cnt[0] > cap:
        raise _rn.RnError('detect: %d candidates exceed capacity %d' % (cnt[0], cap))
    out, off = [], 0
    for i in range(n):
        m = cnt[2 + i]
        out.append(BoxesDecoded(boxes=ob[off:off + m], scores=os_[off:off + m], class_ids=oc[off:off + m].long()))
        off += m
    return out


def iou(a, b, name='iou'):
    """utils.py:62-97 with broadcasting, on the device (rn_iou; the arithmetic the assignment kernel uses).
    The reference's call shape -- a [O,1,1,1,4] against b [1,H,W,A,4] (dataset.py:57-60): the broadcast dims of a and
    b do not overlap -- is a pairwise launch; equal shapes are element-wise; any other broadcast is expanded first."""
    assert a.shape[-1] == 4 and b.shape[-1] == 4
    sa, sb = tuple(a.shape[:-1]), tuple(b.shape[:-1])
    nd = max(len(sa), len(sb))
    sa, sb = (1,) * (nd - len(sa)) + sa, (1,) * (nd - len(sb)) + sb
    out_shape = tuple(max(x, y) for x, y in zip(sa, sb))
    assert all(x in (1, o) and y in (1, o) for x, y, o in zip(sa, sb, out_shape)), "iou: shapes do not broadcast"
    af, bf = a.reshape(-1, 4).contiguous().float(), b.reshape(-1, 4).contiguous().float()
    # pairwise iff every non-1 dim of a comes before every non-1 dim of b (then out[i, j] flattens to out_shape)
    last_a = max([i for i, x in enumerate(sa) if x != 1], default=-1)
    first_b = min([i for i, y in enumerate(sb) if y != 1], default=nd)
    if sa == sb:
        pairwise = 0
    elif last_a < first_b:
        pairwise = 1
    else:
        af = a.reshape(sa + (4,)).expand(out_shape + (4,)).reshape(-1, 4).contiguous().float()
        bf = b.reshape(sb + (4,)).expand(out_shape + (4,)).reshape(-1, 4).contiguous().float()
        pairwise = 0
    out = torch.empty((af.shape[0] * bf.shape[0] if pairwise else af.shape[0],), dtype=torch.float32, device=a.device)
    if out.numel() == 0:
        return out.reshape(out_shape)
    bad = torch.zeros((1,), dtype=torch.int32, device=a.device)
    _rn.check(_rn.lib().rn_iou(_rn.f32(af), af.shape[0], _rn.f32(bf), bf.shape[0], pairwise, _rn.f32(out), _rn.ptr(bad),
                               _rn.stream()), 'rn_iou')
    assert int(bad.item()) == 0, 'iou: a box with y2 < y1 or x2 < x1 (utils.py:65-68)'
    return out.reshape(out_shape)


class _LazyLevels(dict):
    """Per-level dict whose values are computed on first access.  In the reference these tensors are graph nodes that a
    training step never fetches (only summaries / metrics read `regression_postprocessed`), so TensorFlow prunes them; a
    define-by-run host has to be lazy explicitly or it pays ten decode launches per step for nothing."""

    def __init__(self, keys, fn):
        super().__init__()
        self._keys, self._fn = list(keys), fn

    def __missing__(self, k):
        if k not in self._keys:
            raise KeyError(k)
        v = self._fn(k)
        dict.__setitem__(self, k, v)
        return v

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def keys(self):
        return list(self._keys)

    def values(self):
        return [self[k] for k in self._keys]

    def items(self):
        return [(k, self[k]) for k in self._keys]

    def __contains__(self, k):
        return k in self._keys


def postprocess_and_mask(input, trainable_masks, image_size, levels, name='postprocess_and_mask'):
    """utils.py:258-284 without the gather: the per-level tensors and the masks are kept side
    by side (``detection_trainable`` carries the masks); losses.loss weighs rows by the mask,
    which gives the reference's numbers without materialising the compacted [M, C] copies."""
    detection = Detection(
        classification=input['detection']['classifications'],
        regression=input['detection']['regressions'],
        regression_postprocessed=_LazyLevels(
            levels, lambda l: regression_postprocess(input['detection']['regressions'][l],
                                                     levels[l].normalized_anchor_sizes(image_size, ANCHOR_SIZE_MODE))))
    detection_trainable = DetectionTrainable(
        classification=detection.classification, regression=detection.regression,
        regression_postprocessed=detection.regression_postprocessed, trainable_masks=trainable_masks)
    return {**input, 'detection': detection, 'detection_trainable': detection_trainable,
            'trainable_masks': trainable_masks}


def _sigmoid(t):
    import ops                      # (ops imports nothing from here; kept local so that utils stays importable without it)
    return ops.activation(t.detach(), 'sigmoid')


def process_labels_and_logits(labels, logits, levels, name='process_labels_and_logits'):
    """utils.py:240-255.  `labels`: features dict (image, detection{classifications, regressions},
    trainable_masks); `logits`: {'detection': net output}."""
    labels = dict_update(labels, ['detection', 'classifications'], lambda c: Classification(unscaled=None, prob=c))
    # utils.py:245-247 sets prob = sigmoid(unscaled); the loss kernel takes the logits (the sigmoid is fused there), so the
    # probabilities are computed only if somebody reads them (summaries / metrics in the reference): the step pays nothing
    logits = dict_update(logits, ['detection', 'classifications'],
                         lambda c: Classification(unscaled=c, prob=_LazyLevels(c.keys(), lambda k: _sigmoid(c[k]))))
    image_size = tuple(labels['image'].shape[1:3])
    labels = postprocess_and_mask(labels, labels['trainable_masks'], image_size=image_size, levels=levels)
    logits = postprocess_and_mask(logits, labels['trainable_masks'], image_size=image_size, levels=levels)
    return labels, logits
