"""Anchor assignment on the device (drop-in for reference dataset.py:43-142 ``level_labels`` /
``build_labels``), batched over images.

    cls, reg, masks = build_labels(image_size, class_ids, boxes, num_obj, levels, num_classes)
    # dicts P3..P7: [N,H,W,A,C] f32 one-hot (zero where IoU<0.5), [N,H,W,A,4] f32, [N,H,W,A] u8

The tf.data input pipeline of the reference (file reading, JPEG decode, shuffle,
dataset.py:145-233) is out of scope; what remains of it here is the
label construction and the ``[image, hflip(image)]`` batch convention (dataset.py:182-204).
"""
import ctypes as C
import math

import numpy as np
import torch

import _rn
import utils

NEG_IOU_THRESHOLD = 0.4
POS_IOU_THRESHOLD = 0.5
MEAN = [0.46618041, 0.44669811, 0.40252436]
STD = [0.27940595, 0.27489075, 0.28920765]
ANCHOR_SIZE_MODE = 'trunc_int'


def level_labels(image_size, class_id, true_box, level, factor, num_classes, num_obj=None, return_argmax=False):
    """class_id [N, O] int32, true_box [N, O, 4] normalised corners, num_obj [N] (valid objects per
    image, default O) -> (classification [N,H,W,A,C], regression [N,H,W,A,4], trainable [N,H,W,A])."""
    dev = true_box.device
    n, o = true_box.shape[0], true_box.shape[1]
    true_box = true_box.contiguous().float()
    class_id = class_id.to(torch.int32).contiguous()
    if num_obj is None:
        num_obj = torch.full((n,), o, dtype=torch.int32, device=dev)
    num_obj = num_obj.to(torch.int32).contiguous()
    anchors = utils._anchor_tensor(level.normalized_anchor_sizes(image_size, ANCHOR_SIZE_MODE), dev)
    a = anchors.shape[0]
    gh, gw = int(math.ceil(image_size[0] / factor)), int(math.ceil(image_size[1] / factor))
    cls = torch.empty((n, gh, gw, a, num_classes), dtype=torch.float32, device=dev)
    reg = torch.empty((n, gh, gw, a, 4), dtype=torch.float32, device=dev)
    msk = torch.empty((n, gh, gw, a), dtype=torch.uint8, device=dev)
    arg = torch.empty((n, gh, gw, a), dtype=torch.int32, device=dev) if return_argmax else None
    _rn.check(_rn.lib().rn_anchor_assign(_rn.f32(true_box), _rn.ptr(class_id), _rn.ptr(num_obj), n, o,
                                         _rn.f32(anchors), a, gh, gw, num_classes, _rn.f32(cls), _rn.f32(reg),
                                         _rn.ptr(msk), _rn.ptr(arg), _rn.stream()), 'rn_anchor_assign')
    if return_argmax:
        return cls, reg, msk, arg
    return cls, reg, msk


def build_labels(image_size, class_ids, boxes, levels, num_classes, num_obj=None, flip_pair=False):
    """dataset.py:126-142 for a batch: every level's maps from one launch (rn_anchor_assign_levels); the per-level
    results are the ones `level_labels` gives.
    flip_pair=True: the reference's batch of two per sample (dataset.py:182-204) from the same launch -- image i is assigned
    once and fills batch slots 2i (as is) and 2i+1 (= augmentation.flip of its maps: W reversed, x shift negated), so the
    mirror image's labels are bit for bit the flipped maps, and the assignment runs once per sample."""
    dev = boxes.device
    n_src, o = boxes.shape[0], boxes.shape[1]
    n = 2 * n_src if flip_pair else n_src
    boxes = boxes.contiguous().float()
    class_ids = class_ids.to(torch.int32).contiguous()
    if num_obj is None:
        num_obj = torch.full((n_src,), o, dtype=torch.int32, device=dev)
    num_obj = num_obj.to(torch.int32).contiguous()
    names = list(levels)
    lv = (_rn.AssignLevel * len(names))()
    classifications, regressions, trainable_masks, keep = {}, {}, {}, []
    a = None
    for i, pn in enumerate(names):
        factor = 2 ** int(pn[-1])
        anchors = utils._anchor_tensor(levels[pn].normalized_anchor_sizes(image_size, ANCHOR_SIZE_MODE), dev)
        assert a in (None, anchors.shape[0]), "every level carries the same number of anchors (levels.py:32-44)"
        a = anchors.shape[0]
        gh, gw = int(math.ceil(image_size[0] / factor)), int(math.ceil(image_size[1] / factor))
        classifications[pn] = torch.empty((n, gh, gw, a, num_classes), dtype=torch.float32, device=dev)
        regressions[pn] = torch.empty((n, gh, gw, a, 4), dtype=torch.float32, device=dev)
        trainable_masks[pn] = torch.empty((n, gh, gw, a), dtype=torch.uint8, device=dev)
        keep.append(anchors)
        lv[i] = _rn.AssignLevel(anchors.data_ptr(), gh, gw, classifications[pn].data_ptr(), regressions[pn].data_ptr(),
                                trainable_masks[pn].data_ptr(), None)
    fn = _rn.lib().rn_anchor_assign_levels_pair if flip_pair else _rn.lib().rn_anchor_assign_levels
    _rn.check(fn(_rn.f32(boxes), _rn.ptr(class_ids), _rn.ptr(num_obj), n_src, o, lv, len(names), a, num_classes, _rn.stream()),
              'rn_anchor_assign_levels')
    return classifications, regressions, trainable_masks


def flip_boxes(boxes):
    """h-flip of normalised corner boxes [.., 4] = [y1, x1, y2, x2] (labels of the flipped image
    of the reference's [image, hflip] batch, dataset.py:182-204, are built from these)."""
    return torch.stack([boxes[..., 0], 1.0 - boxes[..., 3], boxes[..., 2], 1.0 - boxes[..., 1]], -1)


def rescale_size(size, scale):
    """New (h, w) of dataset.py:145-151: shorter side -> `scale`, tf.round (half to even) of size * ratio, the
    arithmetic in float32 as tf.to_float / tf.round do it."""
    size = np.asarray(size, np.float32)
    ratio = np.float32(scale) / size[int(np.argmin(size))]
    new = np.rint(size * ratio).astype(np.int32)                 # np.rint == tf.round: half to even
    return int(new[0]), int(new[1])


def rescale_image(image, scale=None, size=None, normalize=False, out=None):
    """tf.image.resize_images(image, new_size, BILINEAR, align_corners=True) (dataset.py:145-151) on the device.
    image: [H,W,C] or [N,H,W,C], uint8 (converted like tf.image.convert_image_dtype: * 1/255) or fp32.
    normalize=True also applies train.py:48-49 preprocess_image ((v - MEAN) / STD) in the same pass."""
    batched = image.dim() == 4
    x = (image if batched else image[None]).contiguous()
    n, h, w, c = x.shape
    oh, ow = size if size is not None else rescale_size((h, w), scale)
    if out is None:
        y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    else:                                          # e.g. slot 0 of the [sample, hflip] batch buffer
        y = out if batched else out[None]
        assert y.is_contiguous() and tuple(y.shape) == (n, oh, ow, c) and y.dtype == torch.float32
    assert x.dtype in (torch.uint8, torch.float32)
    mean = std = None
    if normalize:
        assert c == 3
        mean = (C.c_float * 3)(*MEAN)
        std = (C.c_float * 3)(*STD)
    _rn.check(_rn.lib().rn_resize_bilinear_normalize(_rn.ptr(x), 1 if x.dtype == torch.uint8 else 0, _rn.f32(y), n, h, w, c,
                                                     oh, ow, mean, std, _rn.stream()), 'rn_resize_bilinear_normalize')
    return y if batched else y[0]


def preprocess_image(image):
    """(image - MEAN) / STD (train.py:48-49) for a float image already at its final size."""
    return rescale_image(image, size=tuple(image.shape[-3:-1]), normalize=True)


def sample_batch(image, boxes, class_ids, levels, num_classes, scale=None, num_obj=None, normalize=True):
    """One sample -> the reference's batch of two (dataset.py:154-215 after the decode): rescale_image (dataset.py:145-151) +
    preprocess_image (train.py:48-49) in ONE kernel into slot 0, its h-flip (augmentation.py:5-22) into slot 1, and the
    labels of both from ONE assignment launch (build_labels(flip_pair=True)).  Four launches, every shape static: the
    function is capturable into the train step's hipGraph (DeviceFeed.features), and it IS what build_dataset runs eagerly.
      image [H,W,3] uint8 / fp32 (device), boxes [1,O,4] normalised corners, class_ids [1,O] int32, num_obj [1] or None."""
    import augmentation
    h, w = int(image.shape[0]), int(image.shape[1])
    size = rescale_size((h, w), scale) if scale is not None else (h, w)
    pair = torch.empty((2, size[0], size[1], int(image.shape[2])), dtype=torch.float32, device=image.device)
    rescale_image(image, size=size, normalize=normalize, out=pair[0])
    augmentation._flip(pair[0], 1, out=pair[1])          # (per-element normalisation commutes with the flip: same bits)
    c, r, m = build_labels(size, class_ids, boxes, levels, num_classes, num_obj=num_obj, flip_pair=True)
    return {'image': pair, 'image_size': size, 'detection': {'classifications': c, 'regressions': r}, 'trainable_masks': m}


def build_dataset(data_loader, levels, scale=None, shuffle=None, augment=False, device='cuda', normalize=True):
    """Generator form of dataset.py:154-215: per sample  decode -> boxes / image_size -> rescale_image ->
    build_labels -> [sample, hflip(sample)] batch -> preprocess_image.  Everything after the host loader runs on the
    device (sample_batch).  `shuffle` / `augment` are accepted for signature parity (the reference's augment_sample is a
    TODO stub; shuffling belongs to the loader here)."""
    dev = torch.device(device)
    for sample in data_loader:
        image = torch.from_numpy(np.ascontiguousarray(sample['image'])).to(dev)                # uint8 or float [H,W,3]
        h, w = int(image.shape[0]), int(image.shape[1])
        boxes = np.asarray(sample['boxes'], np.float32) / np.asarray([h, w, h, w], np.float32)   # dataset.py:163
        ids = torch.from_numpy(np.asarray(sample['class_ids'], np.int32)).to(dev)[None]
        batch = sample_batch(image, torch.from_numpy(boxes).to(dev)[None], ids, levels, data_loader.num_classes, scale=scale,
                             normalize=normalize)
        batch.update(boxes=boxes, class_ids=sample['class_ids'])
        yield batch


class DeviceFeed(object):
    """train_input_fn (train.py:190-202) for the hipGraph train step: a NEW sample every step without leaving the graph.

      loader thread -> pinned host slots -> async H2D on a copy stream -> STATIC device buffers (raw image, boxes, class
      ids, object count) -> `features()` = sample_batch on those buffers, captured INSIDE segment A of the step's graph.

    Protocol with train.Trainer (input_fn=feed): `stage()` before segment A is launched (the copy stream uploads the next
    sample once the previous segment A has consumed the buffers; the main stream waits for the upload), `features()` inside
    the segment, `consumed()` right after it.  The sample stream is the loader's, in order: a run through DeviceFeed sees
    exactly the samples `build_dataset` would yield.  A sample whose image size or object capacity differs from the
    buffers' gets new buffers, and `shape_key` changes -- the trainer keeps one captured graph per key."""

    def __init__(self, data_loader, levels, scale=None, device='cuda', max_obj=32, normalize=True, prefetch=3):
        import queue
        import threading
        self.levels, self.scale, self.normalize = levels, scale, normalize
        self.num_classes = data_loader.num_classes
        self.device = torch.device(device)
        self.max_obj = int(max_obj)
        self._it = iter(data_loader)
        self._slots = int(prefetch)
        self._free = queue.Queue()
        self._ready = queue.Queue()
        for _ in range(self._slots):
            self._free.put(None)                    # a slot is (host tensors, upload-done event); created lazily per shape
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._consumed = None                       # event: segment A of the previous step has read the static buffers
        self._static = None
        self._statics = {}                          # shape key -> static device buffers
        self.shape_key = None
        self.last_sample = None                     # host-side boxes / class ids of the staged sample (evaluation, logging)
        self.samples_staged = 0
        self._stop = False
        self._error = None
        self._thread = threading.Thread(target=self._produce, name='rn-device-feed', daemon=True)
        self._started = False

    def start(self):
        """Start the loader thread (stage() does it on first use; call it earlier to prefetch)."""
        if not self._started:
            self._started = True
            self._thread.start()

    # -- host side: loader thread
    def _produce(self):
        try:
            for sample in self._it:
                slot = self._free.get()
                if self._stop:
                    return
                if slot is not None and slot[1] is not None:
                    slot[1].synchronize()           # the upload that last used this slot's pinned memory is done
                img = np.ascontiguousarray(sample['image'])
                ids = np.asarray(sample['class_ids'], np.int32).reshape(-1)
                h, w = int(img.shape[0]), int(img.shape[1])
                boxes = (np.asarray(sample['boxes'], np.float32).reshape(-1, 4) / np.asarray([h, w, h, w], np.float32))   # dataset.py:163
                cap = max(self.max_obj, -(-len(ids) // 32) * 32)
                host = slot[0] if slot is not None else None
                if host is None or tuple(host['image'].shape) != img.shape or host['image'].dtype != torch.from_numpy(img).dtype \
                        or host['boxes'].shape[1] != cap:
                    host = {'image': torch.empty(img.shape, dtype=torch.from_numpy(img).dtype).pin_memory(),
                            'boxes': torch.zeros((1, cap, 4), dtype=torch.float32).pin_memory(),
                            'ids': torch.zeros((1, cap), dtype=torch.int32).pin_memory(),
                            'nobj': torch.zeros((1,), dtype=torch.int32).pin_memory()}
                host['image'].numpy()[...] = img
                host['boxes'].zero_(); host['ids'].zero_()
                host['boxes'].numpy()[0, :len(ids)] = boxes
                host['ids'].numpy()[0, :len(ids)] = ids
                host['nobj'][0] = len(ids)
                self._ready.put((host, {'boxes': boxes, 'class_ids': ids, 'image_hw': (h, w)}))
            self._ready.put(None)
        except BaseException as e:                  # surfaces in stage() on the training thread
            self._error = e
            self._ready.put(None)

    def close(self):
        self._stop = True
        for _ in range(self._slots + 1):
            self._free.put(None)

    # -- device side
    def stage(self):
        """Upload the next sample into the static buffers (copy stream) and make the current stream wait for it.  Returns
        the shape key (changes when new static buffers had to be made: the caller captures a new graph for it)."""
        self.start()
        item = self._ready.get()
        if item is None:
            if self._error is not None:
                raise self._error
            raise StopIteration
        host, info = item
        key = (tuple(host['image'].shape), str(host['image'].dtype), int(host['boxes'].shape[1]))
        cur = torch.cuda.current_stream(self.device)
        if key != self.shape_key:
            if key not in self._statics:
                self._statics[key] = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in host.items()}
            self._static, self.shape_key = self._statics[key], key
        cs = self._copy_stream
        if self._consumed is not None:
            cs.wait_event(self._consumed)           # the previous segment A no longer reads the buffers
        else:
            cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            for k, v in host.items():
                self._static[k].copy_(v, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cs)
        cur.wait_event(done)
        self._free.put((host, done))
        self.last_sample = info
        self.samples_staged += 1
        return key

    def consumed(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._consumed = ev

    def features(self):
        """The step's features from the static buffers (inside the captured segment when the trainer runs graphs)."""
        s = self._static
        return sample_batch(s['image'], s['boxes'], s['ids'], self.levels, self.num_classes, scale=self.scale,
                            num_obj=s['nobj'], normalize=self.normalize)

    __call__ = features
