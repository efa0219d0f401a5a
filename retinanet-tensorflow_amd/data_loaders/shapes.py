"""Synthetic 'shapes' dataset (stands in for reference data_loaders/shapes.py:11-55; same classes, same sample
dict, same pixel-unit boxes [y - s, x - s, y + s, x + s]).  The reference draws with cv2 and writes PNG files that the
tf.data pipeline reads back; here the sample carries the uint8 image itself under 'image' (no cv2 / PNG codec in
this image, and dataset.build_dataset accepts either key)."""
import numpy as np

from data_loaders.base import Base


class Shapes(Base):
    def __init__(self, path=None, num_samples=1 << 30, image_size=(256, 256), seed=0):
        self._path = path                      # kept for signature parity; nothing is written
        self._num_samples = num_samples
        self._image_size = tuple(image_size)
        self._class_names = ['square', 'triangle', 'circle']
        self._rng = np.random.default_rng(seed)
        self._drawn = 0                        # position in the sample stream (advanced by every iterator and by skip)

    @property
    def class_names(self):
        return self._class_names

    @property
    def num_classes(self):
        return len(self._class_names)

    def skip(self, n):
        """Advance the sample stream by n samples (resuming a run: the checkpoint stores how many were drawn).  Draws only --
        the same random numbers `__iter__` consumes, nothing is rendered -- so resuming a long run costs microseconds per
        skipped sample.  The stream itself is endless; `num_samples` bounds ONE pass (one iterator), as the reference's loader
        yields num_samples items per `__iter__` (data_loaders/shapes.py:46-55)."""
        h, w = self._image_size
        for _ in range(int(n)):
            self._draw(h, w)

    def _draw(self, h, w):
        """The random numbers of one sample, in the order __iter__ has always drawn them."""
        rng = self._rng
        self._drawn += 1
        background = rng.integers(0, 255, (1, 1, 3)).astype(np.uint8)
        items = []
        for _ in range(int(rng.integers(1, 5))):
            shape = int(rng.integers(0, 3))
            color = rng.integers(0, 255, 3).astype(np.uint8)
            s = int(rng.integers(12, max(13, min(h, w) // 5)))
            y, x = int(rng.integers(s, h - s)), int(rng.integers(s, w - s))
            items.append((shape, color, s, y, x))
        return background, items

    def __iter__(self):
        h, w = self._image_size
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(self._num_samples):             # per ITERATOR: a second iter() yields the next num_samples of the stream
            background, items = self._draw(h, w)
            image = np.ones((h, w, 3), np.uint8) * background
            boxes, class_ids = [], []
            for shape, color, s, y, x in items:
                if shape == 0:
                    mask = (np.abs(yy - y) <= s) & (np.abs(xx - x) <= s)
                elif shape == 1:                                   # upward triangle inscribed in the box
                    mask = (yy >= y - s) & (yy <= y + s) & (np.abs(xx - x) * 2 <= (yy - (y - s)))
                else:
                    mask = (yy - y) ** 2 + (xx - x) ** 2 <= s * s
                image[mask] = color
                boxes.append([y - s, x - s, y + s, x + s])
                class_ids.append(shape)
            yield self.check_sample({'image': image, 'class_ids': np.asarray(class_ids, np.int32),
                                     'boxes': np.asarray(boxes, np.float32)})
