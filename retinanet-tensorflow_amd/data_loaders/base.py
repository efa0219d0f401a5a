"""The loader protocol `dataset.build_dataset` consumes (the role of reference data_loaders/base.py).

A loader names its classes and yields samples; nothing else is assumed about where the data comes from.

    loader.class_names   sequence of str, index = class id
    loader.num_classes   len(class_names)
    iter(loader)         samples: {'image' (uint8 / float [H,W,3]) or 'image_file', 'class_ids' [O] int, 'boxes' [O,4] float
                         pixel corners (y1, x1, y2, x2)}

`Base` spells that protocol out as an abstract class and adds `check_sample`, which the synthetic loaders use to
validate what they emit (shapes, dtypes, box ordering) before it reaches the device-side label kernels.
"""
import abc

import numpy as np


class Base(abc.ABC):
    @property
    @abc.abstractmethod
    def class_names(self):
        """Class names; the position of a name is its class id."""

    @property
    def num_classes(self):
        return len(self.class_names)

    @abc.abstractmethod
    def __iter__(self):
        """Yield sample dicts (see the module docstring)."""

    def check_sample(self, sample):
        """Raise ValueError if `sample` does not follow the protocol; returns it unchanged otherwise."""
        if 'image' not in sample and 'image_file' not in sample:
            raise ValueError("sample has neither 'image' nor 'image_file'")
        ids = np.asarray(sample['class_ids'])
        boxes = np.asarray(sample['boxes'], np.float64).reshape(-1, 4)
        if ids.ndim != 1 or len(ids) != len(boxes):
            raise ValueError("class_ids %s and boxes %s disagree" % (ids.shape, boxes.shape))
        if len(ids) and (ids.min() < 0 or ids.max() >= self.num_classes):
            raise ValueError("class id outside [0, %d)" % self.num_classes)
        if np.any(boxes[:, 2] < boxes[:, 0]) or np.any(boxes[:, 3] < boxes[:, 1]):
            raise ValueError("boxes must be (y1, x1, y2, x2) with y1 <= y2 and x1 <= x2")
        if 'image' in sample:
            img = np.asarray(sample['image'])
            if img.ndim != 3 or img.shape[2] != 3:
                raise ValueError("image must be [H, W, 3], got %s" % (img.shape,))
        return sample
