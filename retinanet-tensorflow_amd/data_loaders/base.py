"""Loader protocol (drop-in for reference data_loaders/base.py:1-11): a loader exposes `class_names`,
`num_classes` and iterates over samples {'image' | 'image_file', 'class_ids' [O], 'boxes' [O,4] in pixels}."""


class Base(object):
    @property
    def class_names(self):
        raise NotImplementedError

    @property
    def num_classes(self):
        raise NotImplementedError

    def __iter__(self):
        raise NotImplementedError
