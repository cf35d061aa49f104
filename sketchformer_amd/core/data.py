"""Data-loader contract of the reference (core/data.py:25-254): ``DataLoader(hps, dir)``, ``.hps`` dict,
``.tokenizer``, ``.n_classes``, ``.n_samples``, ``batch_iterator(split, batch_size, stop_at_end_of_split)``
yielding ``(x, y)`` with x (B,L) int64 / (B,L,5) float and y (B,1) int64, ``get_n_samples_from``,
``get_all_data_from``.  Host-side Python: nothing here is accelerated;
chunk files are loaded by a background thread one megabatch ahead like the reference does.
"""
import threading
from abc import ABCMeta, abstractmethod

import numpy as np


class DatasetSplit(object):
    def __init__(self, name, filepaths):
        self.name, self.filepaths = name, list(filepaths)
        self.n_files = len(self.filepaths)
        self.file_order, self.file_cursor = list(range(self.n_files)), 0
        self.current, self.next, self.thread = None, None, None
        self.order, self.cursor = [], 0


class BaseDataLoader(object, metaclass=ABCMeta):
    def __init__(self, hps, data_directory):
        if not hasattr(self, "name"):
            raise Exception("You must give your data loader a reference name")
        self.hps = hps if isinstance(hps, dict) else dict(hps.values())
        self.data_directory = data_directory
        self.splits = {s.name: s for s in self.get_data_splits()}
        for name, split in self.splits.items():
            if split.n_files:
                split.file_order = list(self.reshuffle_file_indices(name, split.filepaths))
                self._start_load(name)

    @classmethod
    def parse_hparams(cls, params):
        hps = cls.default_hparams()
        if params is not None:
            hps = hps.parse(params)
        return hps

    # ---- children implement
    @classmethod
    @abstractmethod
    def default_hparams(cls):
        pass

    @abstractmethod
    def get_data_splits(self):
        pass

    @abstractmethod
    def load_next_megabatch(self, split_name, selected_file):
        """must call set_future_data_for_split(split_name, {'x': ..., 'y': ...})"""

    def reshuffle_file_indices(self, split_name, filenames):
        return list(range(len(filenames)))

    def reshuffle_sample_indices(self, split_name, data):
        return list(range(len(data["x"])))

    def get_sample(self, data, idx):
        return data["x"][idx], np.expand_dims(data["y"][idx], axis=-1)

    # ---- megabatch plumbing
    def set_future_data_for_split(self, split_name, data):
        self.splits[split_name].next = data

    def _start_load(self, split_name):
        split = self.splits[split_name]
        path = split.filepaths[split.file_order[split.file_cursor]]
        split.thread = threading.Thread(target=self.load_next_megabatch, args=(split_name, path), daemon=True)
        split.thread.start()

    def _swap(self, split_name):
        """Make the preloaded megabatch current; returns True when the whole split has been seen."""
        split = self.splits[split_name]
        split.thread.join()
        split.current = split.next
        split.order = list(self.reshuffle_sample_indices(split_name, split.current))
        split.cursor = 0
        split.file_cursor += 1
        done = split.file_cursor == split.n_files
        if done:
            split.file_cursor = 0
            split.file_order = list(self.reshuffle_file_indices(split_name, split.filepaths))
        if split.n_files > 1:
            self._start_load(split_name)
        else:
            split.thread = threading.Thread(target=lambda: None)
            split.thread.start()
        return done

    def _ready(self, split_name):
        if self.splits[split_name].current is None:
            self._swap(split_name)

    def batch_iterator(self, split_name, batch_size, stop_at_end_of_split):
        self._ready(split_name)
        split = self.splits[split_name]
        while True:
            xs, ys = [], []
            while len(xs) < batch_size:
                if split.cursor >= len(split.order):
                    finished = self._swap(split_name)
                    if finished and stop_at_end_of_split:
                        if xs:
                            yield np.array(xs), np.array(ys)
                        return
                x, y = self.get_sample(split.current, split.order[split.cursor])
                split.cursor += 1
                xs.append(x)
                ys.append(y)
            yield np.array(xs), np.array(ys)

    def get_n_samples_from(self, split_name, n, shuffled=False, seeded=False):
        self._ready(split_name)
        data = self.splits[split_name].current
        idx = np.arange(len(data["x"]))
        if shuffled:
            rng = np.random.RandomState(14) if seeded else np.random
            idx = rng.permutation(len(idx))
        idx = idx[:n]
        return np.array([data["x"][i] for i in idx]), np.array([np.expand_dims(data["y"][i], -1) for i in idx])

    def get_all_data_from(self, split_name):
        xs, ys = [], []
        for x, y in self.batch_iterator(split_name, 256, stop_at_end_of_split=True):
            xs.append(x)
            ys.append(y)
        return np.concatenate(xs), np.concatenate(ys)
