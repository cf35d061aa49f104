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
        """core/data.py:147-176 of the reference: yields [x, y] of batch_size samples in the shuffled order of the
        current megabatch, crossing into the next megabatch when one runs out.  Same sequence of samples; a batch is
        gathered with one fancy-index per megabatch piece instead of a Python loop over samples."""
        self._ready(split_name)
        split = self.splits[split_name]
        while True:
            xs, ys, have = [], [], 0
            while have < batch_size:
                if split.cursor >= len(split.order):
                    finished = self._swap(split_name)
                    if finished and stop_at_end_of_split:
                        if have:
                            yield np.concatenate(xs, axis=0), np.concatenate(ys, axis=0)
                        return
                take = min(batch_size - have, len(split.order) - split.cursor)
                x, y = self.get_samples(split.current, split.order[split.cursor:split.cursor + take])
                split.cursor += take
                have += take
                xs.append(x)
                ys.append(y)
            yield (xs[0], ys[0]) if len(xs) == 1 else (np.concatenate(xs, axis=0), np.concatenate(ys, axis=0))

    def get_samples(self, data, idx):
        """Vectorised get_sample: rows idx of the megabatch -> (x (n, ...), y (n, 1))."""
        idx = np.asarray(idx, dtype=np.int64)
        x, y = data["x"], data["y"]
        if isinstance(x, np.ndarray) and x.dtype != object and isinstance(y, np.ndarray) and y.dtype != object:
            return x[idx], np.expand_dims(y[idx], axis=-1)
        pairs = [self.get_sample(data, i) for i in idx]
        return np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])

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
