"""``stroke3-distributed``: the reference's chunked QuickDraw loader (dataloaders/distributed_stroke3.py:10-204).
Same hparams, same on-disk format (``meta*.npz`` with n_classes / n_samples_train / class_names / std and
``train*/valid*/test*.npz`` chunks holding object arrays x (stroke-3) and y), same per-sketch pipeline:
clamp to +-1000, optional augmentation (continuous mode), normalise by the larger bounding-box side,
tokenise (grid or k-means dictionary), truncate to max_seq_len, pad / convert to stroke-5.
"""
import glob
import os

import numpy as np

from ..core.data import BaseDataLoader, DatasetSplit
from ..utils import hparams as hp
from ..utils.tokenizer import GridTokenizer, Tokenizer


def get_bounds(stroke3, factor=1.0):
    """(min_x, max_x, min_y, max_y) of the absolute pen path starting at the origin (utils/sketch.py:31-50)."""
    xy = np.cumsum(np.asarray(stroke3[:, :2], dtype=np.float64) / factor, axis=0)
    xs = np.concatenate([[0.0], xy[:, 0]])
    ys = np.concatenate([[0.0], xy[:, 1]])
    return xs.min(), xs.max(), ys.min(), ys.max()


class DistributedStroke3DataLoader(BaseDataLoader):
    name = "stroke3-distributed"

    @classmethod
    def default_hparams(cls):
        return hp.HParams(
            max_seq_len=200, shuffle_stroke=False, token_type="dictionary", use_continuous_data=False,
            use_absolute_strokes=False, tokenizer_dict_file="prep_data/sketch_token/token_dict.pkl",
            tokenizer_resolution=100, augment_stroke_prob=0.1, random_scale_factor=0.1)

    def __init__(self, hps, data_directory):
        self.limit = 1000
        h = hps if isinstance(hps, dict) else dict(hps.values())
        if not h["use_continuous_data"] and h["token_type"] == "dictionary":
            self.tokenizer = Tokenizer(h["tokenizer_dict_file"], max_seq_len=0)
        elif not h["use_continuous_data"] and h["token_type"] == "grid":
            self.tokenizer = GridTokenizer(resolution=100)
        meta_file = [f for f in glob.glob("{}/*".format(data_directory)) if os.path.basename(f).startswith("meta")][0]
        meta = np.load(meta_file, allow_pickle=True)
        self.n_classes = int(meta["n_classes"])
        self.n_samples = int(meta["n_samples_train"])
        self.class_names = meta["class_names"]
        self.scale_factor = float(meta["std"])
        super().__init__(hps, data_directory)

    def get_data_splits(self):
        def files(prefix):
            return sorted(f for f in glob.glob("{}/*".format(self.data_directory))
                          if os.path.basename(f).startswith(prefix))
        return [DatasetSplit("train", files("train")), DatasetSplit("test", files("test")),
                DatasetSplit("valid", files("valid"))]

    def reshuffle_file_indices(self, split_name, filenames):
        return np.random.permutation(len(filenames)) if split_name == "train" else list(range(len(filenames)))

    def reshuffle_sample_indices(self, split_name, data):
        return np.random.permutation(len(data["x"])) if split_name == "train" else list(range(len(data["x"])))

    def load_next_megabatch(self, split_name, selected_file):
        loaded = np.load(selected_file, allow_pickle=True)
        self.set_future_data_for_split(split_name, {"x": self.preprocess(loaded["x"], augment=split_name == "train"),
                                                    "y": loaded["y"]})

    # ---- per-sketch pipeline
    def preprocess(self, data, augment=False):
        out = []
        for sketch in data:
            sketch = np.array(np.clip(sketch, -self.limit, self.limit), dtype=np.float32)
            if augment:
                sketch = self._augment_sketch(sketch)
            min_x, max_x, min_y, max_y = get_bounds(sketch)
            sketch[:, :2] /= max([max_x - min_x, max_y - min_y, 1])
            if self.hps["shuffle_stroke"] or self.hps["use_absolute_strokes"]:
                raise NotImplementedError("shuffle_stroke / use_absolute_strokes are not implemented")
            if not self.hps["use_continuous_data"]:
                sketch = self.tokenizer.encode(sketch)
            if len(sketch) > self.hps["max_seq_len"]:
                sketch = sketch[:self.hps["max_seq_len"]]
            sketch = self._cap_pad_and_convert_sketch(sketch)
            if not self.hps["use_continuous_data"]:
                sketch = np.squeeze(sketch)
            out.append(sketch)
        return np.array(out)

    def _cap_pad_and_convert_sketch(self, sketch):
        L, n = self.hps["max_seq_len"], len(sketch)
        if not self.hps["use_continuous_data"]:
            conv = np.ones((L, 1), dtype=int) * self.tokenizer.PAD
            conv[:n, 0] = sketch
        else:
            conv = np.zeros((L, 5), dtype=float)
            conv[:n, 0:2] = sketch[:, 0:2]
            conv[:n, 3] = sketch[:, 2]
            conv[:n, 2] = 1 - sketch[:, 2]
            conv[n:, 4] = 1
            conv[-1:, 4] = 1
        return conv

    def _augment_sketch(self, sketch):
        if self.hps["augment_stroke_prob"] > 0 and self.hps["use_continuous_data"]:
            e = self.hps["random_scale_factor"]
            res = np.copy(sketch)
            res[:, 0] *= (np.random.random() - 0.5) * 2 * e + 1.0
            res[:, 1] *= (np.random.random() - 0.5) * 2 * e + 1.0
            return res          # point-dropping augmentation (utils/sketch.py:127-149) is not ported
        return sketch
