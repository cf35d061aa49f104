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


def augment_strokes(strokes, prob, urnd):
    """utils/sketch.py:127-149 (random point dropping inside a stroke) without the Python loop.  ``urnd``: one uniform
    draw per point, in order (the reference calls np.random.rand() once per point whether it is used or not).
    The loop's state has a closed form: its `prev_stroke[2]` always equals the pen bit of the previous point (a dropped
    point has pen 0 and so has the kept point it is merged into) and `count` is the distance to the last point where
    either pen bit was 1 - so "dropped" is known per point, and the offsets of a run of dropped points are added, in
    order and in the array's precision, to the kept point before them."""
    s = np.asarray(strokes)
    n = len(s)
    if n == 0:
        return np.array([])
    pen = s[:, 2]
    prev = np.concatenate([[1], pen[:-1]])
    reset = (pen == 1) | (prev == 1)
    idx = np.arange(n)
    last_reset = np.maximum.accumulate(np.where(reset, idx, -1))        # index 0 always resets (prev = 1)
    count = idx - last_reset
    drop = (pen == 0) & (prev == 0) & (count > 2) & (np.asarray(urnd) < prob)
    keep = ~drop
    out = s[keep].copy()
    if drop.any():
        owner = np.cumsum(keep) - 1                  # kept point that absorbs a dropped one
        run = idx - np.maximum.accumulate(np.where(keep, idx, -1))      # 1, 2, ... inside a run of dropped points
        for k in range(1, int(run.max()) + 1):       # k-th dropped point of every run: sequential adds, like the loop
            sel = drop & (run == k)
            out[owner[sel], 0] += s[sel, 0]
            out[owner[sel], 1] += s[sel, 1]
    return out


def convert_to_absolute(sketch):
    """utils/sketch.py:169-175: offsets -> running positions (row 0 kept, pen states kept); the running sum is taken in the
    array's own precision, one row after the other, like the reference's loop."""
    out = np.array(sketch, copy=True)
    out[:, :2] = np.cumsum(sketch[:, :2], axis=0, dtype=sketch.dtype)
    return out


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

    # ---- preprocessing.  `preprocess` = the whole chunk at once with array operations (a training rank consumes
    # ~21k sketches/s on one MI355X; the per-sketch Python loop of the reference delivers ~14k/s per core);
    # `preprocess_per_sketch` = the reference's loop, kept as the definition the fast path is tested against.
    def preprocess(self, data, augment=False):
        if (self.hps["shuffle_stroke"] or self.hps["use_absolute_strokes"] or len(data) == 0
                or not (self.hps["use_continuous_data"] or isinstance(self.tokenizer, (GridTokenizer, Tokenizer)))
                or min(len(s) for s in data) == 0):
            return self.preprocess_per_sketch(data, augment)
        out = [self._preprocess_block(data[i:i + 512], augment) for i in range(0, len(data), 512)]   # cache-sized blocks
        return np.concatenate(out, axis=0)

    def _preprocess_block(self, data, augment):
        """One block of sketches as padded (N, T) planes x / y / pen (reductions and scans run along the contiguous
        axis).  Every floating-point step repeats the per-sketch code's operations in the same order and precision."""
        L, N = self.hps["max_seq_len"], len(data)
        do_aug = augment and self.hps["augment_stroke_prob"] > 0 and self.hps["use_continuous_data"]
        if do_aug:
            # scale + point dropping change the lengths: done per sketch (array operations inside a sketch, the random
            # stream consumed in the per-sketch order: two draws, then one per point), the rest of the pipeline below
            # runs on the block
            data = [self._augment_sketch(np.array(np.clip(s, -self.limit, self.limit), dtype=np.float32)) for s in data]
            if min(len(s) for s in data) == 0:
                return self.preprocess_per_sketch_from(data)
        lens = np.fromiter((len(s) for s in data), dtype=np.int64, count=N)
        T = int(lens.max())
        rows = np.repeat(np.arange(N), lens)
        cols = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
        flat = np.concatenate([np.asarray(s)[:, :3] for s in data], axis=0)
        if not do_aug:                     # (augmented sketches were clamped before the augmentation, like the reference)
            flat = np.clip(flat, -self.limit, self.limit)
        flat = flat.astype(np.float32)
        X = np.zeros((N, T), dtype=np.float32); Y = np.zeros((N, T), dtype=np.float32); Pn = np.zeros((N, T), dtype=np.float32)
        X[rows, cols], Y[rows, cols], Pn[rows, cols] = flat[:, 0], flat[:, 1], flat[:, 2]
        ar = np.arange(T)[None, :]
        valid = ar < lens[:, None]
        # normalise by the larger side of the bounding box of the absolute path (origin included): the bounds are summed in
        # float64 (utils/sketch.py:41-44 converts every offset with float()), the division `sketch[:, :2] /= max_dim`
        # (dataloaders/distributed_stroke3.py:103-104) is a float32 operation - the Python-float divisor is rounded to
        # the array's precision first
        X64, Y64 = X.astype(np.float64), Y.astype(np.float64)
        cx, cy = np.cumsum(X64, axis=1), np.cumsum(Y64, axis=1)          # sequential adds per row, like the loop per sketch
        dx = np.maximum(cx.max(axis=1), 0.0) - np.minimum(cx.min(axis=1), 0.0)
        dy = np.maximum(cy.max(axis=1), 0.0) - np.minimum(cy.min(axis=1), 0.0)
        div = np.maximum(np.maximum(dx, dy), 1.0)[:, None].astype(np.float32)
        X, Y = X / div, Y / div
        if self.hps["use_continuous_data"]:
            n = np.minimum(lens, L)
            Tc = min(T, L)
            keep = (np.arange(Tc)[None, :] < n[:, None])
            out = np.zeros((N, L, 5), dtype=float)
            out[:, :Tc, 0] = np.where(keep, X[:, :Tc], 0.0)
            out[:, :Tc, 1] = np.where(keep, Y[:, :Tc], 0.0)
            out[:, :Tc, 3] = np.where(keep, Pn[:, :Tc], 0.0)
            out[:, :Tc, 2] = np.where(keep, 1 - Pn[:, :Tc], 0.0)
            out[:, :, 4] = np.arange(L)[None, :] >= n[:, None]
            out[:, -1, 4] = 1
            return out
        tok = self.tokenizer
        if isinstance(tok, GridTokenizer):
            one, r32 = np.float32(1), np.float32(tok.r)               # float32 throughout, like GridTokenizer.encode
            cx = ((np.cumsum(X, axis=1, dtype=np.float32) + one) * r32).astype(np.int64)
            cy = ((np.cumsum(Y, axis=1, dtype=np.float32) + one) * r32).astype(np.int64)
            cx[cx == tok.resolution] = tok.resolution - 1
            cy[cy == tok.resolution] = tok.resolution - 1
            ids = cx + cy * tok.resolution + 1
        else:                                                            # k-means dictionary: nearest centre per offset
            ids = np.zeros((N, T), dtype=np.int64)
            ids[rows, cols] = tok.nearest_center(X[rows, cols], Y[rows, cols]) + 1
        lift = (Pn == 1) & valid
        nlift = lift.sum(axis=1)
        before = np.cumsum(lift, axis=1) - lift
        out = np.full((N, L), tok.PAD, dtype=np.int64)
        out[:, 0] = tok.SOS
        if isinstance(tok, GridTokenizer):
            # points after the last pen lift are dropped unless the sketch has no lift at all (then: all points, one SEP)
            last = np.where(nlift > 0, T - 1 - np.argmax(lift[:, ::-1], axis=1), lens - 1)
            incl = valid & (ar <= last[:, None])
            sep_after = np.where(nlift[:, None] > 0, lift, ar == (lens - 1)[:, None])
            eos = 1 + (last + 1) + np.maximum(nlift, 1)
        else:
            incl, sep_after = valid, lift
            eos = 1 + lens + nlift
        pos = 1 + ar + before
        r, t = np.nonzero(incl & (pos < L))
        out[r, pos[r, t]] = ids[r, t]
        r, t = np.nonzero(sep_after & incl & (pos + 1 < L))
        out[r, pos[r, t] + 1] = tok.SEP
        r = np.nonzero(eos < L)[0]
        out[r, eos[r]] = tok.EOS
        return out

    def preprocess_per_sketch(self, data, augment=False):
        """The reference's loop (dataloaders/distributed_stroke3.py:90-125), one sketch at a time."""
        out = []
        for sketch in data:
            sketch = np.array(np.clip(sketch, -self.limit, self.limit), dtype=np.float32)
            if augment:
                sketch = self._augment_sketch(sketch)
            out.append(sketch)
        return self.preprocess_per_sketch_from(out)

    def preprocess_per_sketch_from(self, data):
        """Everything after clamping / augmentation, per sketch (float32 stroke-3 arrays in)."""
        out = []
        for sketch in data:
            if len(sketch) == 0:        # augmentation cannot empty a sketch, an empty input fails like the reference
                raise IndexError("empty sketch")
            min_x, max_x, min_y, max_y = get_bounds(sketch)
            # a float32 division: the reference's bounds are Python floats, which numpy rounds to the array's precision
            sketch[:, :2] /= np.float32(max([max_x - min_x, max_y - min_y, 1]))
            if self.hps["shuffle_stroke"]:
                # the reference calls utils.tu_sketch_tools.strokes_to_lines here (:107-110), a module its repository does
                # not contain: the option raises there as well
                raise NotImplementedError("shuffle_stroke needs utils.tu_sketch_tools, which the reference does not ship")
            if self.hps["use_absolute_strokes"]:
                sketch = convert_to_absolute(sketch)
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
        """dataloaders/distributed_stroke3.py:155-160: random_scale (:127-137) then utils.sketch.augment_strokes; the
        random stream is consumed exactly like there: two draws for the scale factors, then one per point."""
        if self.hps["augment_stroke_prob"] > 0 and self.hps["use_continuous_data"]:
            e = self.hps["random_scale_factor"]
            res = np.copy(sketch)
            res[:, 0] *= np.float32((np.random.random() - 0.5) * 2 * e + 1.0)
            res[:, 1] *= np.float32((np.random.random() - 0.5) * 2 * e + 1.0)
            return augment_strokes(res, self.hps["augment_stroke_prob"], np.random.random(size=len(res)))
        return sketch
