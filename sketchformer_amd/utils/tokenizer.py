"""Stroke-3 tokenizers with the id conventions of the reference (utils/tokenizer.py):
PAD=0, SEP=n+1, SOS=n+2, EOS=n+3, VOCAB_SIZE=n+4 with n = dictionary size (Tokenizer :16-101)
or n = resolution^2 (GridTokenizer :104-198).  Vectorised numpy implementations; behaviour
pinned by tests/golden/reference_goldens.json ("grid_tokenizer", "grid_tokenizer_ids").
"""
import pickle

import numpy as np


def _pad_to(out, max_seq_len, seq_len, PAD, SEP, EOS):
    if max_seq_len:
        if len(out) < max_seq_len:
            out = out + [PAD] * (max_seq_len - len(out))
        else:
            out = out[:max_seq_len]
            out[-2:] = [SEP, EOS]
    if len(out) < seq_len:
        out = out + [PAD] * (seq_len - len(out))
    return np.array(out)


class GridTokenizer(object):
    """Each absolute pen position is mapped to the id of its cell in a (2r x 2r) grid over [-1,1]^2."""

    def __init__(self, resolution=100, max_seq_len=0):
        self.max_seq_len = max_seq_len
        self.r = int(resolution / 2)
        self.resolution = 2 * self.r
        self.half_pixel = 1 / self.resolution
        self.PAD = 0
        self.SEP = self.resolution ** 2 + 1
        self.SOS = self.SEP + 1
        self.EOS = self.SEP + 2
        self.VOCAB_SIZE = self.resolution ** 2 + 4

    def encode(self, stroke3, seq_len=0):
        # utils/skt_tools.py:80-88 accumulates the offsets one by one in the dtype of the input (float32 from the loader,
        # dataloaders/distributed_stroke3.py:96) and utils/tokenizer.py:139-142 bins in that dtype too: np.cumsum is the
        # same sequential sum, and staying in the input precision keeps points on cell borders in the reference's cell
        s = np.asarray(stroke3)
        if s.dtype not in (np.float32, np.float64):
            s = s.astype(np.float64)
        xy = np.cumsum(s[:, :2], axis=0, dtype=s.dtype)       # absolute positions (strokes_to_lines, scale 1)
        cell = np.int64((xy + s.dtype.type(1)) * s.dtype.type(self.r))
        cell[cell == self.resolution] = self.resolution - 1   # upper bound lands in the last cell
        ids = (cell[:, 0] + cell[:, 1] * self.resolution + 1).tolist()
        out = [self.SOS]
        ends = np.where(s[:, 2] == 1)[0]
        start = 0
        for e in ends:
            out.extend(ids[start:e + 1])
            out.append(self.SEP)
            start = e + 1
        if len(ends) == 0:                                    # no pen lift: one open stroke, still closed by SEP
            out.extend(ids)
            out.append(self.SEP)
        out.append(self.EOS)
        return _pad_to(out, self.max_seq_len, seq_len, self.PAD, self.SEP, self.EOS)

    def decode(self, seqs):
        if len(seqs) > 0 and isinstance(seqs[0], (list, tuple, np.ndarray)):
            return [self.decode_single(np.squeeze(s)) for s in seqs]
        return self.decode_single(seqs)

    def decode_single(self, tokens):
        lines, line = [], []
        for t in tokens:
            t = int(t)
            if 0 < t < self.SEP:
                line.append([((t - 1) % self.resolution) / self.r - 1 + self.half_pixel,
                             ((t - 1) // self.resolution) / self.r - 1 + self.half_pixel])
            elif t == self.SEP and line:
                lines.append(line)
                line = []
            elif t == self.EOS:
                break
        if line:
            lines.append(line)
        if not lines:
            lines = [[[0.0, 0.0]]]
        pts = np.array([p + [0.0] for ln in lines for p in ln], dtype=np.float64)
        last = np.cumsum([len(ln) for ln in lines]) - 1
        pts[last, 2] = 1.0
        pts[1:, :2] -= pts[:-1, :2].copy()                    # back to offsets (first row keeps its absolute position)
        return pts


class Tokenizer(object):
    """k-means dictionary tokenizer: each (dx,dy) offset -> nearest centroid id + 1."""

    def __init__(self, dict_path, max_seq_len=0):
        self.max_seq_len = max_seq_len
        with open(dict_path, "rb") as f:
            self.dict = pickle.load(f)
        self.centers = np.asarray(self.dict.cluster_centers_, dtype=np.float64)
        n = self.centers.shape[0]
        self.PAD, self.SEP, self.SOS, self.EOS, self.VOCAB_SIZE = 0, n + 1, n + 2, n + 3, n + 4

    def nearest_center(self, x, y):
        """Index of the closest dictionary centre for every offset (x[i], y[i]) (what KMeans.predict computes);
        squared distances (x-cx)^2 + (y-cy)^2 in float64, first minimum wins; chunked to bound the temporaries."""
        x, y = np.asarray(x), np.asarray(y)
        if hasattr(self.dict, "predict") and x.shape[0]:
            # the pickled dictionary is the reference's sklearn KMeans: one predict call for all points of a block
            # (utils/tokenizer.py:43 calls it per sketch); 40x faster than the numpy fallback below.  sklearn computes in
            # the dtype of the fitted centres - float32 for a dictionary made by prep_data/sketch_token/
            # create_token_dict.py:52 from float32 offsets, the dtype the loader's offsets have too (current sklearn
            # refuses a float32 / float64 mix, older versions upcast both) - so the points are cast to that dtype
            pts = np.stack([x, y], axis=1).astype(np.asarray(self.dict.cluster_centers_).dtype)
            return np.asarray(self.dict.predict(pts), dtype=np.int64)
        x, y = x.astype(np.float64), y.astype(np.float64)
        out = np.empty(x.shape[0], dtype=np.int64)
        cx, cy = self.centers[None, :, 0], self.centers[None, :, 1]
        for i in range(0, x.shape[0], 2048):
            dx, dy = x[i:i + 2048, None] - cx, y[i:i + 2048, None] - cy
            out[i:i + 2048] = (dx * dx + dy * dy).argmin(1)
        return out

    def encode(self, stroke3, seq_len=0):
        s = np.asarray(stroke3)
        ids = (self.nearest_center(s[:, 0], s[:, 1]) + 1).tolist()
        out = [self.SOS]
        for tok, pen in zip(ids, s[:, 2]):
            out.append(tok)
            if pen == 1:
                out.append(self.SEP)
        out.append(self.EOS)
        return _pad_to(out, self.max_seq_len, seq_len, self.PAD, self.SEP, self.EOS)

    def decode(self, seqs):
        if len(seqs) > 0 and isinstance(seqs[0], (list, tuple, np.ndarray)):
            return [self.decode_single(np.squeeze(s)) for s in seqs]
        return self.decode_single(seqs)

    def decode_single(self, seq):
        ids, pens = [], []
        for t in seq:
            t = int(t)
            if t not in (self.SOS, self.EOS, self.SEP, self.PAD):
                ids.append(t - 1)
                pens.append(0)
            elif t == self.SEP and pens:
                pens[-1] = 1
            elif t == self.EOS:
                break
        if not ids:
            return np.zeros((1, 3), dtype=np.float32)
        return np.c_[self.centers[np.array(ids)], np.array(pens)]
