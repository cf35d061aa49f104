"""``stroke3-synthetic``: QuickDraw-shaped synthetic batches behind the reference's loader contract
(no files).  Used by bench.py / smoke tests; shapes per SURVEY.md section 8(d)."""
import numpy as np

from ..core.data import BaseDataLoader
from ..utils import hparams as hp
from .. import synthetic


class _Tok(object):
    """Id conventions of the reference's tokenizers (PAD=0, SEP=n+1, SOS=n+2, EOS=n+3); ids 1..n decode as cells of a
    square grid over [-1,1]^2 so that the evaluation plug-ins have something to draw."""

    def __init__(self, vocab):
        self.PAD, self.SEP, self.SOS, self.EOS, self.VOCAB_SIZE = 0, vocab - 3, vocab - 2, vocab - 1, vocab
        self.side = max(1, int(np.ceil(np.sqrt(max(vocab - 4, 1)))))

    def decode_single(self, tokens):
        pts, pens = [], []
        for t in np.asarray(tokens).reshape(-1):
            t = int(t)
            if 0 < t < self.SEP:
                pts.append([((t - 1) % self.side + 0.5) / self.side * 2 - 1, ((t - 1) // self.side + 0.5) / self.side * 2 - 1])
                pens.append(0.0)
            elif t == self.SEP and pens:
                pens[-1] = 1.0
            elif t == self.EOS:
                break
        if not pts:
            return np.zeros((1, 3))
        xy = np.array(pts)
        xy[1:] -= xy[:-1].copy()
        return np.c_[xy, np.array(pens)]

    def decode(self, seqs):
        return [self.decode_single(s) for s in seqs]


class SyntheticStroke3DataLoader(BaseDataLoader):
    name = "stroke3-synthetic"

    @classmethod
    def default_hparams(cls):
        return hp.HParams(max_seq_len=200, use_continuous_data=False, vocab_size=1004, n_classes=345,
                          n_samples=128 * 64, seed=0)

    def __init__(self, hps, data_directory=None):
        h = hps if isinstance(hps, dict) else dict(hps.values())
        self.tokenizer = _Tok(h["vocab_size"])
        self.n_classes, self.n_samples = h["n_classes"], h["n_samples"]
        self.class_names = np.array(["class%d" % i for i in range(self.n_classes)])
        super().__init__(hps, data_directory)

    def get_data_splits(self):
        return []

    def load_next_megabatch(self, split_name, selected_file):
        pass

    def batch_iterator(self, split_name, batch_size, stop_at_end_of_split):
        seed = self.hps["seed"] + {"train": 0, "valid": 1, "test": 2}.get(split_name, 3) * 100003
        n_batches = max(1, self.n_samples // batch_size)
        i = 0
        while True:
            if self.hps["use_continuous_data"]:
                yield synthetic.continuous_batch(batch_size, self.hps["max_seq_len"], self.n_classes, seed + i)
            else:
                yield synthetic.token_batch(batch_size, self.hps["max_seq_len"], self.hps["vocab_size"],
                                            self.n_classes, seed + i)
            i += 1
            if stop_at_end_of_split and i >= n_batches:
                return

    # the evaluation plug-ins sample / sweep a split (core/data.py:115-146 of the reference)
    def get_n_samples_from(self, split_name, n, shuffled=False, seeded=False):
        x, y = next(self.batch_iterator(split_name, n, True))
        return x, y

    def get_all_data_from(self, split_name):
        n = max(64, min(512, self.n_samples // 8))        # a synthetic "split" is as long as we say
        x, y = next(self.batch_iterator(split_name, n, True))
        return x, y
