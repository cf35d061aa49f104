"""Synthetic QuickDraw-shaped batches.

Shapes and conventions mirror what the reference's ``stroke3-distributed``
loader yields (dataloaders/distributed_stroke3.py:90-153): token mode
``x (B,L) int64`` = ``[SOS] body [EOS] PAD...`` truncated at ``max_seq_len``
(:117-118), continuous mode ``x (B,L,5) float32`` stroke-5 rows with pad rows
``[0,0,0,0,1]`` and the last row's pad bit forced to 1 (:146-151); labels
``y (B,1) int64``.
"""
import numpy as np


def _lengths(rng, batch, seq_len, scale=1.0):
    # sequence lengths ~ N(80, 35^2) at seq_len 200; `scale` stretches the distribution with the sequence length
    # (seq_len 512: scale 2.56 keeps the 58 % padding of the seq_len-200 batches instead of 83 %)
    n = np.rint(rng.normal(80.0 * scale, 35.0 * scale, size=batch)).astype(np.int64)
    return np.clip(n, 8, seq_len)


def token_batch(batch, seq_len=200, vocab_size=1004, n_classes=345, seed=0, full=False, length_scale=1.0):
    """Token-mode batch.  PAD=0, SEP=V-3, SOS=V-2, EOS=V-1 (utils/tokenizer.py:30-33)."""
    rng = np.random.RandomState(seed)
    sep, sos, eos = vocab_size - 3, vocab_size - 2, vocab_size - 1
    x = np.zeros((batch, seq_len), dtype=np.int64)
    lens = np.full(batch, seq_len) if full else _lengths(rng, batch, seq_len, length_scale)
    for b in range(batch):
        n = int(lens[b])
        body = rng.randint(1, vocab_size - 3, size=n)
        body[rng.rand(n) < 0.1] = sep
        row = np.concatenate([[sos], body[: max(n - 3, 0)], [sep, eos]])[:n]
        if n == seq_len:                       # truncated rows lose their EOS
            row = np.concatenate([[sos], body])[:seq_len]
        x[b, : len(row)] = row
    y = rng.randint(0, n_classes, size=(batch, 1)).astype(np.int64)
    return x, y


def continuous_batch(batch, seq_len=200, n_classes=345, seed=0, full=False):
    """Continuous stroke-5 batch (use_continuous_data=True)."""
    rng = np.random.RandomState(seed)
    x = np.zeros((batch, seq_len, 5), dtype=np.float32)
    lens = np.full(batch, seq_len) if full else _lengths(rng, batch, seq_len)
    for b in range(batch):
        n = int(lens[b])
        p = (rng.rand(n) < 0.1).astype(np.float32)
        p[n - 1] = 1.0
        x[b, :n, 0:2] = rng.normal(0.0, 0.05, size=(n, 2))
        x[b, :n, 2] = 1.0 - p
        x[b, :n, 3] = p
        x[b, n:, 4] = 1.0
        x[b, seq_len - 1, 4] = 1.0
    y = rng.randint(0, n_classes, size=(batch, 1)).astype(np.int64)
    return x, y
