#!/usr/bin/env python
"""Greedy reconstruction (predict) at the cfg-2 size: B=128 sketches, 200 decode steps, KV-cached device path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import engine, synthetic  # noqa: E402


def main():
    B, L, V = 128, 200, 1004
    eng = engine.TrainEngine(engine.make_config(batch=B, seq_len=L, d_model=128, num_heads=8, dff=512, num_layers=4,
                                                vocab_size=V, n_classes=345, lowerdim=256, dropout_rate=0.1), init_seed=0)
    x, _ = synthetic.token_batch(B, L, V, 345, seed=0)
    for _ in range(2):
        eng.encode(x)
        out = eng.greedy_decode(None, sos=V - 2, eos=V - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.encode(x)
    eng.synchronize()
    t1 = time.perf_counter()
    out = eng.greedy_decode(None, sos=V - 2, eos=V - 1)
    t2 = time.perf_counter()
    steps = out.shape[1] - 1
    print("encode %.2f ms; greedy decode %d steps x %d sketches: %.1f ms (%.3f ms/step, %.0f tokens/s)"
          % (1e3 * (t1 - t0), steps, B, 1e3 * (t2 - t1), 1e3 * (t2 - t1) / steps, B * steps / (t2 - t1)))


if __name__ == "__main__":
    main()
