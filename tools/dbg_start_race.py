#!/usr/bin/env python
"""Stress: TrainEngine built, state[0] = START written on the caller's stream, three steps - the result must not depend on timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sketchformer_amd import engine, synthetic
KW = dict(seq_len=24, d_model=64, num_heads=4, dff=128, num_layers=2, vocab_size=52, n_classes=7, lowerdim=32, dropout_rate=0.0, use_graph=False, seed=5)
batches = [synthetic.token_batch(8, 24, 52, 7, seed=70 + s) for s in range(3)]
sums = set()
for rep in range(40):
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(rep % 5)]      # allocator / timing churn
    if rep % 3 == 0:
        a = torch.randn(4096, 4096, device="cuda"); b = a @ a                    # keep the default stream busy
    ref = engine.TrainEngine(engine.make_config(batch=8, **KW), init_seed=1)
    ref.state[0] = 3000
    for x, y in batches:
        ref.train_step(x, y)
    torch.cuda.synchronize()
    sums.add((ref.iterations, float(ref.params.double().sum().item())))
    del junk
print(len(sums), sorted(sums))
