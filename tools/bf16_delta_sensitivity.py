#!/usr/bin/env python
"""CPU-only experiment behind the cfg-5 gradient bars (tests/test_gpu_bf16_model.py): how sensitive are the gradients of the bf16 path to the
precision of the two terms of dS = P o (dP - delta)?  The restatement of oracle/bf16_storage.py is evaluated at the cfg-5 dimensions (B = 2, the
test's batch) with delta = rowsum(dO o O) formed from (a) bf16(O) + its bf16 residual = what the kernels store (the reference run), (b) bf16(O)
alone, (c) the unrounded O, and with fp32-rounding-sized relative noise (2^-24) injected into (d) delta, (e) dP and delta - the size of the
difference between the device's fp32 accumulation and float64.  Printed per variant: the worst per-tensor change of the gradient relative
to max |G| (the test's norm), the number of tensors moved by >= 1.5e-2, the five most sensitive tensors."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle import bf16_storage  # noqa: E402
from sketchformer_amd import synthetic  # noqa: E402

CFG5 = dict(seq_len=512, d_model=512, num_heads=8, dff=2048, num_layers=8, vocab_size=1004, n_classes=345, lowerdim=256)
SMALL = dict(seq_len=40, d_model=128, num_heads=2, dff=256, num_layers=2, vocab_size=52, n_classes=7, lowerdim=32)


def main():
    kw = SMALL if len(sys.argv) > 1 and sys.argv[1] == "small" else CFG5
    B = 2
    cfg = oracle.Config(dropout_rate=0.0, blind_decoder_mask=True, **kw)
    P = oracle.init_params(cfg, seed=1)
    rng = np.random.RandomState(9)
    for n in P:
        if n.endswith(("/bias", "/beta", "b_attn")):
            P[n] = rng.normal(0, 0.1, P[n].shape)
        elif n.endswith("/gamma"):
            P[n] = 1 + rng.normal(0, 0.1, P[n].shape)
    x, y = synthetic.token_batch(B, cfg.seq_len, cfg.vocab_size, cfg.n_classes, seed=3)
    x[1, cfg.seq_len // 4:] = 0

    def run(hook):
        bf16_storage.DELTA_HOOK = hook
        t0 = time.time()
        _, _, G = bf16_storage.loss_and_grads(P, cfg, x, x, y, None)
        bf16_storage.DELTA_HOOK = None
        return G, time.time() - t0

    G0, dt = run(None)
    print("reference restatement (delta from bf16(O) + bf16 residual): %.1f s, %d tensors" % (dt, len(G0)))
    floor = 1e-2 * np.median([np.abs(G0[k]).max() for k in G0])
    nrng = np.random.RandomState(123)

    def noisy(a, eps):
        return a * (1.0 + eps * nrng.standard_normal(a.shape))

    variants = [
        ("delta from bf16(O) only (8 significand bits)", lambda d, dp, do, ohi, olo, o: ((do * ohi).sum(-1, keepdims=True), dp)),
        ("delta from the unrounded O", lambda d, dp, do, ohi, olo, o: ((do * o).sum(-1, keepdims=True), dp)),
        ("delta x (1 + 2^-24 N(0,1))", lambda d, dp, do, ohi, olo, o: (noisy(d, 2.0 ** -24), dp)),
        ("dP and delta x (1 + 2^-24 N(0,1))  [fp32 accumulation-sized]", lambda d, dp, do, ohi, olo, o: (noisy(d, 2.0 ** -24), noisy(dp, 2.0 ** -24))),
        ("dP and delta x (1 + 2^-21 N(0,1))  [~8 fp32 roundings deep]", lambda d, dp, do, ohi, olo, o: (noisy(d, 2.0 ** -21), noisy(dp, 2.0 ** -21))),
    ]
    for name, hook in variants:
        G, dt = run(hook)
        rel = {k: np.abs(G[k] - G0[k]).max() / max(np.abs(G0[k]).max(), floor) for k in G0 if not k.endswith("wk/bias")}
        top = sorted(rel.items(), key=lambda kv: -kv[1])[:5]
        print("%-62s worst %.3e, median %.3e, %3d of %d tensors moved by >= 1.5e-2; most sensitive: %s  (%.0f s)"
              % (name, top[0][1], np.median(list(rel.values())), sum(v >= 1.5e-2 for v in rel.values()), len(rel),
                 ", ".join("%s %.1e" % (k, v) for k, v in top), dt), flush=True)


if __name__ == "__main__":
    main()
