#!/usr/bin/env python
"""Row-count fuzz of the row-owner launches: the operator tests of tests/test_gpu_ops.py (forward, backward, backward from the LayerNorm
gradient, LayerNorm backward + projection, chained projection, block forward from the attention output) called with random row counts
(1 ... 5000, every residue of the 16-row tile and the 64-row sub-group), random dropout / list / accumulate choices, against the oracle.
usage (GPU box): python tools/fused_fuzz.py [cases=40] [seed=0]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
from sketchformer_amd import ops  # noqa: E402
fails = 0
for c in range(n):
    rows = rnd.choice([rnd.randint(1, 200), rnd.randint(1, 5000), 16 * rnd.randint(1, 300), 64 * rnd.randint(1, 70) + rnd.randint(-1, 1)])
    rate = rnd.choice([0.0, 0.1, 0.35])
    listed, acc = rnd.random() < 0.5, rnd.random() < 0.5
    if listed:
        rows = 199 * rnd.randint(2, 26)      # the listed cases of the tests are (samples x 199 decoder rows)
    n2 = rnd.choice([0, 128, 384])
    calls = [("fwd", lambda: T.test_ffn_fused_forward(ops, rows, rate, rnd.choice([6, 3]))),
             ("bwd", lambda: T.test_ffn_fused_backward(ops, rows, listed, acc)),
             ("bwd_ln", lambda: T.test_ffn_fused_backward_from_layernorm_gradient(ops, rows, listed, rate)),
             ("ln_dgrad", lambda: T.test_layernorm_bwd_dgrad_one_launch(ops, rows, listed, rate)),
             ("fwd_proj", lambda: T.test_ffn_fused_forward_with_chained_projection(ops, rows, n2 or 128)),
             ("block", lambda: T.test_ffn_block_forward_from_attention_output(ops, rows, n2, rate))]
    for name, fn in calls:
        try:
            fn()
        except AssertionError as e:
            fails += 1
            print("FAIL %s rows=%d rate=%.2f listed=%s acc=%s n2=%d: %s" % (name, rows, rate, listed, acc, n2, str(e)[:200]), flush=True)
    if (c + 1) % 10 == 0:
        print("%d cases, %d failures" % (c + 1, fails), flush=True)
print("fuzz: %d cases x 6 operators, %d failures" % (n, fails))
sys.exit(1 if fails else 0)
