#!/usr/bin/env python
"""Experiment: attention backward (skf_attention_bwd3.hip) with the second workgroup of every CU delayed at the start of the launch
(SKF_ATTN_ABLATE = 1000 x delay in 10-ns ticks; -DSKF_MEASURE=1 build copied over sketchformer_amd/libskf.so)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, synthetic  # noqa: E402
from kbench import timeit  # noqa: E402

B, L, d, H = 128, 200, 128, 8
qkv = torch.randn(B, L, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
do = torch.randn(B, L, d, device="cuda")
xs, _ = synthetic.token_batch(B, L, 1004, 345, seed=0)
km = torch.from_numpy((xs == 0).astype("uint8")).cuda()
for name, causal, mask in (("full", False, None), ("causal", True, None), ("bench enc", False, km)):
    o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
    for ticks in [int(a) for a in (sys.argv[1:] or ["0", "200", "400", "600", "800", "1200", "1600"])]:
        os.environ["SKF_ATTN_ABLATE"] = str(1000 * ticks)
        med, mn = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal))
        print("%-9s delay %5.1f us: med %7.1f us  min %7.1f us" % (name, ticks / 100.0, med, mn), flush=True)
