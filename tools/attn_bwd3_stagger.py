#!/usr/bin/env python
"""Attention backward (dh = 16, cfg-2 shape) against the start offset of the second workgroup of a CU (SKF_ATTN_STAGGER, shader cycles;
-DSKF_MEASURE=1 build copied over sketchformer_amd/libskf.so)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

B, L, d, H = 128, 200, 128, 8
qkv = torch.randn(B, L, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
do = torch.randn(B, L, d, device="cuda")
km = (torch.arange(L, device="cuda")[None, :] >= torch.randint(8, L, (B, 1), device="cuda")).to(torch.uint8)
for name, causal, mask in (("full", False, None), ("causal", True, None), ("padded", False, km)):
    o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
    for sg in [int(a) for a in (sys.argv[1:] or ["0", "4000", "8000", "12000", "16000", "24000", "32000"])]:
        os.environ["SKF_ATTN_STAGGER"] = str(sg)
        med, mn = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal))
        print("%-7s stagger %6d: med %7.1f us  min %7.1f us" % (name, sg, med, mn), flush=True)
