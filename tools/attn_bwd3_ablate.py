#!/usr/bin/env python
"""Attention backward (dh = 16, cfg-2 shape) under the SKF_ATTN_ABLATE bits of a -DSKF_MEASURE=1 build (tools/build_measure.sh, copied
over sketchformer_amd/libskf.so on the GPU box): 1 = role A without its products, 2 = role B without its products, 4 = role A
workgroups only, 8 = role B workgroups only.  Full-length rows, random dO; results of ablated runs are wrong by construction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    B, L, d, H = 128, 200, 128, 8
    dev = "cuda"
    qkv = torch.randn(B, L, 3 * d, device=dev)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    do = torch.randn(B, L, d, device=dev)
    lens = torch.randint(8, L, (B, 1), device=dev)
    km = (torch.arange(L, device=dev)[None, :] >= lens).to(torch.uint8)
    for name, causal, mask in (("full", False, None), ("causal", True, None), ("padded", False, km)):
        o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
        for ab in [int(a) for a in (sys.argv[1:] or ["0", "4", "8", "5", "10", "3"])]:
            os.environ["SKF_ATTN_ABLATE"] = str(ab)
            med, mn = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal))
            print("%-7s ablate %2d: med %7.1f us  min %7.1f us" % (name, ab, med, mn), flush=True)
        os.environ["SKF_ATTN_ABLATE"] = "0"
        for env, val in (("SKF_ATTN_BWD3", "0"),):
            os.environ[env] = val
            med, mn = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal))
            print("%-7s one-pass kernel: med %7.1f us  min %7.1f us" % (name, med, mn), flush=True)
            del os.environ[env]


if __name__ == "__main__":
    main()
