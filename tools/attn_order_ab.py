#!/usr/bin/env python
"""Attention forward / backward at the three call shapes of the cfg-2 step on bench.py's batch: workgroups numbered (sample, head) against
workgroups dealt over the shader engines (skf_deal_rank) from an identity list and from the samples sorted by length (skf_sample_order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, synthetic  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    B, L, d, H = 128, 200, 128, 8
    dev = "cuda"
    xs, _ = synthetic.token_batch(B, L, 1004, 345, seed=0, full=False)
    tok = torch.from_numpy(xs).to(dev)
    enc_mask = (tok == 0).to(torch.uint8).contiguous()
    tar_in = tok[:, :-1]
    dec_mask = (tar_in == 0).to(torch.uint8).contiguous()
    live = ops.target_live_len(tok, L - 1)
    order = ops.sample_order(enc_mask, dec_mask)
    ident = torch.arange(B, device=dev, dtype=torch.int32)
    rows = torch.arange(L - 1, device=dev)[None, :, None]
    qkv = torch.randn(B, L, 3 * d, device=dev)
    cases = (("enc self", L, L, False, enc_mask, None), ("dec self", L - 1, L - 1, True, dec_mask, live), ("cross", L - 1, L, False, None, live))
    modes = (("(sample, head)", None), ("deal, unsorted", ident), ("deal, sorted", order))
    tot = {m: [0.0, 0.0] for m, _ in modes}
    for name, Lq, Lk, causal, mask, ql in cases:
        q, k, v = qkv[:, :Lq, :d], qkv[:, :Lk, d:2 * d], qkv[:, :Lk, 2 * d:]
        do = torch.randn(B, Lq, d, device=dev)
        if ql is not None:
            do = do * (rows < ql[:, None, None]).to(do.dtype)
        o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
        ref = ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal, q_live_len=ql)
        for m, od in modes:
            o2, st2 = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal, sample_order=od)
            g2 = ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal, q_live_len=ql, sample_order=od)
            assert torch.equal(o2, o) and all(torch.equal(a, b) for a, b in zip(g2, ref)), "results depend on the numbering"
            f, _ = timeit(lambda: ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal, sample_order=od))
            bw, _ = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal, q_live_len=ql, sample_order=od))
            tot[m][0] += 4 * f
            tot[m][1] += 4 * bw
            print("%-9s %-15s fwd %6.1f us | bwd %6.1f us" % (name, m, f, bw), flush=True)
    for m, _ in modes:
        print("per step (4 layers x 3 calls) %-15s fwd %.0f us, bwd %.0f us" % (m, tot[m][0], tot[m][1]))


if __name__ == "__main__":
    main()
