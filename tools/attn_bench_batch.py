#!/usr/bin/env python
"""Attention forward / backward at the three call shapes of the cfg-2 train step on bench.py's own batch (seed-0 synthetic QuickDraw-shaped
lengths, 58 % padding): encoder self (key padding mask), decoder self (look-ahead + target padding mask, live query lengths), cross (blind:
no key mask, live query lengths); and the same with full-length rows.  Measurement builds: SKF_ATTN_BWD3=0 column = the one-pass kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, synthetic  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    B, L, d, H = 128, 200, 128, 8
    dev = "cuda"
    for full in (False, True):
        xs, _ = synthetic.token_batch(B, L, 1004, 345, seed=0, full=full)
        tok = torch.from_numpy(xs).to(dev)
        enc_mask = (tok == 0).to(torch.uint8).contiguous()                 # (B, L) encoder key padding
        tar_in = tok[:, :-1]
        dec_mask = (tar_in == 0).to(torch.uint8).contiguous()              # (B, L-1) target padding
        live = ops.target_live_len(tok, L - 1)
        rows = torch.arange(L - 1, device=dev)[None, :, None]
        qkv = torch.randn(B, L, 3 * d, device=dev)
        cases = (("enc self", L, L, False, enc_mask, None), ("dec self", L - 1, L - 1, True, dec_mask, live), ("cross", L - 1, L, False, None, live))
        tot = {"new": 0.0, "old": 0.0, "fwd": 0.0}
        for name, Lq, Lk, causal, mask, ql in cases:
            q, k, v = qkv[:, :Lq, :d], qkv[:, :Lk, d:2 * d], qkv[:, :Lk, 2 * d:]
            do = torch.randn(B, Lq, d, device=dev)
            if ql is not None:
                do = do * (rows < ql[:, None, None]).to(do.dtype)          # decoder rows behind the last trained position: dO == 0 exactly
            o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
            fm, _ = timeit(lambda: ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal))
            res = {}
            for tag, env in (("new", None), ("old", "0")):
                if env is None:
                    os.environ.pop("SKF_ATTN_BWD3", None)
                else:
                    os.environ["SKF_ATTN_BWD3"] = env
                res[tag], _ = timeit(lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal, q_live_len=ql))
            os.environ.pop("SKF_ATTN_BWD3", None)
            for t_ in ("new", "old"):
                tot[t_] += res[t_] * 4
            tot["fwd"] += fm * 4
            print("%-12s %-9s fwd %6.1f us | bwd %6.1f us (SKF_ATTN_BWD3=0: %6.1f us)" % ("full-length" if full else "bench batch", name, fm, res["new"], res["old"]), flush=True)
        print("%-12s per step (4 layers x 3 calls): fwd %.0f us, bwd %.0f us (one-pass %.0f us)" % ("full-length" if full else "bench batch", tot["fwd"], tot["new"], tot["old"]), flush=True)


if __name__ == "__main__":
    main()
