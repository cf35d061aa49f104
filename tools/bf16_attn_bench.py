#!/usr/bin/env python
"""bf16 attention kernels at the cfg-5 shape (B = 128, H = 8, L = 512, dh = 64): encoder self (padded keys), decoder self (causal),
cross (no mask); forward and the two backward passes, us per launch (in-library launch profiler)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import _lib, synthetic
lib = _lib.load()
BF = torch.bfloat16
B, H, L, dh = 128, 8, 512, 64
d = H * dh
x, _ = synthetic.token_batch(B, L, 1004, 345, seed=5)
km = torch.as_tensor(x == 0).to(torch.uint8).cuda()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
Q = torch.randn(B, L, d, device="cuda").to(BF); K = torch.randn(B, L, d, device="cuda").to(BF); V = torch.randn(B, L, d, device="cuda").to(BF)
dO = torch.randn(B, L, d, device="cuda").to(BF)
O = torch.empty_like(Q); Olo = torch.empty_like(Q); stats = torch.empty(B, H, L, 2, device="cuda")
ws = torch.empty(B * H * L, device="cuda"); dQ = torch.empty_like(Q); dK = torch.empty_like(Q); dV = torch.empty_like(Q)
from sketchformer_amd import ops
order = ops.sample_order(km, None)
for name, mask, causal in (("enc self (pad %.2f)" % km.float().mean().item(), km, 0), ("dec self (causal+pad)", km, 1), ("cross / full (no mask)", None, 0), ("causal, no pad", None, 1)):
  for od in ((None, order) if mask is not None else (None,)):
    def fwd():
        _lib.call("skf_attention_bf16_fwd_ordered", p(Q), d, p(K), d, p(V), d, p(mask), L if mask is not None else 0, causal, B, H, L, L, dh, p(O), d, p(Olo), p(stats), p(od), s())
    def bwd():
        _lib.call("skf_attention_bf16_bwd_ordered", p(Q), d, p(K), d, p(V), d, p(O), d, p(Olo), p(dO), d, p(stats), p(mask), L if mask is not None else 0, causal,
                  B, H, L, L, dh, p(dQ), d, p(dK), d, p(dV), d, p(ws), ws.numel() * 4, None, p(od), s())
    for _ in range(3): fwd(); bwd()
    torch.cuda.synchronize(); lib.skf_profiler_enable(1)
    for _ in range(20): fwd(); bwd()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16); lib.skf_profiler_report(buf, len(buf)); lib.skf_profiler_enable(0)
    print("%-24s%-8s" % (name, "sorted" if od is not None else ""), "  ".join("%s %.1f us" % (r["tag"].replace("attn_bf16_", "").replace("<dh64>", ""), r["ms"] / r["count"] * 1e3) for r in json.loads(buf.value.decode())))
