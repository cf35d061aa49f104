#!/usr/bin/env python
"""Where does the device leave the bf16-storage restatement on an 83 %-padded cfg-5 batch (encoder/layer7/mha/wq: 9.4e-2)?  The restatement's
own inputs of every ENCODER attention backward (q, k, v, O value + residual, row statistics, dO - all bf16-representable) are handed to the
device kernel (skf_attention_bf16_bwd) and the three outputs compared with the restatement's: a kernel-level difference shows here,
a difference that only builds up along the network does not."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from oracle import bf16_storage  # noqa: E402
from sketchformer_amd import _lib, synthetic  # noqa: E402
import test_gpu_bf16_model as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng, ocfg = T._build(T.CFG5, B)
x, y = synthetic.token_batch(B, ocfg.seq_len, ocfg.vocab_size, ocfg.n_classes, seed=3)
x[1, ocfg.seq_len // 4:] = 0
P = {k: v.astype(np.float64) for k, v in eng.state_dict_numpy().items()}
calls = []
orig = bf16_storage.attn_bwd16


def rec(dout, cache):
    out = orig(dout, cache)
    calls.append((dout, cache, out))
    return out


bf16_storage.attn_bwd16 = rec
bf16_storage.loss_and_grads(P, ocfg, x, x, y, None)
N, H = ocfg.num_layers, ocfg.num_heads
enc_calls = calls[2 * N:]                     # decoder layers first (two calls each), then encoder layers N-1 .. 0
lib = _lib.load()
BF = torch.bfloat16
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
dev = lambda a, dt=BF: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(dt).contiguous()  # noqa: E731
km = torch.as_tensor((x == 0).astype(np.uint8)).cuda()
L, d, dh = ocfg.seq_len, ocfg.d_model, ocfg.d_model // ocfg.num_heads
for li, (dout, cache, (dq, dk, dv)) in zip(range(N - 1, -1, -1), enc_calls):
    qh, kh, vh, mask, m, rinv, ohi, olo, H_, o_exact = cache
    mg = bf16_storage._merge
    Q, K, V, O, Olo, dO = dev(mg(qh)), dev(mg(kh)), dev(mg(vh)), dev(ohi if ohi.ndim == 3 else mg(ohi)), dev(mg(olo) if olo.ndim == 4 else olo), dev(dout)
    stats = torch.as_tensor(np.ascontiguousarray(np.concatenate([m, rinv], -1), dtype=np.float32)).cuda()        # (B, H, L, 2)
    ws = torch.empty(B * H * L, dtype=torch.float32, device="cuda")
    dQ, dK, dV = (torch.empty(B, L, d, dtype=BF, device="cuda") for _ in range(3))
    _lib.call("skf_attention_bf16_bwd", p(Q), d, p(K), d, p(V), d, p(O), d, p(Olo), p(dO), d, p(stats), p(km), L, 0, B, H, L, L, dh,
              p(dQ), d, p(dK), d, p(dV), d, p(ws), ws.numel() * 4, s())
    torch.cuda.synchronize()
    line = "encoder layer %d:" % li
    for name, got, want in (("dq", dQ, dq), ("dk", dK, dk), ("dv", dV, dv)):
        g = got.float().cpu().numpy().astype(np.float64)
        err = np.abs(g - want)
        line += "  %s rel %.2e (max|ref| %.2e)" % (name, err.max() / max(np.abs(want).max(), 1e-30), np.abs(want).max())
    # how much of dq / dk lives in the padded query rows?
    valid = (x != 0)
    line += "  | dq rows: max|ref| valid %.2e padded %.2e" % (np.abs(dq[valid]).max(), np.abs(dq[~valid]).max() if (~valid).any() else 0.0)
    print(line, flush=True)
