#!/usr/bin/env python
"""Library fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) vs. libskf on the step's shapes (comparison point only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops  # noqa: E402


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return ts[len(ts) // 2]


torch.backends.cuda.matmul.allow_tf32 = False
for name, M, N, K in (("fwd o", 25600, 128, 128), ("fwd qkv", 25600, 384, 128), ("fwd ffn1", 25600, 512, 128),
                      ("fwd ffn2", 25600, 128, 512), ("fwd out", 25472, 1004, 128), ("dgrad out", 25472, 128, 1004)):
    x, w = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    a = t(lambda: torch.mm(x, w, out=out))
    b = t(lambda: ops.gemm(x, w, out=out))
    print("%-10s %6dx%4dx%4d  library %6.1f us (%5.1f TF)   libskf %6.1f us (%5.1f TF)" % (name, M, N, K, a, 2e-6 * M * N * K / a, b, 2e-6 * M * N * K / b))
for name, Kin, Nout, rows in (("wgrad o", 128, 128, 25600), ("wgrad ffn1", 128, 512, 25600), ("wgrad out", 128, 1004, 25472)):
    x, dy = torch.randn(rows, Kin, device="cuda"), torch.randn(rows, Nout, device="cuda")
    out = torch.empty(Kin, Nout, device="cuda")
    a = t(lambda: torch.mm(x.t(), dy, out=out))
    print("%-10s %6dx%4dx%5d  library %6.1f us (%5.1f TF)" % (name, Kin, Nout, rows, a, 2e-6 * Kin * Nout * rows / a))
