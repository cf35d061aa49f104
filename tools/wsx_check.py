"""Error (vs float64) and time of the Dense GEMM in its three arithmetic modes (fp32 MFMA, bf16x6, bf16x3)
at the cfg-2 shapes.  python tools/wsx_check.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
M = 25600
SHAPES = [(128, 128, False), (128, 384, False), (128, 512, False), (512, 128, False), (128, 1004, False),
          (128, 128, True), (384, 128, True), (512, 128, True), (128, 512, True), (256, 128, False)]   # (K, N, b_kcontig)
g = torch.Generator(device="cpu").manual_seed(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, N, bkc in SHAPES:
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) if bkc else torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = a.double() @ (b.double().t() if bkc else b.double()) + bias.double()
    scale = (a.double().abs() @ (b.double().abs().t() if bkc else b.double().abs())).mean().item()
    ad, bd, biasd = a.to(dev), b.to(dev), bias.to(dev)
    out = torch.empty(M, N, device=dev)
    line = "K=%3d N=%4d %s " % (K, N, "B[N][K]" if bkc else "B[K][N]")
    for mode in (0, 6, 3):
        lib.skf_set_gemm_precision(mode)
        fn = lambda: ops.gemm(ad, bd, True, bkc, bias=biasd, out=out)
        fn()
        err = (out.cpu().double() - ref).abs()
        us = timeit(fn)
        line += "| %s max %.2e mean %.2e (rel to sum|a||b|: %.1e) %6.1f us " % (
            {0: "f32   ", 6: "bf16x6", 3: "bf16x3"}[mode], err.max().item(), err.mean().item(), err.max().item() / scale, us)
    print(line, flush=True)
lib.skf_set_gemm_precision(0)
