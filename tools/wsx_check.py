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
    """kernel duration from libskf's launch profiler (HIP events around each launch on the stream): a Python
    call costs ~12 us, so timing a loop of calls would measure the host for the short kernels"""
    import ctypes as C, json
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.skf_profiler_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    lib.skf_profiler_report(buf, len(buf))
    lib.skf_profiler_enable(0)
    rows = json.loads(buf.value.decode())
    return sum(r["ms"] for r in rows) / n * 1e3


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
        fn = lambda mode=mode: ops.gemm(ad, bd, True, bkc, bias=biasd, out=out, precision=mode)
        fn()
        err = (out.cpu().double() - ref).abs()
        us = timeit(fn)
        line += "| %s %6.1f us err max %.1e mean %.1e " % ({0: "f32   ", 6: "bf16x6", 3: "bf16x3"}[mode], us, err.max().item(), err.mean().item())
    print(line, flush=True)
