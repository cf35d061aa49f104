#!/usr/bin/env python
"""Decoder-side backward GEMMs of cfg 2 (M = 128 x 199 rows) with and without the live-row-block lists, QuickDraw-shaped lengths."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import _lib, ops, synthetic
lib = _lib.load()
B, L = 128, 200
Ld = L - 1
x, _ = synthetic.token_batch(B, L, 1004, 345, seed=5)
tar = torch.as_tensor(x).cuda()
ll = ops.target_live_len(tar, Ld)
b16, b32 = ops.row_blocks(ll, Ld, 16), ops.row_blocks(ll, Ld, 32)
print("live rows %.3f, live 16-row tiles %.3f, live 32-row blocks %.3f" % (ll.float().sum().item() / (B * Ld), b16[0].item() / b16[1].item(), b32[0].item() / b32[1].item()))
live = (torch.arange(Ld, device="cuda")[None, :] < ll[:, None]).reshape(-1, 1).float()


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.skf_profiler_enable(1)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    lib.skf_profiler_report(buf, len(buf)); lib.skf_profiler_enable(0)
    return sum(r["ms"] / r["count"] * 1e3 for r in json.loads(buf.value.decode()))


M = B * Ld
for N, K, relu in ((128, 128, 0), (512, 128, 1), (128, 512, 0), (128, 384, 0)):
    dy = torch.randn(M, K, device="cuda") * live
    w = torch.randn(N, K, device="cuda"); h = torch.randn(M, N, device="cuda") if relu else None
    out = torch.empty(M, N, device="cuda")
    d = timeit(lambda: ops.gemm(dy, w, a_kcontig=True, b_kcontig=True, relu_src=h, out=out))
    r = timeit(lambda: ops.gemm(dy, w, a_kcontig=True, b_kcontig=True, relu_src=h, out=out, row_blocks=b16, row_block_rows=16))
    print("dgrad N=%4d K=%4d  dense %6.1f us   live blocks %6.1f us" % (N, K, d, r))
for inf, outf in ((128, 384), (128, 128), (128, 512), (512, 128), (128, 1004)):
    xx = torch.randn(M, inf, device="cuda"); dy = torch.randn(M, outf, device="cuda") * live
    sp = lib.skf_gemm_default_splits(inf, outf, M)
    bg = torch.zeros(outf, device="cuda")
    d = timeit(lambda: ops.gemm(xx, dy, a_kcontig=False, b_kcontig=False, splits=sp, bias_grad=bg))
    r = timeit(lambda: ops.gemm(xx, dy, a_kcontig=False, b_kcontig=False, splits=sp, bias_grad=bg, row_blocks=b32, row_block_rows=32))
    print("wgrad %4d x %4d    dense %6.1f us   live blocks %6.1f us   (partial tiles + reduce)" % (inf, outf, d, r))
