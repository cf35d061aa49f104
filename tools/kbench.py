#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the cfg-2 shapes (B=128, L=200, d=128, H=8, dff=512, V=1004).
Times each C-ABI op with HIP events on the current stream; prints one line per op."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    return ts[len(ts) // 2], ts[0]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = "cuda"
    B, L, d, H, F, V = 128, 200, 128, 8, 512, 1004
    Me, Md = B * L, B * (L - 1)
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    rows = []

    def add(name, flops, nbytes, fn):
        if only and only not in name:
            return
        med, mn = timeit(fn)
        rows.append((name, med, mn, flops / med / 1e6 if flops else 0.0, nbytes / med / 1e3 if nbytes else 0.0))
        print("%-34s med %8.1f us  min %8.1f us  %7.1f TF  %7.0f GB/s" % rows[-1], flush=True)

    # ---- forward GEMMs
    for name, M, N, K, act in (("fwd qkv  25600x384x128", Me, 3 * d, d, 0), ("fwd o    25600x128x128", Me, d, d, 0),
                               ("fwd ffn1 25600x512x128", Me, F, d, 1), ("fwd ffn2 25600x128x512", Me, d, F, 0),
                               ("fwd out  25472x1004x128", Md, V, d, 0), ("fwd kv2  25600x256x128", Me, 2 * d, d, 0)):
        x, w, b = r(M, K), r(K, N), r(N)
        out = torch.empty(M, N, device=dev)
        add("gemm " + name, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N), lambda: ops.gemm(x, w, bias=b, act=act, out=out))
    for n in (2048, 4096):
        x, w = r(n, n), r(n, n)
        out = torch.empty(n, n, device=dev)
        add("gemm square %d^3" % n, 2.0 * n ** 3, 12.0 * n * n, lambda: ops.gemm(x, w, out=out))
    # ---- dgrad
    for name, M, N, K in (("dgrad o    25600x128x128", Me, d, d), ("dgrad qkv  25600x128x384", Me, d, 3 * d),
                          ("dgrad ffn2 25600x512x128", Me, F, d), ("dgrad ffn1 25600x128x512", Me, d, F),
                          ("dgrad out  25472x128x1004", Md, d, V)):
        dy, w = r(M, K), r(N, K)
        out = torch.empty(M, N, device=dev)
        add("gemm " + name, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N),
            lambda: ops.gemm(dy, w, a_kcontig=True, b_kcontig=True, out=out))
    # ---- wgrad (+ bias grad, split-K + reduce)
    lib = _lib.load()
    for name, rows_, inf, outf in (("wgrad o    128x128  k25600", Me, d, d), ("wgrad qkv  128x384  k25600", Me, d, 3 * d),
                                   ("wgrad ffn1 128x512  k25600", Me, d, F), ("wgrad ffn2 512x128  k25600", Me, F, d),
                                   ("wgrad out  128x1004 k25472", Md, d, V)):
        x, dy = r(rows_, inf), r(rows_, outf)
        out = torch.empty(inf, outf, device=dev)
        bg = torch.empty(outf, device=dev)
        sp = lib.skf_gemm_default_splits(inf, outf, rows_)
        add("gemm %s s%d" % (name, sp), 2.0 * rows_ * inf * outf, 4.0 * rows_ * (inf + outf),
            lambda: ops.gemm(x, dy, a_kcontig=False, b_kcontig=False, splits=sp, bias_grad=bg, out=out))
    # ---- attention
    qkv = r(B, L, 3 * d)
    km = (torch.arange(L, device=dev)[None, :] >= torch.randint(8, L, (B, 1), device=dev)).to(torch.uint8)
    for name, causal, mask in (("enc self ", False, km), ("dec self ", True, km), ("cross    ", False, None)):
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        fl = 4.0 * B * H * L * L * (d // H)
        add("attn fwd " + name, fl, 16.0 * B * L * d, lambda: ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal))
        o, st = ops.attention_fwd(q, k, v, H, key_mask=mask, causal=causal)
        do = r(B, L, d)
        add("attn bwd " + name, 2 * fl, 32.0 * B * L * d,
            lambda: ops.attention_bwd(q, k, v, o, do, st, H, key_mask=mask, causal=causal))
    # ---- row kernels
    x, y, g, b_ = r(Me, d), r(Me, d), r(d), r(d)
    add("ln fwd", 0, 16.0 * Me * d, lambda: ops.layernorm_residual_fwd(x, y, g, b_))
    out, z, st = ops.layernorm_residual_fwd(x, y, g, b_)
    add("ln bwd", 0, 12.0 * Me * d, lambda: ops.layernorm_residual_bwd(out, z, st, g))
    lg = r(Md, V)
    tgt = torch.randint(0, V, (B, L), device=dev)
    add("softmax_ce 25472x1004", 0, 8.0 * Md * V, lambda: ops.softmax_ce(lg, tgt, tgt_cols=L - 1, tgt_off=1, mask_pad=True, scale=1e-4))


    if not only or "embed" in only:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
        from sketchformer_amd import synthetic
        xs, _ = synthetic.token_batch(B, L, V, 345, seed=0)
        tok = torch.from_numpy(xs).cuda()
        dxe = r(B * L, d)
        dtab = torch.zeros(V, d, device=dev)
        add("embed_bwd 25600x128", 0, 4.0 * Me * d, lambda: _lib.call("skf_embed_bwd", tok.data_ptr(), L, B, L, dxe.data_ptr(), V, d, dtab.data_ptr(), 0.0, 0, None, torch.cuda.current_stream().cuda_stream))


if __name__ == "__main__":
    main()
