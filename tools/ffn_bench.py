#!/usr/bin/env python
"""The fused feed-forward launches (skf_ffn_fused_fwd_f32 / _bwd_f32) against the launches they replace, cfg-2 shape
(25600 x 128 -> 512 -> 128, dropout 0.1), HIP events on the current stream."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    dev = "cuda"
    M, d, F = int(sys.argv[1]) if len(sys.argv) > 1 else 25600, 128, 512
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    x, w1, b1, w2, b2, g, be = r(M, d), r(d, F) / 11, r(F), r(F, d) / 22, r(d), r(d), r(d)
    st = ops.new_step_state(dev, iterations=3)
    ops.step_prologue(st, seed=1)
    rate = 0.1
    img, = ops.ffn_weight_images([(w1, w2)], transpose=False)
    imgt, = ops.ffn_weight_images([(w1, w2)], transpose=True)
    lib = _lib.load()
    s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    h, z, out, stats = torch.empty(M, F, device=dev), torch.empty(M, d, device=dev), torch.empty(M, d, device=dev), torch.empty(M, 2, device=dev)
    bits = torch.zeros(lib.skf_ffn_relu_bits_bytes(M, d, F, 6) // 8, dtype=torch.int64, device=dev)

    def fused_fwd():
        _lib.call("skf_ffn_fused_fwd_f32", M, d, F, p(x), p(img), p(b1), p(b2), p(h), p(bits), p(g), p(be), p(z), p(out), p(stats),
                  rate, 7, p(st), 6, s())

    hb = ops.relu_bits(M, F, d, dev)
    y = torch.empty(M, d, device=dev)

    wp, bp = r(d, 384) / 11, r(384)
    pimg = ops.dense_weight_image(wp, transpose=False)
    po = torch.empty(M, 384, device=dev)

    def fused_fwd_proj():
        _lib.call("skf_ffn_fused_fwd_proj_f32", M, d, F, p(x), p(img), p(b1), p(b2), p(h), p(bits), p(g), p(be), p(z), p(out), p(stats),
                  rate, 7, p(st), p(pimg), p(bp), 384, p(po), 6, s())

    def qkv_gemm():
        ops.gemm(out, wp, bias=bp, out=po)

    def unfused_fwd():
        ops.gemm(x, w1, bias=b1, act=1, out=h, relu_bits_out=hb)
        ops.gemm(h, w2, bias=b2, out=y)
        _lib.call("skf_layernorm_residual_fwd", p(x), p(y), p(g), p(be), p(out), p(stats), M, d, rate, 7, p(st), s())

    dy, dh, dx = r(M, d), torch.empty(M, F, device=dev), r(M, d)

    def fused_bwd():
        _lib.call("skf_ffn_fused_bwd_f32", M, d, F, p(dy), p(imgt), p(bits), p(dh), p(dx), 1, None, 0, 6, s())

    def unfused_bwd():
        ops.gemm(dy, w2, a_kcontig=True, b_kcontig=True, out=dh, relu_bits_in=hb)
        ops.gemm(dh, w1, a_kcontig=True, b_kcontig=True, out=dx, accumulate=True)

    def images():
        ops.ffn_weight_images([(w1, w2)] * 8, transpose=False)

    fl = 4.0 * M * d * F
    for name, fn in (("fused fwd", fused_fwd), ("fused fwd + qkv", fused_fwd_proj), ("qkv gemm alone", qkv_gemm), ("3 launches fwd", unfused_fwd), ("fused bwd", fused_bwd), ("2 launches bwd", unfused_bwd),
                     ("8 image pairs", images)):
        med, mn = timeit(fn, iters=50)
        print("%-16s med %7.1f us  min %7.1f us  %6.1f TF fp32-equivalent" % (name, med, mn, fl / med / 1e6), flush=True)


if __name__ == "__main__":
    main()
