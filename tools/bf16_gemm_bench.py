"""bf16 Dense kernels at the cfg-5 shapes: time (libskf launch profiler) and TFLOP/s.  python tools/bf16_gemm_bench.py"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import _lib
lib = _lib.load()
BF = torch.bfloat16
dev = "cuda"
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, n=20):
    for _ in range(3):
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
    return {r["tag"]: r["ms"] / r["count"] * 1e3 for r in rows}


M = 65536
for name, N, K in (("qkv", 1536, 512), ("o / q", 512, 512), ("ffn1", 2048, 512), ("ffn2", 512, 2048), ("kv", 1024, 512)):
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF); c = torch.empty(M, N, dtype=BF, device=dev)
    bias = torch.randn(N, device=dev)
    t = timeit(lambda: _lib.call("skf_gemm_bf16", M, N, K, p(a), K, p(b), K, p(c), N, p(bias), 0, None, 0, 0, None, 0, s()))
    us = t["gemm_bf16_nt"]
    print("nt   %-6s M=%d N=%4d K=%4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
for name, N, K, relu, acc in (("dgrad ffn2 (relu mask)", 2048, 512, 1, 0), ("dgrad o (accumulate)", 512, 512, 0, 1), ("dgrad qkv", 512, 1536, 0, 0)):
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF); c = torch.zeros(M, N, dtype=BF, device=dev)
    h = torch.randn(M, N, device=dev).to(BF)
    t = timeit(lambda: _lib.call("skf_gemm_bf16", M, N, K, p(a), K, p(b), K, p(c), N, None, 0, p(h) if relu else None, N, acc, None, 0, s()))
    us = t["gemm_bf16_nt"]
    print("nt   %-22s M=%d N=%4d K=%4d  %7.1f us  %6.0f TF" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
for name, P, Q in (("qkv", 512, 1536), ("o / q", 512, 512), ("ffn1", 512, 2048), ("ffn2", 2048, 512)):
    x = torch.randn(M, P, device=dev).to(BF); dy = torch.randn(M, Q, device=dev).to(BF)
    dw = torch.empty(P, Q, device=dev); db = torch.empty(Q, device=dev)
    nb = lib.skf_gemm_bf16_wgrad_workspace_bytes(P, Q, M, lib.skf_gemm_bf16_wgrad_splits(P, Q, M))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    t = timeit(lambda: _lib.call("skf_gemm_bf16_wgrad", P, Q, M, p(x), P, p(dy), Q, p(dw), Q, p(db), p(ws), nb, s()))
    us = t["gemm_bf16_tn(wgrad)"]
    print("tn   %-6s P=%4d Q=%4d R=%d  %7.1f us  %6.0f TF   (+ reduce %.1f us)" % (name, P, Q, M, us, 2.0 * M * P * Q / us / 1e6, t.get("splitk_reduce", 0)))
