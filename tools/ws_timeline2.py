"""s_memtime stamps of one gemm_wsx launch shape (build with SKF_EXTRA_HIPCC_FLAGS=-DSKF_WS_STAMPS=1).
usage: python tools/ws_timeline2.py N K mode   mode: fwd | fwd_relu_bits | dgrad | dgrad_bits | dgrad_acc"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(8 * 32, dtype=torch.int64, device="cuda")
os.environ["SKF_GEMM_DBG"] = str(dbg.data_ptr())
from sketchformer_amd import ops
N, K, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
M = 25600
x = torch.randn(M, K, device="cuda")
out = torch.empty(M, N, device="cuda")
bits = None
if mode.startswith("fwd"):
    w, b = torch.randn(K, N, device="cuda"), torch.randn(N, device="cuda")
    if mode == "fwd_relu_bits":
        bits = ops.relu_bits(M, N, K, "cuda", precision=6)
    run = lambda: ops.gemm(x, w, bias=b, out=out, act=1 if bits is not None else 0, relu_bits_out=bits, precision=6)
else:
    wt = torch.randn(N, K, device="cuda")            # dgrad form: B stored [N][K]
    if mode == "dgrad_bits":
        # the bits of a relu forward launch of the same (M, N, K): ffn dense1 leaves them, the dense2 input gradient reads them
        bits = ops.relu_bits(M, N, K, "cuda", precision=6)
        ops.gemm(x, torch.randn(K, N, device="cuda"), out=out, act=1, relu_bits_out=bits, precision=6)
    run = lambda: ops.gemm(x, wt, b_kcontig=True, out=out, accumulate=(mode == "dgrad_acc"), relu_bits_in=bits if mode == "dgrad_bits" else None, precision=6)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record(); torch.cuda.synchronize()
print("N=%d K=%d %s: %.2f us per call" % (N, K, mode, e0.elapsed_time(e1) * 5.0))
d = dbg.view(8, 32).cpu().numpy()
t0 = d[:, 0][d[:, 0] > 0].min() if (d[:, 0] > 0).any() else 0
for r in d:
    if r[0] == 0: continue
    clk = ""
    if 0 < r[30] < 10 ** 7:
        clk = " | %.2f us, %.2f GHz" % (r[30] / 100.0, r[31] / (r[30] * 10.0))
        r = r[:30]
    s = [int(v - t0) for v in r if v != 0]
    print("  " + " ".join("%5d" % (b_ - a_) for a_, b_ in zip(s[:-1], s[1:])) + " | total %d" % (s[-1] - s[0]) + clk)
