import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(8 * 32, dtype=torch.int64, device="cuda")
os.environ["SKF_GEMM_DBG"] = str(dbg.data_ptr())
from sketchformer_amd import ops
M, N, K = 25600, int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 128
x, w, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda"), torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.gemm(x, w, bias=b, out=out)
torch.cuda.synchronize()
d = dbg.view(8, 32).cpu().numpy()
t0 = d[:, 0][d[:, 0] > 0].min()
for r in d:
    if r[0] == 0: continue
    clk = ""
    if 0 < r[30] < 10 ** 7:   # split-operand kernel: [30] = 100 MHz wall ticks, [31] = shader cycles of the workgroup's life
        clk = " | %.2f us, %.2f GHz" % (r[30] / 100.0, r[31] / (r[30] * 10.0))
        r = r[:30]
    st = [int(v - t0) for v in r if v != 0]
    print("start %6d | " % st[0] + " ".join("%6d" % (b_ - a_) for a_, b_ in zip(st[:-1], st[1:])) + " | total %d end %d" % (st[-1] - st[0], st[-1]) + clk)
