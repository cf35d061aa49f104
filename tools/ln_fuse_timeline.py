"""Per-phase s_memtime stamps of the Dense + LayerNorm launch (build with SKF_EXTRA_HIPCC_FLAGS=-DSKF_WS_STAMPS=1) and the
stand-alone time of the fused launch against the Dense + LayerNorm pair.  usage: python tools/ln_fuse_timeline.py [fused|plain]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(8 * 32, dtype=torch.int64, device="cuda")
os.environ["SKF_GEMM_DBG"] = str(dbg.data_ptr())
from sketchformer_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
M, N, K = 25600, 128, 128
a, x = torch.randn(M, K, device="cuda"), torch.randn(M, N, device="cuda")
w, b = torch.randn(K, N, device="cuda") / 11, torch.randn(N, device="cuda")
g, be = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
st = ops.new_step_state("cuda", iterations=3)
ops.step_prologue(st, seed=5)
y = torch.empty(M, N, device="cuda")


def run():
    if mode == "fused":
        ops.gemm_ln_residual(a, w, b, x, g, be, rate=0.1, site=4, state=st, precision=6)
    else:
        ops.gemm(a, w, bias=b, out=y, precision=6)
        if mode == "pair":
            ops.layernorm_residual_fwd(x, y, g, be, rate=0.1, site=4, state=st)


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record(); torch.cuda.synchronize()
print("%s: %.2f us per call (back to back, includes host allocation of the outputs)" % (mode, e0.elapsed_time(e1) * 5.0))
d = dbg.view(8, 32).cpu().numpy()
if (d[:, 0] > 0).any():
    t0 = d[:, 0][d[:, 0] > 0].min()
    for r in d:
        if r[0] == 0: continue
        clk = ""
        if 0 < r[30] < 10 ** 7:
            clk = " | %.2f us, %.2f GHz" % (r[30] / 100.0, r[31] / (r[30] * 10.0))
            r = r[:30]
        s = [int(v - t0) for v in r if v != 0]
        print("start %6d | " % s[0] + " ".join("%6d" % (b_ - a_) for a_, b_ in zip(s[:-1], s[1:])) + " | total %d" % (s[-1] - s[0]) + clk)
