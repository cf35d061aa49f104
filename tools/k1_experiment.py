"""Determinism of the accumulate-only epilogue (kind 1) of gemm_wsx: N x K shapes with >= 3 column groups, `reps` runs each;
prints the number of runs that differ from run 0 and which cells (tile row, lane group) differ."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from sketchformer_amd import ops
torch.manual_seed(0)
M = 25600
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for (N, K) in [(256, 256), (768, 256), (1024, 256), (512, 256), (128, 256), (512, 128), (256, 512)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") / K ** 0.5
    c0 = torch.randn(M, N, device="cuda")
    ref = (x.double() @ wt.double().t()) + c0.double()
    first, bad_runs, rows, cols, worst = None, 0, set(), set(), 0.0
    for r in range(reps):
        out = c0.clone()
        ops.gemm(x, wt, b_kcontig=True, out=out, accumulate=True, precision=6)
        if first is None:
            first = out.clone()
        err = (out.double() - ref).abs()
        worst = max(worst, err.max().item() / ref.abs().max().item())
        bad = (err > 1e-3 * ref.abs().max()).nonzero()
        if len(bad):
            bad_runs += 1
            rows |= set((bad[:, 0] % 16).tolist()); cols |= set(((bad[:, 1] % 64) // 16).tolist())
    print("N=%d K=%d: %d / %d runs wrong, worst rel err %.2e, rows-in-tile %s, wave-in-group %s" % (N, K, bad_runs, reps, worst, sorted(rows), sorted(cols)), flush=True)
