"""Run-to-run determinism and float64 error of the accumulate / chained input-gradient launches of gemm_wsx (8 shapes x 4 runs)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from sketchformer_amd import ops
torch.manual_seed(0)
M = 25600
for (N, K, acc) in [(256, 768, False), (256, 256, True), (256, 1024, False), (256, 512, True), (128, 256, True), (128, 128, True), (768, 256, True), (1024, 256, True)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") / K ** 0.5
    c0 = torch.randn(M, N, device="cuda")
    ref = (x.double() @ wt.double().t()) + (c0.double() if acc else 0)
    outs = []
    for r in range(4):
        out = c0.clone()
        ops.gemm(x, wt, b_kcontig=True, out=out, accumulate=acc, precision=6)
        outs.append(out.clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
    errs = [((o.double() - ref).abs().max().item() / ref.abs().max().item()) for o in outs]
    print("N=%d K=%d acc=%d: deterministic=%s, rel err per run %s" % (N, K, acc, same, ["%.2e" % e for e in errs]))
