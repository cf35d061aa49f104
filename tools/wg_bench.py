"""Stand-alone time of the weight-gradient launches of one layer shape.  usage: python tools/wg_bench.py [Kin Nout]..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib
M = 25600
shapes = [(128, 384), (128, 128), (128, 512), (512, 128)]
for kin, nout in shapes:
    x, dy = torch.randn(M, kin, device="cuda"), torch.randn(M, nout, device="cuda")
    dw, db = torch.empty(kin, nout, device="cuda"), torch.empty(nout, device="cuda")
    splits = _lib.load().skf_gemm_default_splits(kin, nout, M)
    run = lambda: ops.gemm(x, dy, a_kcontig=False, b_kcontig=False, out=dw, splits=splits, bias_grad=db, precision=6)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record(); torch.cuda.synchronize()
    print("wgrad %dx%d over %d rows, %d splits: %.2f us per call (partial tiles + reduce)" % (kin, nout, M, splits, e0.elapsed_time(e1) * 10.0))
