"""One Dense GEMM shape in one arithmetic mode, a few launches (for rocprofv3 --pmc / --kernel-trace runs).
python tools/wsx_one.py N K mode [b_kcontig]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchformer_amd import ops, _lib
N, K, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bkc = len(sys.argv) > 4 and sys.argv[4] == "1"
M = 25600
a = torch.randn(M, K, device="cuda")
b = torch.randn(N, K, device="cuda") if bkc else torch.randn(K, N, device="cuda")
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    ops.gemm(a, b, True, bkc, bias=bias, out=out, precision=mode)
torch.cuda.synchronize()
