"""Host enqueue time per train step against the GPU time per step (is the step host-bound anywhere?).
usage: python tools/host_time.py [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sketchformer_amd import engine, synthetic
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = bench.WORKLOADS["cfg2"]
B = 128
kw = dict(batch=B, seq_len=w["L"], d_model=w["d"], num_heads=8, dff=w["dff"], num_layers=w["N"], vocab_size=w["V"], n_classes=bench.CN,
          lowerdim=bench.U, dropout_rate=0.1, seed=1234, continuous=False, act_dtype=w["act"])
eng = engine.TrainEngine(engine.make_config(use_graph=False, **kw), init_seed=0)
xs, ys = bench.make_batch(synthetic, w, B, 0, False)
x, y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
for _ in range(20):
    eng.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# true host cost: a few steps from an empty queue (the launch queue holds ~13 steps of work before launches block)
for n in (4, 8):
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        eng.train_step(x, y)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    print("%d steps from an empty queue: host %.3f ms/step, GPU-complete %.3f ms/step" % (n, 1e3 * (b - a) / n, 1e3 * (c - a) / n))
print("host enqueue %.3f ms/step, GPU-complete %.3f ms/step (host finished %.1f ms before the GPU)" % (1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps, 1e3 * (t2 - t1)))
