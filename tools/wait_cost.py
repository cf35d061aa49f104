#!/usr/bin/env python
"""What a cross-stream wait costs the WAITING stream: N kernels of ~25 us on stream A (the host runs ahead), a short kernel + event record
on stream B per iteration; with / without A waiting for that event in front of its kernel.  The event is complete long before A gets
there (B's kernels depend on nothing), so the difference is what the wait itself costs A."""
import time
import torch

a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 24, device="cuda")        # 64 MB: add_ takes ~25 us
y = torch.zeros(1 << 10, device="cuda")
N = 400


def run(wait, every=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        ev = None
        with torch.cuda.stream(b):
            y.add_(1.0)
            if i % every == 0:
                ev = torch.cuda.Event(); ev.record(b)
        with torch.cuda.stream(a):
            if wait and ev is not None:
                a.wait_event(ev)
            x.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for _ in range(2):
    print("no wait            %.2f us per iteration" % run(False), flush=True)
    print("wait every kernel  %.2f us per iteration" % run(True), flush=True)
    print("wait every 4th     %.2f us per iteration" % run(True, 4), flush=True)
