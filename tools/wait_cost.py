#!/usr/bin/env python
"""What cross-stream synchronisation costs a stream whose kernels run back to back: N kernels of ~25 us on stream A (the host runs
ahead) and a short kernel on stream B per iteration, with
  wait   : A waits (hipStreamWaitEvent) for an event B recorded - complete long before A gets there
  record : A records an event that B waits for (A itself never waits)
  both   : both of the above per iteration
The difference to `none` is what the packet costs stream A."""
import time
import torch

a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 24, device="cuda")        # 64 MB: add_ takes ~25 us
y = torch.zeros(1 << 10, device="cuda")
N = 400


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(b):
            y.add_(1.0)
            if mode in ("wait", "both"):
                ev = torch.cuda.Event(); ev.record(b)
        with torch.cuda.stream(a):
            if mode in ("wait", "both"):
                a.wait_event(ev)
            x.add_(1.0)
            if mode in ("record", "both"):
                e2 = torch.cuda.Event(); e2.record(a)
                b.wait_event(e2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


run("none")
for _ in range(2):
    for mode in ("none", "wait", "record", "both"):
        print("%-7s %.2f us per iteration" % (mode, run(mode)), flush=True)
