#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV of `bench.py --no-profile`: wall time per step vs. the sum of kernel durations
(the difference is inter-kernel gaps minus two-stream overlap), for the last N steps."""
import csv
import sys


def main(path, steps):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    # one 'adam' kernel per step marks the step boundary
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
    ends = ends[-(steps + 1):]
    seg = rows[ends[0] + 1: ends[-1] + 1]
    n = len(ends) - 1
    wall = (seg[-1][1] - seg[0][0]) / 1e3 / n
    busy = sum(e - s for s, e, _ in seg) / 1e3 / n
    # union of busy intervals (two streams overlap)
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print("steps %d  kernels/step %.1f  wall/step %.1f us  sum(kernel durations)/step %.1f us  union busy/step %.1f us  idle/step %.1f us"
          % (n, len(seg) / n, wall, busy, union / 1e3 / n, wall - union / 1e3 / n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
