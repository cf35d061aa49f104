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


def top_gaps(path, steps=10, n=25):
    """Largest idle intervals (no kernel running on any stream) inside the last `steps` steps, with their neighbours."""
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]][-(steps + 1):]
    seg = rows[ends[0] + 1: ends[-1] + 1]
    gaps, cur_e, prev = {}, None, None
    for s, e, name in seg:
        if cur_e is not None and s > cur_e:
            short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]  # noqa: E731
            key = (short(prev), short(name))
            g = gaps.setdefault(key, [0, 0.0])
            g[0] += 1
            g[1] += (s - cur_e) / 1e3
        if cur_e is None or e > cur_e:
            cur_e, prev = e, name
    tot = sum(v[1] for v in gaps.values())
    print("idle total %.1f us/step in %d distinct transitions" % (tot / steps, len(gaps)))
    for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:n]:
        print("%7.1f us/step  x%5.1f/step  %-40s -> %s" % (t / steps, c / steps, a, b))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "gaps":
    top_gaps(sys.argv[1], int(sys.argv[2]))
