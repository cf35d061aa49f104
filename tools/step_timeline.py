#!/usr/bin/env python
"""One train step of a rocprofv3 --kernel-trace CSV as a timeline: every kernel with its queue, start (us from the step's first kernel),
duration, and per-queue busy totals - to see which stream the backward is waiting for."""
import csv
import sys
import collections


def main(path, which=-2):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(path))]
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
    a, b = ends[which - 1] + 1, ends[which] + 1
    seg = rows[a:b]
    t0 = seg[0][0]
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]  # noqa: E731
    busy = collections.Counter()
    for s, e, n, q in seg:
        busy[q] += (e - s) / 1e3
        print("%8.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)))
    print("wall %.1f us; busy per queue:" % ((seg[-1][1] - t0) / 1e3), dict(busy))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
