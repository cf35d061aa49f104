#!/usr/bin/env python
"""Largest GPU-idle interval of each of the last steps of a rocprofv3 --kernel-trace (+ optional --hip-trace) run: the kernels
on either side with their queue ids, and - when the HIP API trace is present - when the host issued the launch that ended the gap."""
import csv
import glob
import sys


def main(d, steps=6):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40],
             r.get("Queue_Id", "?"), int(r.get("Correlation_Id", 0))) for r in csv.DictReader(open(kt))]
    rows.sort()
    api = {}
    ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)
    if ht:
        for r in csv.DictReader(open(ht[0])):
            api[int(r["Correlation_Id"])] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam" in r[2]][-(steps + 1):]
    for a, b in zip(ends[:-1], ends[1:]):
        seg = rows[a + 1:b + 1]
        cur_e, best = None, None
        for i, (s, e, n, q, c) in enumerate(seg):
            if cur_e is not None and s > cur_e and (best is None or s - cur_e > best[0]):
                best = (s - cur_e, i, cur_e)
            cur_e = e if cur_e is None else max(cur_e, e)
        gap, i, t_end = best
        print("step: largest idle gap %.1f us" % (gap / 1e3))
        for j in range(max(0, i - 6), min(len(seg), i + 3)):
            s, e, n, q, c = seg[j]
            extra = ""
            if c in api:
                f, hs, he = api[c]
                extra = "  host %s issued %.1f us before the kernel started (call took %.1f us)" % (f, (s - hs) / 1e3, (he - hs) / 1e3)
            print("  %s q%s  start %+9.1f  end %+9.1f  %s%s" % ("->" if j == i else "  ", q, (s - t_end) / 1e3, (e - t_end) / 1e3, n, extra))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
