#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: median/min duration per (kernel, grid size)."""
import csv
import collections
import sys


def main(path, flt=""):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        key = (short[:70], int(r.get("Grid_Size", r.get("Grid_Size_X", 0))))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid), ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        ts.sort()
        print("%-72s grid %8d  n %4d  med %8.1f us  min %8.1f us" % (name, grid, len(ts), ts[len(ts) // 2], ts[0]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
