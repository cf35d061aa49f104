#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> per-kernel stats CSV (what `--stats` would print), for profiles/."""
import collections
import csv
import sys


def main(src, dst):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in agg.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "MinDuration(us)", "MaxDuration(us)", "Percentage"])
        for name, ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(ts), "%.3f" % sum(ts), "%.3f" % (sum(ts) / len(ts)), "%.3f" % min(ts), "%.3f" % max(ts),
                        "%.3f" % (100.0 * sum(ts) / total)])
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
