#!/usr/bin/env python
"""Every kernel of the step against its floors: launches per step and average launch under concurrency (kernel-trace stats CSV of
tools/collect_profiles.sh), HBM bytes per launch by PMC (<tag>_pmc_traffic.json), the time of those bytes at 8 and 4.8 TB/s, MFMA-pipe busy
share (<tag>_mfma_util.json).
    python tools/kernel_floors.py <kernel_stats.csv> <pmc_traffic.json> <mfma_util.json> <steps in the trace> > profiles/<tag>_kernel_floors.txt"""
import csv
import json
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(stats, traffic, mfma, steps):
    steps = float(steps)
    tr = json.load(open(traffic))
    mu = json.load(open(mfma))
    per = tr.get("kernels", tr)
    tbytes = {}
    for k, v in per.items():
        if isinstance(v, dict):
            b = v.get("bytes_per_launch", v.get("hbm_bytes_per_launch", None))
            if b is None and "fetch_bytes_per_launch" in v:
                b = v["fetch_bytes_per_launch"] + v.get("write_bytes_per_launch", 0.0)
            if b is not None:
                tbytes[short(k)] = float(b)
    print("%-62s %7s %8s %8s %9s %8s %8s %7s" % ("kernel", "n/step", "avg us", "us/step", "MB/launch", "us@8TB/s", "us@4.8", "MFMA %"))
    rows = []
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        calls, avg = float(r["Calls"]), float(r["AverageDuration(us)"])
        b = tbytes.get(k)
        m = mu.get(k, {})
        util = m.get("mfma_util_percent", m.get("mfma_util"))
        if util is not None and "mfma_util_percent" not in m and util <= 1.0:
            util *= 100.0
        rows.append((calls * avg / steps, "%-62s %7.1f %8.1f %8.1f %9s %8s %8s %7s" % (
            k[:62], calls / steps, avg, calls * avg / steps, "%.1f" % (b / 1e6) if b else "-", "%.1f" % (b / 8e6) if b else "-",
            "%.1f" % (b / 4.8e6) if b else "-", "%.1f" % util if util is not None else "-")))
    for _, line in sorted(rows, reverse=True)[:40]:
        print(line)


if __name__ == "__main__":
    main(*sys.argv[1:5])
