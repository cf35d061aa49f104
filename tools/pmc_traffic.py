#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB-ish units of 1024 bytes... they are reported
in KB (x1024 -> bytes); on gfx950 FETCH_SIZE reports HALF the bytes of a wide coalesced stream, so it is doubled.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/b_counter_collection.csv \\
                                gpurun_out/pmc_WRITE_SIZE/b_counter_collection.csv profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, out):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for name in sorted(set(f) | set(w)):
        fa = sum(f.get(name, [0])) / max(len(f.get(name, [])), 1)
        wa = sum(w.get(name, [0])) / max(len(w.get(name, [])), 1)
        res[name] = {"launches": len(f.get(name, [])), "fetch_kb_raw_avg": fa, "write_kb_avg": wa,
                     "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0,
                     "note": "FETCH_SIZE doubled (gfx950 wide-load correction), x1024"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print("%-60s n=%4d  %8.2f MB/launch" % (k[:60], v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
