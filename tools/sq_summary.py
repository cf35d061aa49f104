#!/usr/bin/env python
"""Per-kernel SQ summary of a train step from two rocprofv3 --pmc passes of bench.py (tools/collect_sq.sh):
pass 1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE,
pass 2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR.
    python tools/sq_summary.py <pass1 counter_collection.csv> <pass2 counter_collection.csv> > profiles/<tag>_sq_summary.txt"""
import collections
import csv
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[k] += 1
    return agg, n


def main(p1, p2):
    a1, n1 = load(p1)
    a2, n2 = load(p2)
    print("SQ counters of the train step, per kernel (two rocprofv3 --pmc passes of bench.py --steps 3): share of wave cycles with an instruction issued /")
    print("waiting for an instruction's operands / waiting on a counter or barrier; LDS bank-conflict share of the LDS-active cycles; VALU, SALU, LDS and")
    print("VMEM instructions per MFMA instruction; MFMA instructions per launch (thousands, all waves).")
    print("%-62s %6s %8s %8s %8s | %7s %7s %7s %7s %8s" % ("kernel", "issue%", "waitins%", "waitany%", "ldsconf%", "VALU/MF", "SALU/MF", "LDS/MF", "VMEM/MF", "MFMA k"))
    rows = []
    for k in a1:
        c1, c2 = a1[k], a2.get(k, {})
        wc = c1.get("SQ_WAVE_CYCLES", 0.0)
        mf = c2.get("SQ_INSTS_MFMA", 0.0)
        if wc <= 0 or mf <= 0:
            continue
        lds_act = c1.get("SQ_LDS_IDX_ACTIVE", 0.0)
        rows.append((wc, "%-62s %6.1f %8.1f %8.1f %8.1f | %7.2f %7.2f %7.2f %7.2f %8.0f" % (
            k[:62], 100 * c1.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c1.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c1.get("SQ_WAIT_ANY", 0) / wc,
            100 * c1.get("SQ_LDS_BANK_CONFLICT", 0) / lds_act if lds_act else 0.0, c2.get("SQ_INSTS_VALU", 0) / mf, c2.get("SQ_INSTS_SALU", 0) / mf,
            c2.get("SQ_INSTS_LDS", 0) / mf, (c2.get("SQ_INSTS_VMEM_RD", 0) + c2.get("SQ_INSTS_VMEM_WR", 0)) / mf, mf / max(n2[k], 1) / 1e3)))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
