#!/usr/bin/env python
"""Per-kernel MFMA pipe utilisation from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * number of SIMDs) (the formula of rocprofv3's derived
counter, evaluated per dispatch here; 256 CUs x 4 SIMDs on MI355X).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -- python bench.py ...
    python tools/pmc_mfma_util.py out/.../*_counter_collection.csv profiles/<tag>_mfma_util.json
"""
import collections
import csv
import json
import sys

SIMDS = 1024


def main(path, out):
    rows = {}
    for r in csv.DictReader(open(path)):
        d = rows.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0],
                                               "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ds = [d for d in rows.values() if d.get("GRBM_GUI_ACTIVE", 0) > 0 and d["us"] > 5]
    # GRBM_GUI_ACTIVE is the sum over the 8 XCDs and carries a constant per-dispatch profiling overhead:
    # fit  GUI = a + 8 * f * t  over all dispatches -> f = shader clock in cycles/us while the kernels run
    n = len(ds)
    st, sg = sum(d["us"] for d in ds), sum(d["GRBM_GUI_ACTIVE"] for d in ds)
    stt, stg = sum(d["us"] ** 2 for d in ds), sum(d["us"] * d["GRBM_GUI_ACTIVE"] for d in ds)
    slope = (n * stg - st * sg) / (n * stt - st * st)
    clock = slope / 8.0                                   # cycles per microsecond
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for d in ds:
        a = agg[d["name"]]
        a[0] += 1
        a[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += d["us"]
    res = {"_clock_ghz_from_gui_active_slope": clock / 1e3,
           "_note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration * clock * 1024 SIMDs); durations are those of the profiled (serialised) run"}
    for k, (cnt, m, t) in agg.items():
        res[k] = {"launches": cnt, "mfma_busy_cycles_per_launch": m / cnt, "us_per_launch_profiled": t / cnt,
                  "mfma_util_percent": 100.0 * m / (t * clock * SIMDS)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("shader clock from the GRBM_GUI_ACTIVE slope: %.2f GHz" % (clock / 1e3))
    for k, v in sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -kv[1]["us_per_launch_profiled"] * kv[1]["launches"])[:14]:
        print("%-58s n=%4d  %6.1f us  MFMA busy %10.0f cyc  util %5.1f %%" % (k[:58], v["launches"], v["us_per_launch_profiled"],
                                                                          v["mfma_busy_cycles_per_launch"], v["mfma_util_percent"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
