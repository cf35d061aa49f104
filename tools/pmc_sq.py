"""Per-kernel SQ counter summary from a rocprofv3 --pmc counter_collection CSV (all counters of one pass)."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
for k, c in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k, "launches", n[k])
    for name, v in sorted(c.items()):
        print("   %-28s %14.0f per launch" % (name, v / n[k]))
