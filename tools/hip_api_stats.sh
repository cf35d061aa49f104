#!/bin/bash
# Run ON the GPU box: HIP API call counts of the bench step (rocprofv3 --hip-trace --stats): how many event records / stream waits / launches a step issues.
#   tools/hip_api_stats.sh <tag> [extra bench.py arguments]
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/hip_$tag -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find /tmp/hip_$tag -name "*hip_api_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/${tag}_hip_api_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("HIP API calls of bench.py --steps 20 --warmup 5 (25 steps + set-up), per step = calls / 25")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:18]:
    print("%-34s %7d calls %8.1f per step  avg %8.2f us" % (r["Name"], int(r["Calls"]), int(r["Calls"]) / 25.0, float(r["AverageDuration(ns)"] if "AverageDuration(ns)" in r else r.get("AverageDuration(us)", 0)) / (1e3 if "AverageDuration(ns)" in r else 1)))
PY
cat $R/gpurun_out/${tag}_hip_api_stats.txt
