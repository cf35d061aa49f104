#!/bin/bash
# Run ON the GPU box (via gpurun): kernel-trace stats + two separate PMC passes of the same bench command,
# summaries written under gpurun_out/ (copy what should be judged into profiles/).
#   tools/collect_profiles.sh <tag> [extra bench.py arguments, e.g. --workload cfg5]
tag=${1:-x}
shift
extra="$@"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py $extra --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}_bench.json 2>/dev/null
f=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/ktrace_stats_csv.py $f $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/step_gaps.py $f 15 > $R/gpurun_out/${tag}_step_gaps.txt
fs=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1); [ -n "$fs" ] && cp $fs $R/gpurun_out/${tag}_rocprofv3_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -- python $R/bench.py $extra --steps 3 --warmup 1 --no-profile --no-cpu-baseline --no-extras > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_pmc_traffic.json
cat $R/gpurun_out/${tag}_step_gaps.txt
