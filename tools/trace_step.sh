#!/bin/bash
# usage (on the GPU box): tools/trace_step.sh <tag> [ENV=.. ...]  - rocprofv3 kernel trace of the headline step (15 steps) -> per-kernel averages under
# concurrency (gpurun_out/<tag>_ktrace_stats.txt) and the step timeline of one step (gpurun_out/<tag>_timeline.txt)
tag=$1; shift
R=$PWD; export PYTHONPATH=$R
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python $R/bench.py --steps 15 --warmup 5 --no-extras --no-cpu-baseline --no-profile > /dev/null 2>&1
f=$(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/ktrace_summary.py $f > $R/gpurun_out/${tag}_ktrace_stats.txt 2>&1
python $R/tools/step_timeline.py $f > $R/gpurun_out/${tag}_timeline.txt 2>&1
head -30 $R/gpurun_out/${tag}_ktrace_stats.txt
