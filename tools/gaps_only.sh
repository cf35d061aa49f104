#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of the headline step -> per-step idle-gap analysis (tools/step_gaps.py) in gpurun_out/<tag>_step_gaps.txt
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/step_gaps.py $f 15 gaps > $R/gpurun_out/${tag}_step_gaps.txt
python $R/tools/ktrace_stats_csv.py $f $R/gpurun_out/${tag}_kernel_stats.csv
head -50 $R/gpurun_out/${tag}_step_gaps.txt
