#!/bin/bash
# Run ON the GPU box: rocprofv3 --hip-trace --stats of the headline step -> host time per HIP API call (gpurun_out/<tag>_hip_api_stats.csv)
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ht_$tag -- python $R/bench.py "$@" --steps 40 --warmup 10 --no-profile --no-cpu-baseline --no-extras > /tmp/ht_$tag.json 2>/dev/null
f=$(find /tmp/ht_$tag -name "*hip_api_stats.csv" | head -1)
cp $f $R/gpurun_out/${tag}_hip_api_stats.csv
head -25 $f
python -c "
import json; d=json.load(open('/tmp/ht_$tag.json')); print('ms/step under hip-trace', d['ms_per_step'])"
