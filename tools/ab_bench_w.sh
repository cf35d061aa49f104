#!/bin/bash
# usage (on the GPU box): tools/ab_bench_w.sh <tag> <workload> "<ENV=1 for variant B>" [rounds] [steps] - interleaved A/B of one workload's step time
tag=$1; wl=$2; envb=$3; rounds=${4:-2}; steps=${5:-30}
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then pre="env $envb"; else pre=""; fi
    $pre python bench.py --workload $wl --steps $steps --warmup 5 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag $v round $r: %.4f ms/step' % d['ms_per_step'])"
  done
done
