#!/bin/bash
# usage (on the GPU box): tools/ab_bench5.sh <tag> "<ENV=1 for variant B>" [rounds] [extra bench args] - interleaved A/B, cfg5 step time
tag=$1; envb=$2; rounds=${3:-2}; extra=${4:-}
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then pre="env $envb"; else pre=""; fi
    $pre python bench.py --workload cfg5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-profile $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag $v round $r: %.3f ms/step' % d['ms_per_step'])"
  done
done
