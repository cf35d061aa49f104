#!/bin/bash
# usage (on the GPU box): tools/ab_bench.sh <tag> "<ENV=1 for variant B>" [rounds]  - interleaved A/B of the headline step time
tag=$1; envb=$2; rounds=${3:-3}
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then pre="env $envb"; else pre=""; fi
    $pre python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag $v round $r: %.4f ms/step' % d['ms_per_step'])"
  done
done
