#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh <tag> <older libskf.so> [rounds] [bench flags]
# interleaved A/B of the headline step time between the library of the working tree (A) and an older build of it (B)
tag=$1; old=$2; rounds=${3:-3}; shift 3
cp sketchformer_amd/libskf.so /tmp/libskf_new.so
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then cp $old sketchformer_amd/libskf.so; else cp /tmp/libskf_new.so sketchformer_amd/libskf.so; fi
    python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag $v round $r: %.4f ms/step' % d['ms_per_step'])"
  done
done
cp /tmp/libskf_new.so sketchformer_amd/libskf.so
