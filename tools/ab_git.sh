#!/bin/bash
# usage (on the GPU box): tools/ab_git.sh <tag> <file with the OLD version of sketchformer_amd/engine.py> [rounds]
# interleaved A/B of the headline step time between the working tree (A) and the tree with that file swapped in (B)
tag=$1; old=$2; rounds=${3:-3}
cp sketchformer_amd/engine.py /tmp/engine_new.py
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = B ]; then cp $old sketchformer_amd/engine.py; else cp /tmp/engine_new.py sketchformer_amd/engine.py; fi
    python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag $v round $r: %.4f ms/step' % d['ms_per_step'])"
  done
done
cp /tmp/engine_new.py sketchformer_amd/engine.py
