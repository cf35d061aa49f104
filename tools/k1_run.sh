#!/bin/bash
# on the GPU box: each experiment library of tools/tmp/k1 in place of libskf.so, tools/k1_experiment.py in a fresh process
cp sketchformer_amd/libskf.so /tmp/libskf_keep.so
for v in 0 1 2 3; do
  cp tools/tmp/k1/libskf_v$v.so sketchformer_amd/libskf.so
  echo "=== kind-1 variant $v"
  timeout 300 python tools/k1_experiment.py 30 2>&1 | tail -8
done
cp /tmp/libskf_keep.so sketchformer_amd/libskf.so
