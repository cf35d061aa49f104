#!/bin/bash
# usage (on the GPU box): tools/ffn_ablate_run.sh <mask> ...  - tools/ffn_bench.py with each ablation variant of the library
export PYTHONPATH=$PWD
cp sketchformer_amd/libskf.so /tmp/libskf_keep.so
echo "== default"; python tools/ffn_bench.py 2>/dev/null | grep fused
for m in "$@"; do
  cp tools/tmp/libskf_ffn_$m.so sketchformer_amd/libskf.so
  echo "== ablate $m"; python tools/ffn_bench.py 2>/dev/null | grep fused
done
cp /tmp/libskf_keep.so sketchformer_amd/libskf.so
