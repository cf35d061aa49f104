#!/bin/bash
# usage (on the GPU box): tools/flag_sweep.sh <out.txt> "<flags of variant 1>" "<flags of variant 2>" ...
# rebuilds libskf.so with SKF_EXTRA_HIPCC_FLAGS per variant (the default build first and last) and prints the headline ms/step
out=$1; shift
run() {
  for r in 1 2; do
    SKF_EXTRA_HIPCC_FLAGS="$1" python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-profile $BENCH_EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[%s] run $r: %.4f ms/step' % ('$1', d['ms_per_step']))" >> $out
  done
}
: > $out
run ""
for f in "$@"; do run "$f"; done
run ""
cat $out
