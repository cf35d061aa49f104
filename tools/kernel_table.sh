#!/bin/bash
# usage (on the GPU box): tools/kernel_table.sh [ENV=..]  - per-kernel table of the headline step from bench.py's launch profiler
env "$@" python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step %.4f' % d['ms_per_step'])
for k in d.get('kernels', []):
    print('%-44s n=%3d  %7.2f us  %6.1f TF  %6.2f TB/s' % (k['tag'][:44], k.get('launches_per_step', k.get('launches', 0)), k.get('avg_us', 0), k.get('tflops', 0) or 0, (k.get('gbs', 0) or 0)/1000))
"
