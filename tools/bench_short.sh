#!/bin/bash
# usage: tools/bench_short.sh <tag>   (run on the GPU box) - bench line + top kernels
tag=${1:-x}
timeout 500 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<EOF2
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print(d["value"], d["ms_per_step"], d["step_mfma_frac"])
for k in d["kernels"][:${2:-9}]: print(k["tag"], k["launches_per_step"], k["avg_us"], k["per_step_ms"])
EOF2
