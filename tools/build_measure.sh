#!/bin/bash
# usage (in the build container): tools/build_measure.sh  - libskf.so with the measurement knobs (-DSKF_MEASURE=1) as tools/tmp/libskf_measure.so;
# the default library is left in place.  On the GPU box: cp tools/tmp/libskf_measure.so sketchformer_amd/libskf.so before tools/ab_bench.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp/mobj
pids=()
for f in sketchformer_amd/csrc/*.hip; do
  o=tools/tmp/mobj/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DSKF_MEASURE=1 -c $f -o $o 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libskf_measure.so tools/tmp/mobj/*.o
echo built tools/tmp/libskf_measure.so
