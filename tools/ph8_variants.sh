#!/bin/bash
# usage (build container): tools/ph8_variants.sh <mask> ...  - libskf.so with -DSKF_PH8_ABLATE=<mask> as tools/tmp/libskf_ph8_<mask>.so (only skf_bf16_gemm.hip is rebuilt)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp/ph8
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DSKF_PH8_ABLATE=$m -c sketchformer_amd/csrc/skf_bf16_gemm.hip -o tools/tmp/ph8/gemm_$m.o 2>/dev/null &
done
wait
for m in "$@"; do
  objs=$(ls sketchformer_amd/build/*.o | grep -v skf_bf16_gemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libskf_ph8_$m.so $objs tools/tmp/ph8/gemm_$m.o
done
echo built "$@"
