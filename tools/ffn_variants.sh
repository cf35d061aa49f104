#!/bin/bash
# usage (in the build container): tools/ffn_variants.sh <mask> [<mask> ...]  - libskf.so variants with -DSKF_FFN_ABLATE=<mask> in
# tools/tmp/libskf_ffn_<mask>.so (only skf_ffn_fused.hip is recompiled; the other objects are those of the current build)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 $( [ "$m" = stamps ] && echo -DSKF_FFN_STAMPS=1 || echo -DSKF_FFN_ABLATE=$m ) \
     -c sketchformer_amd/csrc/skf_ffn_fused.hip -o tools/tmp/ffn_$m.o
  objs=$(ls sketchformer_amd/build/*.o | grep -v skf_ffn_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libskf_ffn_$m.so $objs tools/tmp/ffn_$m.o
  echo built tools/tmp/libskf_ffn_$m.so
done
