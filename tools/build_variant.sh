#!/bin/bash
# usage (on the GPU box): tools/build_variant.sh <output .so> <extra hipcc flags...>  - builds libskf.so with extra -D flags into <output>, restores the default library
out=$1; shift
cp sketchformer_amd/libskf.so /tmp/libskf_default.so; cp sketchformer_amd/libskf.so.stamp /tmp/libskf_default.stamp
SKF_EXTRA_HIPCC_FLAGS="$*" python -m sketchformer_amd.build --force > /dev/null 2>&1
cp sketchformer_amd/libskf.so $out
cp /tmp/libskf_default.so sketchformer_amd/libskf.so; cp /tmp/libskf_default.stamp sketchformer_amd/libskf.so.stamp
