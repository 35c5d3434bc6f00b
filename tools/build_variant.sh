#!/bin/bash
# Build an ablation / experiment variant of the library: tools/build_variant.sh NAME [-DFLAG ...]  ->  build/abl/liba2f_NAME.so
# Only ONE translation unit is recompiled with the flags (VARIANT_SRC, default head_fused); the others are cached in build/obj.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
vs=${VARIANT_SRC:-head_fused}
src=audio2face-pytorch_amd/csrc
mkdir -p build/abl build/obj
all="capi gemm wgrad voca_trunk loss pack adam bn ff_decoder interp head_fused"
objs=""
for f in $all; do
  if [ $f = $vs ]; then objs="$objs build/obj/${f}_$name.o"; continue; fi
  objs="$objs build/obj/$f.o"
  if [ ! -f build/obj/$f.o ] || [ $src/$f.hip -nt build/obj/$f.o ] || [ $src/a2f_common.h -nt build/obj/$f.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -o build/obj/$f.o $src/$f.hip &
  fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function "$@" -o build/obj/${vs}_$name.o $src/$vs.hip
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o build/abl/liba2f_$name.so $objs
echo build/abl/liba2f_$name.so
