#!/bin/bash
# Build an ablation / experiment variant of the library: tools/build_variant.sh NAME [-DFLAG ...]  ->  build/abl/liba2f_NAME.so
# Only head_fused.hip is recompiled with the flags; the other translation units are cached in build/obj.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=audio2face-pytorch_amd/csrc
mkdir -p build/abl build/obj
for f in capi gemm wgrad voca_trunk loss pack adam bn ff_decoder interp; do
  if [ ! -f build/obj/$f.o ] || [ $src/$f.hip -nt build/obj/$f.o ] || [ $src/a2f_common.h -nt build/obj/$f.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -o build/obj/$f.o $src/$f.hip &
  fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function "$@" -o build/obj/head_fused_$name.o $src/head_fused.hip
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o build/abl/liba2f_$name.so build/obj/{capi,gemm,wgrad,voca_trunk,loss,pack,adam,bn,ff_decoder,interp}.o build/obj/head_fused_$name.o
echo build/abl/liba2f_$name.so
