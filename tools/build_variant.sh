#!/bin/bash
# Build an ablation / experiment variant of the library: tools/build_variant.sh NAME [-DFLAG ...]  ->  tools/bin/abl/liba2f_NAME.so
# Only ONE translation unit is recompiled with the flags (VARIANT_SRC, default head_fused); the others come from build/obj (the
# objects of the regular build: run `python -c "import __graft_entry__ as g; g.build()"` first).  tools/bin is git-ignored but travels
# to the GPU box; load a variant with A2F_LIB_PATH=tools/bin/abl/liba2f_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
vs=${VARIANT_SRC:-head_fused}
src=audio2face-pytorch_amd/csrc
mkdir -p tools/bin/abl build/obj
all=$(python3 -c "import sys; sys.path.insert(0, '.'); import importlib; b = importlib.import_module('a2f_amd._build'); print(' '.join(s[:-4] for s in b.SOURCES))")
objs=""
for f in $all; do
  if [ $f = $vs ]; then objs="$objs build/obj/${f}_$name.o"; continue; fi
  objs="$objs build/obj/$f.o"
  [ -f build/obj/$f.o ] || { echo "missing build/obj/$f.o: run the regular build first"; exit 1; }
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function "$@" -o build/obj/${vs}_$name.o $src/$vs.hip
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/bin/abl/liba2f_$name.so $objs
echo tools/bin/abl/liba2f_$name.so
