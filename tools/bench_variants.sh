#!/bin/bash
# run tools/bench_head.py against several builds of the library: tools/bench_variants.sh NAME... (tools/bin/abl/liba2f_NAME.so)
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo -n "$v: "; A2F_LIB_PATH=tools/bin/abl/liba2f_$v.so timeout -k 10 120 python tools/bench_head.py 2>&1 | grep -E "head" | tr "\n" " "; echo
done
