# usage: tools/mm_variant_ab.sh VARIANT   (A-B of tools/bin/abl/liba2f_VARIANT.so against the regular build on four Transformer shapes, plain and GELU epilogue)
set -e
mkdir -p gpurun_out
v=$1
for lib in "" tools/bin/abl/liba2f_$v.so "" tools/bin/abl/liba2f_$v.so; do
 for s in "38400 768 768" "38400 3072 768" "38400 2304 768" "38400 768 3072"; do
  echo "${lib:-regular} plain: $(A2F_LIB_PATH=$lib MM_PLAIN_EPILOGUE=1 python tools/bench_mm_one.py $s 20 2>&1 | grep a2f_mm)"
 done
done > gpurun_out/mm_ab_$v.txt 2>&1
cat gpurun_out/mm_ab_$v.txt
