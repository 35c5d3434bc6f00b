set -e
mkdir -p gpurun_out
for e in 1 0; do
 for s in "38400 768 768" "38400 3072 768" "38400 2304 768" "38400 768 3072"; do
  echo "EPI32=$e plain: $(A2F_MM_EPI32=$e MM_PLAIN_EPILOGUE=1 python tools/bench_mm_one.py $s 20)"
  echo "EPI32=$e gelu : $(A2F_MM_EPI32=$e python tools/bench_mm_one.py $s 20)"
 done
done > gpurun_out/epi32_ab.txt 2>&1
cat gpurun_out/epi32_ab.txt
python -m pytest tests/test_gpu_mm.py -x -q -m gpu 2>&1 | tail -5
