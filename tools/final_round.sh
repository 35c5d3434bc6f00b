#!/bin/bash
# Round-end evidence on one GPU box: the -m gpu suite, the default bench line (as the driver runs it) and rocprofv3 kernel statistics of the three workloads.
# Outputs under gpurun_out/final/ (copy what is to be judged into profiles/).  usage: bash tools/final_round.sh
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/final
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 && tail -1 $out/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err && tail -c 300 $out/bench_default.json
for w in voca faceformer audio2mesh; do
  echo "== rocprofv3 stats: $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --no-cpu-baseline --no-secondary --no-batch-points --steps $([ $w = voca ] && echo 200 || echo 10) > $out/$w.json 2> $out/$w.err || { echo "rocprof $w failed"; tail -5 $out/$w.err; }
  f=$(find $out/$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${w}_kernel_stats.csv && head -4 $out/${w}_kernel_stats.csv | cut -c1-160
  rm -rf $out/$w
done
