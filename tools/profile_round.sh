#!/bin/bash
# Round-end profile set on the GPU box (outputs under gpurun_out/prof_round/, copy the summaries into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command per workload (kernel_stats CSVs)
#   2. PMC traffic of the two kernels bench.py's roofline objects name (tools/pmc_traffic.sh)
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/prof_round
mkdir -p $out
export TMPDIR=/tmp
for w in voca faceformer audio2mesh; do
  echo "== rocprofv3 stats: $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --no-cpu-baseline --no-secondary --no-batch-points --steps $([ $w = voca ] && echo 200 || echo 10) > $out/$w.json 2> $out/$w.err || { echo "rocprof $w failed"; tail -5 $out/$w.err; }
  f=$(find $out/$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${w}_kernel_stats.csv && head -4 $out/${w}_kernel_stats.csv | cut -c1-160
  tail -1 $out/$w.json | cut -c1-200
done
bash tools/pmc_traffic.sh $out/pmc
