#!/bin/bash
# HBM traffic of the head kernel per launch: two separate rocprofv3 --pmc passes (no trace domains) over tools/bench_head.py.
# usage (on the GPU box): tools/pmc_head.sh [outdir]   -> prints per-kernel mean FETCH_SIZE / WRITE_SIZE (KB)
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/pmc}
mkdir -p $out
export TMPDIR=/tmp HI=5
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 tools/bench_head.py > $out/$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "head" in k:
            print(f"{c:11s} {k:60s} launches {len(v):3d}  mean {sum(v)/len(v):14.0f} KB")
PY
