#!/bin/bash
# Per-kernel means of hardware counters, ONE rocprofv3 --pmc pass per counter (no trace domains):
#   tools/pmc_run.sh OUTDIR "COUNTER_A COUNTER_B ..." python3 script.py args...
# prints: counter, kernel, launches, mean value.  The program after the counters is run directly (no env / shell hop).
cd "$(dirname "$0")/.."
out=$1; shift
counters=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for c in $counters; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- "$@" > $out/$c.log 2>&1 || echo "pass $c failed (see $out/$c.log)"
done
python3 - "$out" $counters <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f"{c:28s} {k:70s} launches {len(v):4d}  mean {sum(v)/len(v):16.0f}")
PY
