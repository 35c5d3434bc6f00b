#!/bin/bash
# HBM traffic per launch of the two dominant kernels bench.py names in its roofline objects, from the PMC counters (run on the GPU box):
#   head_x3_kernel<50> at B = 19200 (tools/bench_head.py) and gemm_mfma_kernel NT 38400 x 3072 x 768 bf16x3 (tools/bench_mm_one.py).
# Separate rocprofv3 --pmc passes per counter (FETCH_SIZE, WRITE_SIZE), no trace domains; FETCH_SIZE is doubled (gfx950 tallies the 128-B
# requests of 16-B/lane loads at 64 B: MI355X_MICROARCH.md, HBM section); counters are in KB (1024 B).
# usage: tools/pmc_traffic.sh [outdir]  -> writes <outdir>/r05_pmc_traffic.json (copy it to profiles/)
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/pmc_traffic}
mkdir -p $out
export TMPDIR=/tmp HI=5
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/head_$c -- python3 tools/bench_head.py > $out/head_$c.log 2>&1 || echo "head pass $c failed"
  rocprofv3 --pmc $c --output-format csv -d $out/gemm_$c -- python3 tools/bench_mm_one.py 38400 3072 768 > $out/gemm_$c.log 2>&1 || echo "gemm pass $c failed"
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def mean(tag, counter, needle):
    vals = []
    for f in glob.glob(f"{out}/{tag}_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and needle in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
res = {}
for key, tag, needle, extra in (("head_x3_kernel", "head", "head_x3_kernel", {"frames": 19200}),
                                ("gemm_mfma_kernel NT 38400x3072x768", "gemm", "gemm_img16_kernel", {"M": 38400, "N": 3072, "K": 768})):
    f, nf = mean(tag, "FETCH_SIZE", needle)
    w, nw = mean(tag, "WRITE_SIZE", needle)
    if f is None or w is None:
        continue
    res[key] = dict(extra, hbm_bytes_per_launch=int(2 * f * 1024 + w * 1024), fetch_size_kb_raw=round(f, 1), write_size_kb_raw=round(w, 1), launches=[nf, nw],
                    how="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.sh); 2 x FETCH_SIZE + WRITE_SIZE, KB = 1024 B")
json.dump(res, open(f"{out}/r05_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
