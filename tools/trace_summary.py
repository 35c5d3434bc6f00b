#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace CSV: calls, total / average / min / max duration.  The --stats summary groups
by kernel NAME only; the GEMM family launches one template instantiation at many shapes, so the roofline's dominant launch variant
(bench.py `roofline.kernel`) is looked up here by its grid (workgroups = ceil(M / BM) * ceil(N / BN) [* batch * split_k]).
usage: trace_summary.py <dir or kernel_trace.csv> [out.csv]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        g = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        w = tuple(int(r.get(k, 0) or 0) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        wgs = tuple(a // b if b else 0 for a, b in zip(g, w))
        acc[(r["Kernel_Name"], wgs, w[0])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(((k, v) for k, v in acc.items()), key=lambda kv: -sum(kv[1]))
total = sum(sum(v) for _, v in rows)
out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
wr = csv.writer(out)
wr.writerow(["Name", "Workgroups_XYZ", "Workgroup_Size", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for (name, wgs, w0), v in rows:
    wr.writerow([name, "x".join(str(x) for x in wgs), w0, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
