#!/usr/bin/env python3
"""One step of a workload as the list of its kernel launches in start order, from a rocprofv3 --kernel-trace CSV: the step is the span between the last two
launches of a marker kernel (default: the input packing kernel that opens an Audio2Mesh step).  Prints start offset, duration, gap to the previous kernel's end.
usage: trace_step.py <dir or kernel_trace.csv> [marker substring]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "a2m_pack"
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 2:
    sys.exit(f"fewer than two launches of a kernel matching {marker!r}")
lo, hi = marks[-2], marks[-1]
t0, prev_end, busy = rows[lo][0], rows[lo][0], 0
print(f"step = launches {lo} .. {hi - 1} ({hi - lo} kernels), {1e-3 * (rows[hi][0] - t0):.1f} us")
for s, e, name in rows[lo:hi]:
    short = name.replace("void ", "").replace("a2f::", "")[:110]
    print(f"{1e-3 * (s - t0):9.1f} us  {1e-3 * (e - s):8.1f} us  gap {1e-3 * (s - prev_end):6.1f}  {short}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"kernel time {1e-3 * busy:.1f} us")
