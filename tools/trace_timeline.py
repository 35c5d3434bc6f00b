#!/usr/bin/env python3
"""Timeline figures from a rocprofv3 --kernel-trace CSV: wall span, union of busy intervals (any kernel running), per-queue kernel time, the largest idle gaps and
the kernels around them.  usage: trace_timeline.py <dir or kernel_trace.csv> [skip_fraction]   (skip_fraction: leading share of the trace to drop, e.g. warm-up)"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
for s, e, name, q, st in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = name
busy += cur_e - cur_s
per_q = collections.defaultdict(int)
for s, e, name, q, st in rows:
    per_q[(q, st)] += e - s
print(f"span {1e-6 * (t1 - t0):.2f} ms, busy (union) {1e-6 * busy:.2f} ms = {100.0 * busy / (t1 - t0):.1f} %, kernel time summed {1e-6 * sum(e - s for s, e, *_ in rows):.2f} ms, {len(rows)} kernels")
for k, v in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {k[0]} stream {k[1]}: {1e-6 * v:.2f} ms")
tot_gap = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)} totalling {1e-6 * tot_gap:.2f} ms; histogram (us): " + ", ".join(f"<{b}: {sum(1 for g in gaps if g[0] < b * 1000)}" for b in (2, 5, 10, 50, 200)))
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {1e-3 * g:8.1f} us between {a[:60]} -> {b[:60]}")
