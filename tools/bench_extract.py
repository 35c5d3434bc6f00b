#!/usr/bin/env python3
"""ms_per_step of the headline and of every secondary entry of a bench.py JSON line (file argument): a one-line digest for A-B runs."""
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
out=[("head",r.get("ms_per_step"))]
for e in r.get("secondary",[]): out.append((e["model"], e["ms_per_step"]))
print(sys.argv[1], out)
