#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV and print, for the last `decoder forward -> next decoder forward` window, how much of the
decoder recurrence (ff_decoder_fwd/bwd kernels) ran concurrently with other kernels, plus the busy time per queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
t0 = ev[0][0]
dec = [e for e in ev if "ff_decoder_fwd" in e[2] or "ff_decoder_bwd_kernel" in e[2]]
print("kernels", len(ev), "decoder launches", len(dec))
adam = [e for e in ev if "adam" in e[2].lower()]
wins = []
for i in range(1, len(adam)):
    a0, a1 = adam[i - 1][1], adam[i][1]
    if any(a0 <= d[0] <= a1 for d in dec):
        wins.append((a0, a1))
if wins:
    a0, a1 = wins[-1]
    win = [e for e in ev if e[0] >= a0 and e[1] <= a1]
    print(f"last step window {(a1 - a0) / 1e6:.2f} ms, {len(win)} kernels")
    byq = collections.defaultdict(float)
    for s, e, n, q, st in win:
        byq[(q, st)] += (e - s) / 1e6
    for k, v in sorted(byq.items()):
        print("  queue/stream", k, f"busy {v:.2f} ms")
    for s, e, n, q, st in win:
        if "ff_decoder" in n and "fold" not in n:
            ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2, q2, st2 in win if (q2, st2) != (q, st))
            print(f"  {n[:40]:40s} start {(s - a0) / 1e6:8.2f} ms dur {(e - s) / 1e6:6.2f} ms, other-queue kernel time inside it {ov / 1e6:6.2f} ms")
    # union busy time
    iv = sorted((s, e) for s, e, *_ in win)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    tot += ce - cs
    print(f"  union busy {tot / 1e6:.2f} ms; idle {(a1 - a0 - tot) / 1e6:.2f} ms")
    top = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q, st in win:
        k = n.split("(")[0][:60]
        top[k][0] += 1; top[k][1] += (e - s) / 1e6
    for k, (c, v) in sorted(top.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {v:8.2f} ms {c:5d}x {k}")
