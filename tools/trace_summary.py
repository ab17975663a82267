#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, union busy time, idle gaps, overlap depth.
    python tools/trace_summary.py gpurun_out/trace/*/*_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += e - s
    per[n][1] += 1
print(f"span {(t1 - t0) / 1e6:.1f} ms, {len(rows)} kernels")
for n, (d, c) in sorted(per.items(), key=lambda x: -x[1][0])[:14]:
    print(f"  {d / 1e6:10.1f} ms {c:7d}  {n}")
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = sum(hist.values())
for k in sorted(hist):
    print(f"  depth {k}: {hist[k] / 1e6:9.1f} ms ({100 * hist[k] / tot:.1f} %)")

# how well are the two heavy kernels (k_scan, k_align_fwd) kept running?
heavy = [(s, e) for s, e, n in rows if "k_scan<" in n or "k_align_fwd" in n]
ev = []
for s, e in heavy:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
hist[0] += t1 - last
tot = sum(hist.values())
print("heavy kernels (k_scan / k_align_fwd) in flight:")
for k in sorted(hist):
    print(f"  {k}: {hist[k] / 1e6:9.1f} ms ({100 * hist[k] / tot:.1f} %)")
