#!/usr/bin/env python3
"""Time-resolved view of a rocprofv3 --kernel-trace CSV: per bin of `ms` milliseconds, the time-average number of heavy
kernels (k_scan / k_align_fwd / k_align_band) and of all other kernels in flight.  Shows ramp-up, steady state and drain
of a scan.      python tools/trace_timeline.py <kernel_trace.csv> [bin_ms]"""
import csv
import sys

path = sys.argv[1]
bin_ns = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 50_000_000
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        heavy = (("k_scan<" in n) and (", true>" not in n)) or ("k_align_fwd" in n) or ("k_align_band" in n)   # (k_scan<.., true> = the checkpoint pass of the hazard re-run: a few hundred waves)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), heavy))
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
nb = (t1 - t0) // bin_ns + 1
acc = [[0.0, 0.0] for _ in range(nb)]
for s, e, h in rows:
    b = (s - t0) // bin_ns
    while s < e:
        end = min(e, t0 + (b + 1) * bin_ns)
        acc[b][0 if h else 1] += end - s
        s = end
        b += 1
print(f"# bin {bin_ns / 1e6:g} ms; columns: t_ms, heavy kernels in flight (time average), other kernels in flight")
for b, (h, o) in enumerate(acc):
    print(f"{b * bin_ns / 1e6:8.0f} {h / bin_ns:6.2f} {o / bin_ns:6.2f}  " + "#" * int(round(10 * h / bin_ns)))
