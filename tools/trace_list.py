#!/usr/bin/env python3
"""List the stage-3 kernels of a rocprofv3 kernel trace in launch order (second scan only): name, workgroups, ms."""
import csv
import glob
import sys
path = sys.argv[1]
files = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
scans = [i for i, r in enumerate(rows) if "k_scan<" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"]]
start = scans[-1] if scans else 0
tot = {}
for r in rows[start:]:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "").replace("fasim::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[short] = tot.get(short, 0) + d
    if any(k in n for k in ("band", "align_fwd", "k_scan<")):
        print(f"{short:40s} wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):7d} {d:8.3f} ms")
print("--- totals of the last scan")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:40s} {v:8.3f} ms")
