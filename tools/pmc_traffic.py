#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/r0N_pmc_traffic.json.

On the GPU box (counters in their own runs, no trace domains mixed in):
    export FASIM_WORKERS=1 FASIM_SEG_BATCH=1024
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --dna-mb 5 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --dna-mb 5 --warmup 0 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/bench_pmc.json > gpurun_out/r0N_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md,
HBM / rocprofv3 section): fetch figures are doubled; the k_scan launch, which must read exactly units x tstride bytes
of target codes, serves as the check."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def fold(d, counter):
    out = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[k][0] += float(r["Counter_Value"])
            out[k][1] += 1
    return out


def main():
    fdir, wdir, bench_json = sys.argv[1:4]
    f, w = fold(fdir, "FETCH_SIZE"), fold(wdir, "WRITE_SIZE")
    b = json.load(open(bench_json))
    # the run = timed step(s) + warm-up + the untimed isolated pass of bench.py: count what the kernels really processed
    doc = {"command": "FASIM_WORKERS=1 FASIM_SEG_BATCH=1024 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
                      "--dna-mb 5 --warmup 0 --no-cpu-baseline (two separate passes)",
           "note": "KB as reported by rocprofv3, summed over the launches of each kernel; launches counts dispatches (a long query "
                   "takes one dispatch per query tile)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        doc["kernels"][k] = {"FETCH_SIZE_KB_sum": round(f[k][0], 1), "FETCH_SIZE_launches": f[k][1],
                             "WRITE_SIZE_KB_sum": round(w[k][0], 1), "WRITE_SIZE_launches": w[k][1]}
    units = b["counts"]["units"] + b["isolated_kernels"].get("units", 0)
    tries = b["counts"]["align_calls"] + b["isolated_kernels"].get("align_calls", 0)
    band_tries = b["counts"].get("band_tries", 0) + b["isolated_kernels"].get("band_tries", 0)
    for name, key, per in (("k_scan", "fasim::k_scan<", units), ("k_align_fwd", "fasim::k_align_fwd<", tries),
                           ("k_align_band", "fasim::k_align_band<", band_tries), ("k_band_decide", "fasim::k_band_decide", tries),
                           ("k_band_emit", "fasim::k_band_emit", tries)):
        fk = sum(v[0] for k, v in f.items() if k.startswith(key))
        wk = sum(v[0] for k, v in w.items() if k.startswith(key))
        doc[name] = {"fetch_KB": round(fk, 1), "write_KB": round(wk, 1), "per": "unit" if name == "k_scan" else ("band try" if name == "k_align_band" else "window try"),
                     "count_in_run": per,
                     "hbm_bytes_per_item": round((2 * fk + wk) * 1024 / max(1, per), 1),
                     "note": "fetch doubled (gfx950 FETCH_SIZE counts 64 B per 128-B request)"}
    doc["k_scan"]["hbm_bytes_per_unit"] = doc["k_scan"]["hbm_bytes_per_item"]
    json.dump(doc, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
