#!/usr/bin/env python3
"""Banded stage 3 against the full-height organisation on whole records: for every (lncRNA, DNA kind, seed) the records and string
pool of a scan with band = 1 must equal those with band = 0 byte for byte.

    python3 tools/band_selfcheck.py [Mb per record] [seeds]      (defaults: 50 Mb, seeds 12346..12352 = ranks 1..7 of the scaling run)"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
import synth  # noqa: E402

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
mod = entry.load()
_, h19 = synth.read_fasta(os.path.join(ROOT, "tests", "golden", "H19.fa"))
_, meg3 = synth.read_fasta(os.path.join(ROOT, "tests", "golden", "MEG3.fa"))
n = int(mb * 1e6)
cases = [("H19", h19, "random", 12346 + k, n) for k in range(nseeds)]
cases += [("H19", h19, "genome", 777, n // 2), ("H19", h19, "planted", 778, n // 4), ("MEG3", meg3, "genome", 779, n // 2),
          ("syn3k", synth.random_rna(3000, 7), "random", 780, n // 2), ("syn10k", synth.random_rna(10000, 515), "genome", 781, n // 10)]
bad = 0
for name, rna, kind, seed, size in cases:
    dna = mod.synth_dna(size, seed) if kind == "random" else (synth.genome_like(size, seed, soft_mask=False) if kind == "genome" else synth.planted_dna(size, seed, rna))
    out = []
    for band in (1, 0):
        e = mod.Engine(0)
        e.set_option("band", band)
        e.set_query(rna)
        e.load_dna(dna)
        t0 = time.perf_counter()
        r = e.scan(None, mod.default_params())
        dt = time.perf_counter() - t0
        out.append((hashlib.sha256(r.recs).hexdigest()[:16], hashlib.sha256(r.pool).hexdigest()[:16], r.count, dt, r.stats["band_proven"], r.stats["align_calls"], r.stats["cells_stage3"]))
        del r
        e.close()
    same = out[0][:3] == out[1][:3]
    bad += 0 if same else 1
    print(f"{name:7s} x {kind:7s} {size / 1e6:6.1f} Mb seed {seed}: records {out[0][2]:8d} sha {out[0][0]} / {out[0][1]}  band 1: {out[0][3]:.2f} s ({out[0][4]} band results for {out[0][5]} tries, "
          f"stage-3 cells {out[0][6] / 1e12:.3f}e12)  band 0: {out[1][3]:.2f} s ({out[1][6] / 1e12:.3f}e12)  {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
print("all identical" if not bad else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)
