#!/usr/bin/env python3
"""Debug aid (GPU box): cost of the host tail (cluster + -TFOsorted / -TFOclass text) on the bench's 50 Mb result."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, synth
mod = entry.load()
_, rna = synth.read_fasta(os.path.join(ROOT, "tests/golden/H19.fa"))
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 50_000_000
dna = mod.synth_dna(n, 12345)
e = mod.Engine(0); e.set_query(rna); e.load_dna(dna)
p = mod.default_params()
res = e.scan(None, p)
t0 = time.perf_counter(); txt = mod.tfosorted(res, "chrB", 1, p); t1 = time.perf_counter()
c1 = mod.tfoclass(res, 1, "chrB", 1, n, "H19", p); t2 = time.perf_counter()
print(f"records {res.count}: tfosorted {t1 - t0:.3f} s ({len(txt) / 1e6:.1f} MB), tfoclass1 {t2 - t1:.3f} s ({len(c1) / 1e6:.1f} MB), scan {res.stats['t_total_s']:.3f} s")
