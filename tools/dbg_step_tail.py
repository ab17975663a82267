#!/usr/bin/env python3
"""Debug aid (GPU box): where does the time between the end of the last kernel and the next step go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, synth
mod = entry.load()
_, rna = synth.read_fasta(os.path.join(ROOT, "tests/golden/H19.fa"))
n = 50_000_000
dna = mod.synth_dna(n, 12345)
e = mod.Engine(0); e.set_query(rna); e.load_dna(dna)
p = mod.default_params()
for it in range(3):
    t0 = time.perf_counter(); res = e.scan(None, p); t1 = time.perf_counter()
    merged = mod.gather_results(res, None, 0, 1, None) if hasattr(mod, "gather_results") else None
    t2 = time.perf_counter()
    print(f"scan() wall {t1 - t0:.3f} s, engine t_total {res.stats['t_total_s']:.3f} s, gather {t2 - t1:.3f} s, records {res.count}")
