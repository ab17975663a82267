#!/usr/bin/env python3
"""Heavy-kernel gate x worker engines on the 50 Mb bench record (batch shape fitted to the record, as in the default scan)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
import synth  # noqa: E402

mod = entry.load()
eng = mod.Engine(0)
_, rna = synth.read_fasta(os.path.join(ROOT, "tests", "golden", "H19.fa"))
eng.set_query(rna)
eng.load_dna(mod.synth_dna(50_000_000, 12345))
p = mod.default_params()
eng.scan(None, p)
combos = [(4, 10), (3, 10), (5, 10), (6, 10), (4, 12), (5, 12), (4, 8), (4, 10)]
if len(sys.argv) > 1:
    combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for gate, w in combos:
    eng.set_option("workers", w)
    eng.set_option("heavy_gate", gate)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        r = eng.scan(None, p)
        ts.append(time.perf_counter() - t0)
        n = r.count
        del r
    print(f"gate {gate} workers {w:2d}: mean {sum(ts) / len(ts):.3f} min {min(ts):.3f} s (runs {' '.join(f'{t:.3f}' for t in ts)}), {n} records", flush=True)
