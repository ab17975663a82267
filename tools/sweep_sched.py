#!/usr/bin/env python3
"""Scheduling sweep of the 50 Mb bench record: segments per batch x batches in flight -> seconds per scan."""
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
combos = [(512, 10), (512, 12), (512, 16), (384, 10), (384, 12), (384, 16), (256, 10), (256, 12), (256, 16), (640, 10), (768, 12), (512, 8)]
if len(sys.argv) > 1:
    combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for c in combos:
    sb, w = c[0], c[1]
    taper, gate = (c[2] if len(c) > 2 else 0), (c[3] if len(c) > 3 else 3)
    eng.set_option("seg_batch", sb)
    eng.set_option("workers", w)
    eng.set_option("taper", taper)
    eng.set_option("heavy_gate", gate)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        r = eng.scan(None, p)
        ts.append(time.perf_counter() - t0)
        n = r.count
        del r
    print(f"seg_batch {sb:4d} workers {w:2d} taper {taper:2d} gate {gate}: mean {sum(ts) / len(ts):.3f} min {min(ts):.3f} s (runs {' '.join(f'{t:.3f}' for t in ts)}), {n} records", flush=True)
