#!/usr/bin/env python3
"""Host-side split of one bench step: FASIM_PROFILE=1 python tools/prof_step.py  (engine phases on stderr)."""
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
dna = mod.synth_dna(int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 50_000_000, 12345)
eng.load_dna(dna)
p = mod.default_params()
for k in range(3):
    t0 = time.perf_counter()
    r = eng.scan(None, p)
    t1 = time.perf_counter()
    print(f"python: scan call {t1 - t0:.3f} s, engine t_total {r.stats['t_total_s']:.3f}, {r.count} records", file=sys.stderr)
    del r
    print(f"python: result free {time.perf_counter() - t1:.3f} s", file=sys.stderr)
