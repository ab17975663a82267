#!/usr/bin/env python3
"""One batch of segments ALONE on the GPU (one worker), for kernel traces and counters:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/iso_batch.py [segments] [rna.fa] [dna kind] [repeat]
Prints the scan's stats (kernel families in ms, band counters) as JSON on stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
import synth  # noqa: E402

nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rna_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "H19.fa")
kind = sys.argv[3] if len(sys.argv) > 3 else "random"
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 2
mod = entry.load()
eng = mod.Engine(0)
if rna_path.startswith("syn:"):
    rna = synth.random_rna(int(rna_path[4:]), 515)
else:
    _, rna = synth.read_fasta(rna_path)
eng.set_query(rna)
n = nseg * 4900 + 100
dna = mod.synth_dna(n, 12345) if kind == "random" else synth.genome_like(n, 12345, soft_mask=False)
eng.load_dna(dna)
p = mod.default_params()
eng.set_option("workers", 1)
eng.set_option("seg_batch", nseg)
for k in range(repeat):
    r = eng.scan(None, p, 0, nseg)
    st = r.stats
    if k == repeat - 1:
        print(json.dumps({"m": len(rna), "segments": nseg, "records": r.count, **{k2: v for k2, v in st.items()}}))
    del r
eng.close()
