import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(4_000_000, 12345)); p = mod.default_params()
eng.set_option("workers", 1)
r = eng.scan(None, p, 0, 384)
print("---- second batch", file=sys.stderr, flush=True)
r = eng.scan(None, p, 384, 384)
print(r.stats["hazard_units"])
