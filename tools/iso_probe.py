"""Why does the exclusive k_scan time of a one-batch pass depend on what ran before?  (planted DNA)"""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load()
_, rna = synth.read_fasta("tests/golden/H19.fa")
dna = synth.planted_dna(50_000_000, 12345, rna)
p = mod.default_params()
def iso(eng, tag):
    eng.set_option("workers", 1); eng.set_option("seg_batch", 1024)
    s = eng.scan(None, p, 0, 1024).stats
    eng.set_option("workers", 0); eng.set_option("seg_batch", 0)
    print(f"{tag}: isolated k_scan {s['kernel_ms'][0]:.1f} ms in {s['kernel_launches'][0]} launch(es)", flush=True)
eng = mod.Engine(0); eng.set_query(rna); eng.load_dna(dna)
iso(eng, "fresh engine")
iso(eng, "again")
t0 = time.perf_counter(); eng.scan(None, p); print(f"full scan {time.perf_counter() - t0:.3f} s", flush=True)
iso(eng, "after a full scan (default batch shape)")
iso(eng, "again")
eng.set_option("seg_batch", 384); t0 = time.perf_counter(); eng.scan(None, p); print(f"full scan, batches of 384: {time.perf_counter() - t0:.3f} s", flush=True); eng.set_option("seg_batch", 0)
iso(eng, "after a full scan with batches of 384")
eng.set_option("seg_batch", 512); eng.set_option("taper", 25); t0 = time.perf_counter(); eng.scan(None, p); print(f"full scan, batches of 512 + taper: {time.perf_counter() - t0:.3f} s", flush=True); eng.set_option("seg_batch", 0); eng.set_option("taper", -1)
iso(eng, "after a full scan with batches of 512 + taper")
