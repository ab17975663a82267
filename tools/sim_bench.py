"""Throughput of the -F path (classic SIM): seconds per unit on a small synthetic record."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
if len(sys.argv) > 2:
    eng.set_option("seg_batch", int(sys.argv[2]))      # segments per batch: several batches in flight on the worker engines
dna = mod.synth_dna(n, 4242)
p = mod.default_params(); p.classicSim = 1
t0 = time.perf_counter(); r = eng.scan(dna, p); dt = time.perf_counter() - t0
s = r.stats
print(f"-F: {n} nt, {s['units']} units, {len(r.triplexes())} records, {dt:.2f} s = {1e3 * dt / max(1, s['units']):.2f} ms per unit; "
      f"k_sim_forward {s['kernel_ms'][7]:.0f} ms in {s['kernel_launches'][7]} launches; t_stage2 {s['t_stage2_s']:.2f} t_stage3 {s['t_stage3_s']:.2f} t_host {s['t_host_s']:.2f}", flush=True)
