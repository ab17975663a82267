"""-F: records with the re-sweeps on the device (default) against the host restatement (FASIM_SIM_RESWEEP=host) on a planted record."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dna = synth.planted_dna(n, 4711, rna, every=3000)
p = mod.default_params(); p.classicSim = 1
out = {}
for mode in ("device", "host"):
    if mode == "host":
        os.environ["FASIM_SIM_RESWEEP"] = "host"
    t0 = time.perf_counter(); r = eng.scan(dna, p); dt = time.perf_counter() - t0
    out[mode] = r.triplexes()
    print(f"{mode}: {r.stats['units']} units, {len(out[mode])} records, {dt:.1f} s", flush=True)
print("identical" if out["device"] == out["host"] else "DIFFERENT", flush=True)
sys.exit(0 if out["device"] == out["host"] else 1)
