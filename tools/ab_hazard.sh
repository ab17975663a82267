# A/B of the chunked hazard re-run (FASIM_HAZARD_CHUNKS=1, default) against the whole-unit re-run (=0): seconds per 50 Mb scan and
# the exclusive duration of the hazard family (kernel slot 1) in an isolated pass (one worker, one batch in flight).
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
  for ch in 0 1; do
    FASIM_HAZARD_CHUNKS=$ch python3 - <<PY
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(50_000_000, 12345)); p = mod.default_params()
eng.scan(None, p)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); r = eng.scan(None, p); ts.append(time.perf_counter() - t0); del r
print(f"pass $pass chunks $ch: mean {sum(ts)/len(ts):.3f} s  min {min(ts):.3f}  max {max(ts):.3f}", flush=True)
eng.set_option("workers", 1)
eng.scan(None, p, 0, 3840)
t0 = time.perf_counter(); r = eng.scan(None, p, 0, 3840); dt = time.perf_counter() - t0
s = r.stats; ik, il = s["kernel_ms"], s["kernel_launches"]
print(f"   isolated 10 batches: {dt:.3f} s; k_scan {ik[0]:.1f} ms/{il[0]}; hazard family {ik[1]:.1f} ms/{il[1]} launches; hazard units {s['hazard_units']}", flush=True)
PY
  done
done
