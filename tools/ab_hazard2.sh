# full-pipeline cost of the pieces of the chunked hazard re-run: seconds per 50 Mb scan, 6 scans per configuration, two passes
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
  for cfg in "FASIM_HAZARD_CHUNKS=0" "FASIM_HAZARD_CHUNKS=1" "FASIM_HAZARD_SNAP=0" "FASIM_HAZARD_SPREAD=0" "FASIM_HAZARD_SPREAD=0 FASIM_HAZARD_SNAP=0"; do
    env $cfg python3 - "$cfg" $pass <<'PY'
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
print(f"pass {sys.argv[2]} {sys.argv[1]:45s}: mean {sum(ts)/len(ts):.3f} s  min {min(ts):.3f}  max {max(ts):.3f}", flush=True)
PY
  done
done
