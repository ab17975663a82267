# A/B of the two scheduling-shape changes of round 2 on the headline workload: seconds per 50 Mb scan, 8 scans per cell, two passes
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
  for ft in 256 512; do
    FASIM_FWD_THREADS=$ft python3 - <<PY
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(50_000_000, 12345)); p = mod.default_params()
eng.scan(None, p)
for sb in (512, 384, 512, 384):
    eng.set_option("seg_batch", sb)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = eng.scan(None, p); ts.append(time.perf_counter() - t0); del r
    print(f"pass $pass fwd_threads $ft seg_batch {sb}: mean {sum(ts)/len(ts):.3f} s  min {min(ts):.3f}  max {max(ts):.3f}", flush=True)
PY
  done
done
