# same-box comparison of source trees with the per-phase timers of one 50 Mb scan (sums over the batches in flight)
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  dir=${spec%%:*}; envs=""; [ "$spec" != "$dir" ] && envs=${spec#*:}
  ( cd $dir && env $envs python3 - "$spec" <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(50_000_000, 12345)); p = mod.default_params()
eng.scan(None, p)
for _ in range(3):
    t0 = time.perf_counter(); r = eng.scan(None, p); dt = time.perf_counter() - t0
    s = r.stats
    keys = [k for k in s if k.startswith("t_")]
    print(f"{sys.argv[1]:32s} {dt:.3f} s  " + " ".join(f"{k}={s[k]:.2f}" for k in keys) + "  kernel_ms=" + " ".join(f"{x:.0f}" for x in s["kernel_ms"]) + "  launches=" + " ".join(str(x) for x in s["kernel_launches"]), flush=True)
    del r
PY
  )
done
