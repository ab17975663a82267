# same-box comparison of source trees: exclusive kernel time of k_scan with ONE worker (10 batches of 384 segments)
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for spec in "$@"; do
  dir=${spec%%:*}; envs=""; [ "$spec" != "$dir" ] && envs=${spec#*:}
  ( cd $dir && env $envs python3 - "$spec" <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(8_000_000, 12345)); p = mod.default_params()
eng.set_option("workers", 1)
eng.scan(None, p, 0, 768)
res = []
for _ in range(3):
    r = eng.scan(None, p, 0, 3840); s = r.stats
    res.append((s["kernel_ms"][0], s["kernel_ms"][2], s["kernel_launches"][0]))
    del r
print(f"{sys.argv[1]:32s} k_scan ms per 10 batches: " + " ".join(f"{a:.1f}" for a, b, c in res) + "   k_align_fwd: " + " ".join(f"{b:.1f}" for a, b, c in res), flush=True)
PY
  )
done
done
