import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import __graft_entry__ as entry, synth
mod = entry.load(); eng = mod.Engine(0)
_, rna = synth.read_fasta("tests/golden/H19.fa"); eng.set_query(rna)
eng.load_dna(mod.synth_dna(50_000_000, 12345)); p = mod.default_params()
eng.scan(None, p)
for _ in range(3):
    c0 = os.times()
    t0 = time.perf_counter(); r = eng.scan(None, p); t1 = time.perf_counter(); n = r.count; del r; t2 = time.perf_counter()
    c1 = os.times()
    print(f"cpu: user {c1.user - c0.user:.2f} s + system {c1.system - c0.system:.2f} s per scan = {(c1.user - c0.user + c1.system - c0.system) / (t1 - t0):.1f} cores busy on average", file=sys.stderr, flush=True)
    print(f"python: scan call {t1 - t0:.3f} s, result release {t2 - t1:.3f} s, records {n}", file=sys.stderr, flush=True)
