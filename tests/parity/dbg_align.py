import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as entry, synth, helpers
m = entry.load()
e = m.Engine(0)
_, rna = synth.read_fasta('/root/repo/tests/golden/H19.fa')
e.set_query(rna)
o = helpers.Oracle('/root/repo/oracle/_build')
wins = [synth.planted_dna(150, 100+k, rna, every=40, max_len=100)[:60+ (k*7)%130] for k in range(64)]
try:
    als = e.align_batch(wins)
except Exception as ex:
    print("ERR", ex); sys.exit(1)
bad = 0
for w, a in zip(wins, als):
    five, cig = o.align(rna, w)
    got = (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end)
    if got != five or a.cigar_string() != cig:
        bad += 1
        print("MISMATCH", got, a.cigar_string(), "exp", five, cig)
print("bad", bad, "of", len(wins))
_, dna = synth.read_fasta('/root/repo/tests/golden/planted40k.fa')
r = e.scan(dna, m.default_params(cLength=40))
print({k: r.stats[k] for k in ('units','candidates','align_calls','align_word_reruns','hazard_units','stage2_overflow_units','kernel_ms','kernel_launches','t_total_s','t_stage2_s','t_stage3_s','t_host_s')})
