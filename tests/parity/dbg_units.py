#!/usr/bin/env python3
"""Debug aid (GPU box): per-unit stage-1/2 summary of the systolic scan (FASIM_DEBUG_UNITS=1) against a golden scan
fixture.    python tests/parity/dbg_units.py demo.scan.gz testDNA.fa"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402

code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import __graft_entry__ as entry, synth
mod = entry.load()
_, rna = synth.read_fasta(os.path.join(%r, "tests/golden/H19.fa"))
_, dna = synth.read_fasta(sys.argv[1])
e = mod.Engine(0); e.set_query(rna)
e.scan(dna, mod.default_params(cLength=20))
''' % (ROOT, ROOT, ROOT)
gold = os.path.join(ROOT, "tests", "golden")
_, units = helpers.parse_scan(helpers.gunzip(os.path.join(gold, sys.argv[1])))
env = dict(os.environ, FASIM_DEBUG_UNITS="1", FASIM_WORKERS="1", FASIM_SEG_BATCH="100000")
r = subprocess.run([sys.executable, "-c", code, os.path.join(gold, sys.argv[2])], env=env, capture_output=True, text=True)
lines = [l for l in r.stderr.splitlines() if l.startswith("[unit]")]
print(len(lines), "unit lines,", len(units), "golden units", r.stderr[-400:] if not lines else "")
bad = 0
for l, u in zip(lines, units):
    m = re.match(r"\[unit\] (\d+) s1=(\d+) thr=(-?\d+) hits=(\d+) flags=(\d+)", l)
    s1, thr, nh, fl = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
    if (s1, thr, nh) != (u["stage1"], u["thr"], u["nhits"]):
        bad += 1
        if bad <= 12:
            print("MISMATCH", l, "| golden s1/thr/nhits", u["stage1"], u["thr"], u["nhits"], "seg", u["seg"], "enc", u["enc"])
print("mismatching units:", bad, " flagged:", sum(1 for l in lines if int(re.search(r"flags=(\d+)", l).group(1)) & 5))
