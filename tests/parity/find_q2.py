#!/usr/bin/env python3
"""Locate (segment, encoding) units in which the reference's signed lazy-F exit (Q2) changes the column maxima.
Uses the oracle twice per unit (faithful / unsigned exit).  Test tooling only."""
import ctypes, os, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, helpers

def main():
    kind, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    _, rna = synth.read_fasta(os.path.join(ROOT, "tests", "golden", "H19.fa"))
    dna = synth.planted_dna(n, seed, rna) if kind == "planted" else synth.random_dna(n, seed)
    o = helpers.Oracle(os.path.join(ROOT, "oracle", "_build"))
    L = o.lib
    L.fo_pre_align_noq2.restype = None
    L.fo_pre_align_noq2.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    units = []
    for si, start in enumerate(range(0, n, 4900)):
        seg = dna[start:start + 5000]
        for enc in range(48):
            units.append((si, enc, seg))
    def work(u):
        si, enc, seg = u
        t, _ = o.encode_unit(seg, enc)
        a = o.pre_align(rna, t)
        b = (ctypes.c_int * len(t))()
        L.fo_pre_align_noq2(rna, len(rna), t, len(t), b)
        b = list(b)
        if a != b:
            first = next(i for i in range(len(a)) if a[i] != b[i])
            return (si, enc, first, a[first], b[first])
        return None
    with cf.ThreadPoolExecutor(8) as ex:
        for r in ex.map(work, units):
            if r: print("Q2", *r, flush=True)
    print("done", len(units))
if __name__ == "__main__":
    main()
