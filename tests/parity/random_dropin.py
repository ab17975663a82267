#!/usr/bin/env python3
"""One-off randomized check of the drop-in entry points (GPU box): fasim_align_batch / fasim_pre_align_batch /
fasim_calc_score_once against the CPU oracle on a few thousand random (query, target) pairs with planted similarity,
N / U letters and lengths that are not multiples of 16.      python tests/parity/random_dropin.py [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import helpers  # noqa: E402
import synth  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    mod = entry.load()
    ob = os.path.join(ROOT, "oracle", "_build")
    if not os.path.exists(os.path.join(ob, "libfasim_oracle.so")):
        entry.build()
    o = helpers.Oracle(ob)
    rng = synth._Rng(seed)
    bad = 0
    total = 0
    for qi in range(6):
        m = [130, 333, 1000, 1599, 2812, 4100][qi] + rng.below(15)
        q = bytearray(synth.random_rna(m, seed * 100 + qi))
        for _ in range(m // 200):                               # a few N / U oddities in the query
            q[rng.below(m)] = ord("NU"[rng.below(2)])
        q = bytes(q)
        e = mod.Engine(0)
        e.set_query(q)
        wins, targets = [], []
        for k in range(400):
            L = 20 + rng.below(180)
            w = bytearray(synth.planted_dna(L + 40, seed * 7919 + qi * 1000 + k, q, every=60, min_len=15, max_len=min(120, m - 5), mut_pct=5 + rng.below(25), indel_pct=rng.below(8)))[:L]
            if rng.below(10) == 0:
                w[rng.below(L)] = ord("N")
            wins.append(bytes(w))
        for k in range(40):
            n = 200 + rng.below(1500)
            targets.append(synth.planted_dna(n, seed * 104729 + qi * 100 + k, q, every=300, max_len=min(140, m - 5)))
        als = e.align_batch(wins)
        for w, a in zip(wins, als):
            five, cig = o.align(q, w)
            total += 1
            if (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end) != five or a.cigar_string() != cig:
                bad += 1
                if bad <= 5:
                    print("ALIGN MISMATCH m=%d" % m, five, cig, "| ours", (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end), a.cigar_string())
        cols, s1 = e.pre_align_batch(targets)
        for t, c, s in zip(targets, cols, s1):
            total += 2
            if list(c) != o.pre_align(q, t):
                bad += 1; print("PRE_ALIGN MISMATCH m=%d n=%d" % (m, len(t)))
            if s != o.stage1_max(q, t):
                bad += 1; print("STAGE1 MISMATCH m=%d n=%d" % (m, len(t)), s, o.stage1_max(q, t))
        e.close()
        print(f"query {qi} (m={m}): done, {total} checks so far, {bad} mismatches", flush=True)
    print(f"checks={total} mismatches={bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
