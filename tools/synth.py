"""Deterministic synthetic inputs (shared by tests/, bench.py and tests/golden/make_golden.py).

splitmix64 -> 2 bits per base, 32 bases per 64-bit word, so the same sequence can be regenerated
anywhere (the C++ driver implements the same generator: fasim-longtarget_amd/csrc/synth.hpp).

    random_dna(n, seed)            i.i.d. uniform ACGT                      (SURVEY.md 8(d) "syn50M")
    planted_dna(n, seed, rna, ..)  random background + tracts that are (mutated) pre-images of
                                   lncRNA windows under random rule encodings, so that high-scoring
                                   hits, byte overflows (Q1), signed-lazy-F cases (Q2) and gapped
                                   window alignments are common.
    genome_like(n, seed, ..)       chromosome-like record (hg38 is not in the container): telomere N run,
                                   N gaps from 1 nt to whole segments, soft-masked (lower-case) interspersed
                                   repeats, microsatellites and purine / pyrimidine tracts (the sequence
                                   class triplexes form on).  See the function for the exact recipe.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def splitmix64_array(seed: int, count: int) -> np.ndarray:
    """count successive splitmix64 outputs for `seed` (vectorised, uint64 wrap-around)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + idx * np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(n: int, seed: int) -> bytes:
    words = splitmix64_array(seed, (n + 31) // 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((words[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n]
    return _BASES[codes].tobytes()


def random_rna(n: int, seed: int) -> bytes:
    return random_dna(n, seed ^ 0x5DEECE66D)


# outputs for DNA letters A,T,G,C per encoding, canonical order
RULE_OUT = [
    "TGGT", "GTTG", "TGCT", "GTTC", "TGTT", "GTTT", "TGGC", "GTCG", "TGCC", "GTCC", "TGTC", "GTCT",
    "GTTG", "TGGT", "GTTC", "TGCT", "GTTA", "TGAT", "GTCG", "TGGC", "GTCC", "TGCC", "GTCA", "TGAC",
    "GATG", "AGGT", "GATC", "AGCT", "GATA", "AGAT", "GACG", "AGGC", "GACC", "AGCC", "GACA", "AGAC",
    "GCTG", "CGGT", "GCTC", "CGCT", "GCTA", "CGAT", "GCCG", "CGGC", "GCCC", "CGCC", "GCCA", "CGAC",
]


def enc_reversed(enc: int) -> bool:
    return bool(enc & 1)


class _Rng:
    """tiny scalar splitmix64 stream (python ints) for the planting decisions"""

    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + _GOLDEN) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        return self.next() % n


def planted_dna(n: int, seed: int, rna: bytes, every: int = 1500, min_len: int = 25, max_len: int = 140,
                mut_pct: int = 12, indel_pct: int = 3) -> bytes:
    dna = bytearray(random_dna(n, seed))
    rng = _Rng(seed * 7919 + 13)
    m = len(rna)
    pos = rng.below(every)
    while pos + max_len + 8 < n:
        enc = rng.below(48)
        out = RULE_OUT[enc]
        pre = {}
        for base, o in zip("ATGC", out):
            pre.setdefault(o, []).append(base)
        ln = min_len + rng.below(max_len - min_len + 1)
        start = rng.below(max(1, m - ln))
        window = rna[start:start + ln]
        tract = bytearray()
        for ch in window:
            c = chr(ch)
            c = "T" if c == "U" else c
            r = rng.below(100)
            if r < indel_pct:
                continue                       # deletion in the DNA
            if r < 2 * indel_pct:
                tract.append(b"ACGT"[rng.below(4)])   # insertion
            if r < 2 * indel_pct + mut_pct or c not in pre:
                tract.append(b"ACGT"[rng.below(4)])
            else:
                cands = pre[c]
                tract.append(ord(cands[rng.below(len(cands))]))
        if enc_reversed(enc):
            tract.reverse()
        dna[pos:pos + len(tract)] = tract
        pos += len(tract) + rng.below(2 * every) + 1
    return bytes(dna[:n])


def genome_like(n: int, seed: int, every: int = 6000, telomere: int = 10000, soft_mask: bool = True) -> bytes:
    """Chromosome-like synthetic DNA (stands in for hg38 chr1 in BASELINE configs 3-5; SURVEY.md 8(d) "planted").

    Background: i.i.d. uniform ACGT (random_dna(n, seed)).  Features are laid over it left to right by a scalar
    splitmix64 stream (seed * 7919 + 17); after each feature the walk skips 1 .. 2*every nt.  Feature mix:
      40 %  interspersed repeat: 200-3000 nt of the background turned to lower case (soft-masking, as UCSC hg38)
      25 %  microsatellite: a random unit of 1-6 nt repeated to 30-400 nt, 3 % of the copies mutated;
            half of them lower case (simple repeats are soft-masked too)
      23 %  purine or pyrimidine tract: 20-300 nt over {A,G} or {C,T}, 8 % impurities -- the low-complexity
            class Hoogsteen / reverse-Hoogsteen rules map onto lncRNA motifs (byte overflow Q1 and the signed
            lazy-F exit Q2 become common, as on real promoters)
      10 %  small N gap of 1-50 nt (inside a segment: the separate stage-1 pass, N = -1 vs -4, Q4)
       2 %  large N gap of 5 000-40 000 nt (whole 5 kb segments of N: same_seq() skips)
    plus `telomere` N at the start when n >= 20 * telomere.  soft_mask=False returns the record upper-cased
    (what `fasim --upper` scans, and what the reference must be given: it does not upper-case its input)."""
    dna = bytearray(random_dna(n, seed))
    rng = _Rng(seed * 7919 + 17)
    if n >= 20 * telomere:
        dna[:telomere] = b"N" * telomere
    pos = (telomere if n >= 20 * telomere else 0) + rng.below(every)
    while pos + 64 < n:
        kind = rng.below(100)
        if kind < 40:
            ln = min(200 + rng.below(2801), n - pos)
            dna[pos:pos + ln] = bytes(dna[pos:pos + ln]).lower()
        elif kind < 65:
            unit = bytes(b"ACGT"[rng.below(4)] for _ in range(1 + rng.below(6)))
            ln = min(30 + rng.below(371), n - pos)
            tract = bytearray((unit * (ln // len(unit) + 1))[:ln])
            for _ in range(ln * 3 // 100):
                tract[rng.below(ln)] = b"ACGT"[rng.below(4)]
            if rng.below(2):
                tract = bytearray(bytes(tract).lower())
            dna[pos:pos + ln] = tract
        elif kind < 88:
            pair = b"AG" if rng.below(2) else b"CT"
            ln = min(20 + rng.below(281), n - pos)
            tract = bytearray(pair[rng.below(2)] for _ in range(ln))
            for _ in range(ln * 8 // 100):
                tract[rng.below(ln)] = b"ACGT"[rng.below(4)]
            dna[pos:pos + ln] = tract
        elif kind < 98:
            ln = min(1 + rng.below(50), n - pos)
            dna[pos:pos + ln] = b"N" * ln
        else:
            ln = min(5000 + rng.below(35001), n - pos)
            dna[pos:pos + ln] = b"N" * ln
        pos += ln + 1 + rng.below(2 * every)
    out = bytes(dna[:n])
    return out if soft_mask else out.upper()


def write_fasta(path: str, header: str, seq: bytes) -> None:
    """single record, single sequence line (the reference reader is O(lines x length) and
    accumulates multi-record files: SURVEY.md B1)"""
    with open(path, "wb") as f:
        f.write(b">" + header.encode() + b"\n" + seq + b"\n")


def read_fasta(path: str):
    header, parts = None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if header is not None:
                    break
                header = line[1:].decode()
            else:
                parts.append(line)
    return header, b"".join(parts)


if __name__ == "__main__":
    import sys
    kind, n, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if kind == "random":
        write_fasta(out, f"syn|chrS|1-{n}", random_dna(n, seed))
    elif kind == "planted":
        _, rna = read_fasta(sys.argv[5])
        write_fasta(out, f"syn|chrP|1-{n}", planted_dna(n, seed, rna))
    elif kind == "rna":
        write_fasta(out, f"synrna{seed}", random_rna(n, seed))
    elif kind in ("genome", "genome-upper"):
        write_fasta(out, f"syn|chrG|1-{n}", genome_like(n, seed, soft_mask=(kind == "genome")))
    else:
        raise SystemExit("usage: synth.py random|planted|rna|genome|genome-upper n seed out.fa [rna.fa]")
