"""Deterministic synthetic inputs (shared by tests/, bench.py and tests/golden/make_golden.py).

splitmix64 -> 2 bits per base, 32 bases per 64-bit word, so the same sequence can be regenerated
anywhere (the C++ driver implements the same generator: fasim-longtarget_amd/csrc/synth.hpp).

    random_dna(n, seed)            i.i.d. uniform ACGT                      (SURVEY.md 8(d) "syn50M")
    planted_dna(n, seed, rna, ..)  random background + tracts that are (mutated) pre-images of
                                   lncRNA windows under random rule encodings, so that high-scoring
                                   hits, byte overflows (Q1), signed-lazy-F cases (Q2) and gapped
                                   window alignments are common.
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
    else:
        raise SystemExit("usage: synth.py random|planted|rna n seed out.fa [rna.fa]")
