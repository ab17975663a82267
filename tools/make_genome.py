#!/usr/bin/env python3
"""hg38-shaped stand-in genome (hg38 itself is not in the image): 24 records with the chromosome lengths of GRCh38, each a
synth.genome_like() sequence (soft-masked repeats, N gaps, telomeres, microsatellites, purine tracts), 60-column lines.

    python3 tools/make_genome.py OUT.fa [scale]      scale < 1 shrinks every chromosome (default 1.0 = 3.09 Gb)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402

HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
        ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
        ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
        ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]


def main():
    out = sys.argv[1]
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    t0 = time.time()
    total = 0
    with open(out, "wb") as f:
        for k, (name, n) in enumerate(HG38):
            n = max(10000, int(n * scale))
            seq = synth.genome_like(n, 1000 + k)
            f.write(f">hg38like|{name}|1-{n}\n".encode())
            # 60-column lines without a Python loop: reshape and join
            import numpy as np
            a = np.frombuffer(seq, dtype=np.uint8)
            full = (n // 60) * 60
            body = np.empty((n // 60, 61), dtype=np.uint8)
            body[:, :60] = a[:full].reshape(-1, 60)
            body[:, 60] = 10
            f.write(body.tobytes())
            if full < n:
                f.write(a[full:].tobytes() + b"\n")
            total += n
            print(f"{name}: {n} nt ({time.time() - t0:.1f} s)", file=sys.stderr, flush=True)
    print(f"wrote {out}: {total} nt in {len(HG38)} records, {time.time() - t0:.1f} s", file=sys.stderr)


if __name__ == "__main__":
    main()
