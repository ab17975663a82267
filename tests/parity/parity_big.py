#!/usr/bin/env python3
"""Large parity run on the GPU box: our `fasim` CLI (HIP path) against the compiled reference CLI
(oracle/_ref/fasim_ref) on the same synthetic FASTA; compares the -TFOsorted files byte for byte.

    python tests/parity/parity_big.py random 1000000 12345
    python tests/parity/parity_big.py planted 500000 4242 -lg 40
    FASIM_PARITY_RNA=tests/golden/MALAT1.fa python tests/parity/parity_big.py planted 1000000 7    (another query)
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402


def main():
    kind, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    extra = sys.argv[4:]
    rna_path = os.environ.get("FASIM_PARITY_RNA", os.path.join(ROOT, "tests", "golden", "H19.fa"))
    if not os.path.isabs(rna_path):
        rna_path = os.path.join(ROOT, rna_path)
    rna_name = open(rna_path).readline().strip().replace(">", "")
    _, rna = synth.read_fasta(rna_path)
    dna = synth.planted_dna(n, seed, rna) if kind == "planted" else synth.random_dna(n, seed)
    wd = tempfile.mkdtemp(prefix="parity_")
    try:
        synth.write_fasta(os.path.join(wd, "big.fa"), f"syn|chrB|1-{n}", dna)
        shutil.copyfile(rna_path, os.path.join(wd, "H19.fa"))     # (file name kept; the output name uses the header)
        os.makedirs(os.path.join(wd, "ref"))
        os.makedirs(os.path.join(wd, "gpu"))
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "fasim-longtarget_amd", "fasim"), "-f1", "big.fa", "-f2", "H19.fa", "-O", "gpu/",
                        "--stats"] + extra, cwd=wd, check=True, stdout=subprocess.DEVNULL)
        t_gpu = time.time() - t0
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fasim_ref"), "-f1", "big.fa", "-f2", "H19.fa", "-O", "ref/"] + extra,
                       cwd=wd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_ref = time.time() - t0
        name = f"syn-{rna_name}-big-TFOsorted"
        a = open(os.path.join(wd, "gpu", name), "rb").read()
        b = open(os.path.join(wd, "ref", name), "rb").read()
        same = a == b
        print(f"{rna_name} ({len(rna)} nt) x {kind} n={n} seed={seed} opts={extra}: identical={same} lines={a.count(10)} "
              f"sha256={hashlib.sha256(b).hexdigest()[:16]} t_gpu_cli={t_gpu:.1f}s t_ref_cpu={t_ref:.1f}s", flush=True)
        if not same:
            al, bl = a.split(b"\n"), b.split(b"\n")
            print("lines ours/ref:", len(al), len(bl))
            for i, (x, y) in enumerate(zip(al, bl)):
                if x != y:
                    print("first difference at line", i, "\n ours:", x[:200], "\n ref :", y[:200])
                    break
            sys.exit(1)
    finally:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()
