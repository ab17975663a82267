#!/usr/bin/env python3
"""Whole-genome-shaped end-to-end run of the CLI on one GPU: 24 records, 3.09 Gb, one 3 kb lncRNA.

    python3 tools/genome_run.py [scale] [devices]   -> gpurun_out/genome_run_<devices>.txt (written as the run goes)"""
import glob
import hashlib
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = sys.argv[2] if len(sys.argv) > 2 else "0"
work = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fasim_genome")
outdir = os.path.join(work, "out_" + dev.replace(",", "_"))
os.makedirs(outdir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
report = os.path.join(ROOT, "gpurun_out", f"genome_run_{dev.replace(',', '_')}.txt")
genome = os.path.join(work, "genome.fa")
with open(report, "w") as rep:
    def say(s):
        rep.write(s + "\n"); rep.flush(); print(s, flush=True)
    if not os.path.exists(genome):
        t0 = time.time()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), genome, str(scale)], check=True)
        say(f"# genome: tools/make_genome.py scale {scale}: {os.path.getsize(genome)} bytes of FASTA in {time.time() - t0:.1f} s")
    lnc = os.path.join(work, "lnc3k.fa")
    open(lnc, "wb").write(b">syn3k_1\n" + synth.random_rna(3000, 1) + b"\n")
    # (-f1 is a bare file name in the working directory, as for the reference: the output names are built from it, B9)
    cmd = [os.path.join(ROOT, "fasim-longtarget_amd", "fasim"), "-f1", "genome.fa", "-f2", "lnc3k.fa", "-O", os.path.basename(outdir), "--upper",
           "--all-records", "--stats", "--devices", dev]
    say("# " + " ".join(cmd))
    t0 = time.time()
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=work)
    for line in pr.stdout:
        say(line.rstrip("\n"))
    rc = pr.wait()
    wall = time.time() - t0
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    say(f"# exit code {rc}; wall {wall:.1f} s; peak RSS of the CLI {ru.ru_maxrss / 1048576:.2f} GiB; user {ru.ru_utime:.1f} s, sys {ru.ru_stime:.1f} s")
    files = sorted(glob.glob(os.path.join(outdir, "*")))
    nbytes = sum(os.path.getsize(f) for f in files)
    h = hashlib.sha256(); lines = 0
    for f in sorted(glob.glob(os.path.join(outdir, "*-TFOsorted"))):
        with open(f, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk); lines += chunk.count(b"\n")
    say(f"# outputs: {len(files)} files, {nbytes} bytes; -TFOsorted lines (with 24 header lines): {lines}; sha256 of the -TFOsorted files in name order: {h.hexdigest()[:16]}")
sys.exit(rc)
