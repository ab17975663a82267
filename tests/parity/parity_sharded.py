#!/usr/bin/env python3
"""Parity at the bench's full size: the 50 Mb synthetic record of bench.py (splitmix64 seed 12345) is cut into slices at
segment boundaries (multiples of 4900 nt, each slice = whole 5000-nt segments of the full record); every slice is run
through our `fasim` CLI (HIP path, one after the other) and through the compiled reference CLI (oracle/_ref/fasim_ref,
single-threaded, all slices side by side on the box's cores), and the -TFOsorted / -TFOclass files are compared byte
for byte.
    python tests/parity/parity_sharded.py [total_mb=50] [slices=13]"""
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
    total = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 50_000_000
    nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    dna = synth.random_dna(total, 12345)
    nseg = (total - 5000) // 4900 + 1
    per = (nseg + nsl - 1) // nsl
    wd = tempfile.mkdtemp(prefix="parity50_")
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "H19.fa"), os.path.join(wd, "H19.fa"))
    names = []
    for k in range(nsl):
        a, b = k * per, min(nseg, (k + 1) * per)
        if a >= b:
            break
        lo, hi = a * 4900, min(total, (b - 1) * 4900 + 5000)
        name = f"s{k:02d}"
        synth.write_fasta(os.path.join(wd, name + ".fa"), f"syn|chrB|{lo + 1}-{hi}", dna[lo:hi])
        os.makedirs(os.path.join(wd, "gpu_" + name))
        os.makedirs(os.path.join(wd, "ref_" + name))
        names.append(name)
    t0 = time.time()
    for name in names:                                   # GPU runs one after the other
        subprocess.run([os.path.join(ROOT, "fasim-longtarget_amd", "fasim"), "-f1", name + ".fa", "-f2", "H19.fa", "-O", f"gpu_{name}/"],
                       cwd=wd, check=True, stdout=subprocess.DEVNULL)
    t_gpu = time.time() - t0
    print(f"{len(names)} slices, {total / 1e6:.0f} Mb: our CLI {t_gpu:.1f} s in total (incl. process start-up, FASTA parsing, file output)", flush=True)
    t0 = time.time()
    procs = [subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", "fasim_ref"), "-f1", name + ".fa", "-f2", "H19.fa", "-O", f"ref_{name}/"],
                              cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for name in names]
    while any(p.poll() is None for p in procs):
        time.sleep(30)
        print(f"  reference running: {sum(p.poll() is None for p in procs)} of {len(procs)} left, {time.time() - t0:.0f} s", flush=True)
    t_ref = time.time() - t0
    ok, lines = True, 0
    h = hashlib.sha256()
    for name in names:
        for f in sorted(os.listdir(os.path.join(wd, "ref_" + name))):
            a = open(os.path.join(wd, "ref_" + name, f), "rb").read()
            b = open(os.path.join(wd, "gpu_" + name, f), "rb").read()
            h.update(a)
            if f.endswith("-TFOsorted"):
                lines += a.count(b"\n") - 1
            if a != b:
                ok = False
                print("DIFFERENT:", name, f)
    print(f"identical={ok} slices={len(names)} triplex_lines={lines} sha256={h.hexdigest()[:16]} t_gpu_cli_total={t_gpu:.1f}s "
          f"t_ref_wall={t_ref:.0f}s ({len(names)} reference processes side by side)", flush=True)
    shutil.rmtree(wd, ignore_errors=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
