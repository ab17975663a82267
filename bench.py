#!/usr/bin/env python3
"""bench.py -- headline benchmark of the triplex-scan hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): H19 (2 812 nt, tests/golden/H19.fa) x synthetic uniform-random DNA
(splitmix64 seed 12345 + rank, `--dna-mb` million nt, default 50) with the reference's default parameters:
10 205 segments x 48 rule encodings = 489 840 work units per 50 Mb.  One STEP = one complete pass of the hot
path over that record: stage-1 max score, stage-2 column maxima, candidate picking, every window alignment
with traceback, triplex records -- i.e. everything fasim_scan() does.  The DNA record is uploaded once before
the timed region and stays resident in HBM.  With N ranks every rank scans its own record (weak scaling,
no data-path collective) and the records are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  value = logical SW Gcells/s of the whole job =
sum_ranks(m * sum(len(segment)) * 48) * K / max_rank(time) / 1e9   (SURVEY.md 8(d)).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything initialises HIP (see fasim_engine_create)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import __graft_entry__ as entry  # noqa: E402
import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TOPS = 78.6           # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz: plain VALU, one lane-instruction per lane per clock
# The DP kernels consist of packed 16-bit ops (v_pk_add_i16 / v_pk_max_* / v_pk_sub_u16).  Measured with
# rocprofv3 --pmc on k_scan and k_align_fwd (profiles/r01_sq_counters.txt): SQ_INSTS_VALU == SQ_BUSY_CU_CYCLES, i.e. one
# wave64 VALU instruction per CU clock = 4 clocks per instruction and SIMD with four waves resident: the packed ops
# issue at half the plain rate, and at that rate the kernels sit at the issue limit.
VALU_PK16_PEAK_TOPS = 39.3      # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (lane-instructions/s of packed 16-bit ops)
KERNEL_NAMES = ["k_scan (fused stage 1+2)", "k_striped<PRE|MAX1> (stage 1/2 hazard re-runs)", "k_align_fwd (stage 3 forward)",
                "k_finish_lds (reverse pass + traceback)", "k_encode/k_scan_post/k_hits/k_build_stream",
                "k_striped<ALIGN|REV> (stage 3 exact replays)", "k_finish/k_banded (global scratch)", "-"]


def cpu_baseline(rna_path, m, sample_nt, seed):
    """The reference's own SSE2 binary (oracle/_ref/fasim_ref, built from /root/reference by oracle/Makefile)
    on the first `sample_nt` nt of the same synthetic record, 1 thread (the reference is single-threaded).
    Falls back to the oracle port when the reference binary did not travel."""
    ref = os.path.join(ROOT, "oracle", "_ref", "fasim_ref")
    dna = synth.random_dna(sample_nt, seed)
    nseg_cells = sum(min(5000, sample_nt - s) for s in range(0, sample_nt, 4900))
    cells = m * nseg_cells * 48
    wd = tempfile.mkdtemp(prefix="fasim_cpu_")
    try:
        synth.write_fasta(os.path.join(wd, "sample.fa"), f"syn|chrS|1-{sample_nt}", dna)
        shutil.copyfile(rna_path, os.path.join(wd, "H19.fa"))
        os.makedirs(os.path.join(wd, "out"))
        if os.access(ref, os.X_OK):
            t0 = time.perf_counter()
            subprocess.run([ref, "-f1", "sample.fa", "-f2", "H19.fa", "-O", "out/"], cwd=wd, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            kind = "reference"
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
            exe = os.path.join(ROOT, "oracle", "_build", "fasim_oracle")
            t0 = time.perf_counter()
            subprocess.run([exe, "tfosorted", "H19.fa", "sample.fa"], cwd=wd, check=True, stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            kind = "port"
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    return {"value": round(cells / dt / 1e9, 4), "unit": "Gcells/s", "cores": 1, "kind": kind,
            "sample": f"H19 x first {sample_nt} nt of the same synthetic record, default parameters, {dt:.1f} s wall",
            "mbp_per_s": round(sample_nt / dt / 1e6, 5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dna-mb", type=float, default=50.0, help="million nt of synthetic DNA per rank (default 50)")
    ap.add_argument("--cpu-sample-nt", type=int, default=250000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses device 0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    xdev = "cuda" if args.backend == "nccl" else "cpu"      # where the exchanged record bytes live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    mod = entry.load()
    eng = mod.Engine(local)
    rna_path = os.path.join(ROOT, "tests", "golden", "H19.fa")
    _, rna = synth.read_fasta(rna_path)
    eng.set_query(rna)
    n = int(args.dna_mb * 1e6)
    dna = mod.synth_dna(n, 12345 + rank)
    eng.load_dna(dna)                      # resident in HBM before the timed region
    p = mod.default_params()

    def step():
        res = eng.scan(None, p)
        # the path's one exchange step: gather every rank's records on rank 0 (RCCL over xGMI)
        merged = mod.gather_results(res, dist, rank, world, xdev)
        return res, (merged.count if merged is not None else 0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    agg = None
    nrec = 0
    for _ in range(args.steps):
        res, nrec = step()
        st = res.stats
        if agg is None:
            agg = {k: (list(v) if isinstance(v, list) else v) for k, v in st.items()}
        else:
            for k, v in st.items():
                if isinstance(v, list):
                    agg[k] = [a + b for a, b in zip(agg[k], v)]
                else:
                    agg[k] += v
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
    cells = torch.tensor([float(agg["logical_cells"])], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cells, op=dist.ReduceOp.SUM)
    tmax = float(tmax.item())
    total_cells = float(cells.item())

    # untimed extra pass for the kernel-quality figures: ONE batch in flight, so HIP-event durations are those of a
    # kernel that has the GPU to itself (in the timed steps six batches share it and every duration is stretched)
    iso = None
    if rank == 0:
        eng.set_option("workers", 1)
        eng.set_option("seg_batch", 1024)
        nseg_iso = min(1024, mod.segment_count(n, p))
        iso = eng.scan(None, p, 0, nseg_iso).stats
        eng.set_option("workers", 0)
        eng.set_option("seg_batch", 0)

    if rank == 0:
        m = len(rna)
        units_per_step = agg["units"] / args.steps
        kms = agg["kernel_ms"]
        n_avg = agg["cells_stage2"] / max(1, agg["units"]) / m        # mean segment length
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        except Exception:
            pmc = {}

        def kernel_roofline(idx):
            """HBM roofline of one of the two DP kernels: algorithmic bytes per launch / HIP-event launch duration."""
            launches_k = max(1, agg["kernel_launches"][idx])
            avg = kms[idx] / launches_k
            if idx == 0:   # k_scan, per unit: segment codes in (n) + u16 column maxima out (2n); the profile is LDS-resident
                per_item, items, key = 3 * n_avg, agg["units"] + agg["stage1_word_reruns"], "k_scan"
            else:          # k_align_fwd, per window try: L + 2 stream bytes in, 16-byte descriptor in, 24-byte result out
                calls = max(1, agg["align_calls"])
                per_item, items, key = agg["cells_stage3"] / m / calls + 2 + 16 + 24, calls + agg["align_word_reruns"], "k_align_fwd"
            alg = per_item * items / launches_k
            tr = pmc.get(key, {}).get("hbm_bytes_per_item")
            return {"bound": "hbm", "kernel": KERNEL_NAMES[idx], "achieved": round(alg / (avg * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "traffic": int(tr * items / launches_k) if tr else None, "avg_launch_ms": round(avg, 3), "launches": int(launches_k),
                    "algorithmic_bytes_per_launch": int(alg),
                    "note": "integer DP is VALU-bound by construction; see valu"}

        # dominant kernel: k_scan (fused stage 1+2: most DP cells; its time and k_align_fwd's are within a few percent)
        dom = 0
        launches = max(1, agg["kernel_launches"][dom])
        avg_ms = kms[dom] / launches
        cells_dom, ops_per_cell = agg["cells_stage2"], 4.5
        traffic_per_unit = pmc.get("k_scan", {}).get("hbm_bytes_per_unit")
        out = {
            "metric": "SW Gcells/s (logical, whole job: stage 1+2+3 of the triplex scan)",
            "value": round(total_cells / tmax / 1e9, 3),
            "unit": "Gcells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(tmax / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"H19 (2812 nt) x synthetic {args.dna_mb:g} Mb DNA per GPU (splitmix64 seed 12345+rank), "
                                   "default parameters, 48 rule encodings, records gathered to rank 0",
                       "units_per_step_per_gpu": int(units_per_step), "records_rank0": int(nrec)},
            "mbp_per_s": round(n * world * args.steps / tmax / 1e6, 3),
            "executed_gcells_per_s": round((agg["cells_stage1"] + agg["cells_stage2"] + agg["cells_stage3"]) / (tmax) / 1e9 * 1.0, 3),
            "phase_wall_s": {k: round(agg[k], 3) for k in ("t_stage1_s", "t_stage2_s", "t_stage3_s", "t_host_s", "t_total_s")},
            "kernel_ms": {KERNEL_NAMES[i]: round(kms[i], 2) for i in range(7)},
            "kernel_launches": {KERNEL_NAMES[i]: int(agg["kernel_launches"][i]) for i in range(7)},
            "counts": {k: int(agg[k]) for k in ("segments", "units", "candidates", "align_calls", "hazard_units", "rev_exact", "exact_replays", "tries_skipped",
                                                "align_word_reruns", "stage2_overflow_units")},
            "roofline": kernel_roofline(0),
            "roofline_stage3": kernel_roofline(2),
            "valu": {"kernel": KERNEL_NAMES[dom], "gcells_per_s": round(cells_dom / (kms[dom] * 1e-3) / 1e9, 2),
                     "ops_per_cell": ops_per_cell, "achieved_tops": round(cells_dom * ops_per_cell / (kms[dom] * 1e-3) / 1e12, 3),
                     "peak_tops": VALU_PK16_PEAK_TOPS,
                     "frac": round(cells_dom * ops_per_cell / (kms[dom] * 1e-3) / 1e12 / VALU_PK16_PEAK_TOPS, 4),
                     "issue_utilisation_measured": 0.99,
                     "issue_utilisation_source": "profiles/r01_sq_counters.txt: SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES of this kernel with one batch in "
                                                 "flight (rocprofv3 --pmc), i.e. one wave64 packed op per CU clock = the packed-16 issue limit",
                     "note": "VALU instructions per DP cell x executed cells / HIP-event time of this kernel family; "
                             "peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz: packed 16-bit VALU ops issue at 4 clocks per wave64 (measured, "
                             "profiles/r01_sq_counters.txt), half the plain VALU rate; the "
                             "packed 16-bit kernels process two cells per lane-instruction); kernel times include "
                             "sharing the GPU with the other batches in flight"},
        }
        ik = iso["kernel_ms"]
        out["isolated_kernels"] = {
            "what": f"untimed pass over the first {iso['segments']} segments with ONE batch in flight (kernels run alone)",
            "units": iso["units"], "align_calls": iso["align_calls"],
            "k_scan": {"ms": round(ik[0], 2), "gcells_per_s": round(iso["cells_stage2"] / (ik[0] * 1e-3) / 1e9, 1),
                       "valu_frac": round(iso["cells_stage2"] * 4.5 / (ik[0] * 1e-3) / 1e12 / VALU_PK16_PEAK_TOPS, 4),
                       "hbm_GBps": round((traffic_per_unit or 0) * iso["units"] / (ik[0] * 1e-3) / 1e9, 2)},
            "k_align_fwd": {"ms": round(ik[2], 2), "gcells_per_s": round(iso["align_calls"] and iso["cells_stage3"] / (ik[2] * 1e-3) / 1e9, 1),
                            "valu_frac": round(iso["cells_stage3"] * 5.5 / (ik[2] * 1e-3) / 1e12 / VALU_PK16_PEAK_TOPS, 4)},
            "k_striped_hazard_reruns_ms": round(ik[1], 2), "k_finish_lds_ms": round(ik[3], 2),
            "k_striped_exact_replays_ms": round(ik[5], 2), "k_finish_k_banded_global_ms": round(ik[6], 2),
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rna_path, m, args.cpu_sample_nt, 12345)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
