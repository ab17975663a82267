#!/usr/bin/env python3
"""bench.py -- headline benchmark of the triplex-scan hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Default workload (BASELINE.json configs[1]): H19 (2 812 nt, tests/golden/H19.fa) x synthetic uniform-random DNA
(splitmix64 seed 12345 + rank, `--dna-mb` million nt, default 50) with the reference's default parameters:
10 205 segments x 48 rule encodings = 489 840 work units per 50 Mb.  One STEP = one complete pass of the hot
path over that record: stage-1 max score, stage-2 column maxima, candidate picking, every window alignment
with traceback, triplex records -- i.e. everything fasim_scan() does.  The DNA record is uploaded once before
the timed region and stays resident in HBM.  With N ranks every rank scans its own record (weak scaling,
no data-path collective) and the records are gathered to rank 0 over RCCL inside the timed region.

Other workloads (second bench lines, parity-test configurations at scale):
    --dna genome|planted   chromosome-like record (tools/synth.py genome_like, upper-cased) / lncRNA pre-images planted
    --lncrnas K            BASELINE config 4: K synthetic 3 000-nt lncRNAs (synth.random_rna(3000, k)) scanned as ONE
                           batch over the resident record (fasim_scan_queries); records gathered per lncRNA
    --shard                ONE record of N x dna-mb million nt, sharded over the N ranks by contiguous segment ranges
                           (the layout of the north star), instead of one record per rank

Prints ONE JSON line (rank 0).  value = logical SW Gcells/s of the whole job =
sum_ranks(sum_queries(m) * sum(len(segment)) * 48) * K / max_rank(time) / 1e9   (SURVEY.md 8(d)).

Kernel-quality figures ("roofline", "valu", "isolated_kernels") come from an UNTIMED extra pass with ONE batch in
flight, so that HIP-event durations are exclusive: in the timed steps ten batches share the GPU and per-kernel
durations overlap (their sum exceeds the step time; they are reported as "kernel_ms_overlapped" for reference only).
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
# VALU issue peak for the instruction class the DP kernels are made of (packed 16-bit integer VOP3P ops: v_pk_add_i16,
# v_pk_max_i16/u16, v_pk_sub_u16 clamp; v_perm_b32 and the DPP moves cost the same).  tools/valu_issue_bench.hip, run on the
# same box (profiles/r02_valu_issue_bench.txt), measures a pure stream of such instructions at 1.81 ns per wave64 instruction
# and SIMD with four 256-thread workgroups per CU (k_scan's occupancy), and 1.75-1.93 ns at any other occupancy: the chip
# lowers its clock as more waves issue (about 1.8 GHz at this occupancy), i.e. the rate is power-limited.
#   measured sustained peak = 256 CUs x 4 SIMDs x 64 lanes / 1.81 ns = 36.2 T packed lane-instructions/s (2 cells each)
#   nominal figure used in round 1 = 4 clocks per instruction at the 2.4 GHz peak clock = 39.3 T (printed for continuity)
#   plain 32-bit VALU (v_add_u32 ...): 1.05 ns = 62.4 T measured; 2 clocks at 2.4 GHz = 78.6 T nominal (the guide's rate)
VALU_PK16_PEAK_TOPS = 256 * 4 * 64 / 1.81e-9 / 1e12
VALU_PK16_NOMINAL_TOPS = 256 * 4 * 64 / 4 * 2.4e9 / 1e12
VALU_PLAIN_PEAK_TOPS = 256 * 4 * 64 / 2 * 2.4e9 / 1e12
KERNEL_NAMES = ["k_scan (fused stage 1+2)", "k_striped<PRE|MAX1> (stage 1/2 hazard re-runs)", "k_align_fwd (stage 3 forward)",
                "k_finish_lds (reverse pass + traceback)", "k_encode/k_scan_post/k_hits/k_build_stream",
                "k_striped<ALIGN|REV> (stage 3 exact replays)", "k_finish/k_banded (global scratch)", "k_sim_forward (-F only)",
                "k_align_band (stage 3 forward on row bands)", "k_band_select"]
KERNEL_SHOWN = [0, 1, 2, 8, 9, 3, 4, 5, 6]
PMC_FILE = "r03_pmc_traffic.json"      # HBM counters of the current build (tools/pmc_traffic.py)
# packed VALU instructions per DP cell of the variants that are launched by default:
#   k_scan<RP>: perm, add, 3 x max, 3 x sat-sub, 2 x max = 10 per row pair = 5.0 per cell
#   k_align_fwd<RP,TAINT> and k_align_band<G>: the same + the row-key OR = 11 per row pair = 5.5 per cell
OPS_PER_CELL = {"k_scan": 5.0, "k_align_fwd": 5.5}


def host_cores_uncapped():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def host_cores():
    """cores this process may really use: scheduler affinity, capped by the cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


REF_EXTRA_ARGS = []       # e.g. ["-na", "1000"] (set from --nt-max)


def _ref_run(wd, name, dna, rna_path, m, cells_out):
    """one reference process on one slice (the reference is single-threaded); returns the Popen"""
    d = os.path.join(wd, name)
    os.makedirs(os.path.join(d, "out"))
    synth.write_fasta(os.path.join(d, "sample.fa"), f"syn|chrS|1-{len(dna)}", dna)
    shutil.copyfile(rna_path, os.path.join(d, "rna.fa"))
    cells_out.append(m * sum(min(5000, len(dna) - s) for s in range(0, len(dna), 4900)) * 48)
    ref = os.path.join(ROOT, "oracle", "_ref", "fasim_ref")
    return subprocess.Popen([ref, "-f1", "sample.fa", "-f2", "rna.fa", "-O", "out/"] + REF_EXTRA_ARGS, cwd=d, stdout=subprocess.DEVNULL,
                            stderr=subprocess.DEVNULL)


def cpu_baseline(rna_path, m, slices):
    """The reference's own SSE2 binary (oracle/_ref/fasim_ref, built from /root/reference by oracle/Makefile) on slices
    of the same kind of synthetic record: (a) ONE process on slices[0] = the 1-thread figure (the reference is
    single-threaded); (b) len(slices) processes side by side, one slice each = the all-core figure.
    Falls back to the oracle port when the reference binary did not travel."""
    sample_nt, cores = len(slices[0]), len(slices)

    def make_dna(_n, k):
        return slices[k]
    ref = os.path.join(ROOT, "oracle", "_ref", "fasim_ref")
    wd = tempfile.mkdtemp(prefix="fasim_cpu_")
    out = {}
    try:
        if os.access(ref, os.X_OK):
            cells = []
            t0 = time.perf_counter()
            assert _ref_run(wd, "one", make_dna(sample_nt, 0), rna_path, m, cells).wait() == 0
            dt1 = time.perf_counter() - t0
            out = {"value": round(cells[0] / dt1 / 1e9, 4), "unit": "Gcells/s", "cores": 1, "kind": "reference",
                   "sample": f"query x first {sample_nt} nt of the same synthetic record, default parameters, {dt1:.1f} s wall",
                   "mbp_per_s": round(sample_nt / dt1 / 1e6, 5)}
            if cores > 1:
                cells = []
                t0 = time.perf_counter()
                procs = [_ref_run(wd, f"p{k}", make_dna(sample_nt, k), rna_path, m, cells) for k in range(cores)]
                assert all(p.wait() == 0 for p in procs)
                dtn = time.perf_counter() - t0
                out["all_cores"] = {"value": round(sum(cells) / dtn / 1e9, 3), "unit": "Gcells/s", "cores": cores,
                                    "sample": f"{cores} reference processes side by side, one {sample_nt}-nt slice each, {dtn:.1f} s wall",
                                    "mbp_per_s": round(sample_nt * cores / dtn / 1e6, 5)}
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
            exe = os.path.join(ROOT, "oracle", "_build", "fasim_oracle")
            dna = make_dna(sample_nt, 0)
            synth.write_fasta(os.path.join(wd, "sample.fa"), f"syn|chrS|1-{sample_nt}", dna)
            shutil.copyfile(rna_path, os.path.join(wd, "rna.fa"))
            cells = m * sum(min(5000, sample_nt - s) for s in range(0, sample_nt, 4900)) * 48
            t0 = time.perf_counter()
            subprocess.run([exe, "tfosorted", "rna.fa", "sample.fa"], cwd=wd, check=True, stdout=subprocess.DEVNULL)
            dt1 = time.perf_counter() - t0
            out = {"value": round(cells / dt1 / 1e9, 4), "unit": "Gcells/s", "cores": 1, "kind": "port",
                   "sample": f"oracle port, query x first {sample_nt} nt, {dt1:.1f} s wall", "mbp_per_s": round(sample_nt / dt1 / 1e6, 5)}
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dna-mb", type=float, default=50.0, help="million nt of synthetic DNA per rank (default 50)")
    ap.add_argument("--dna", choices=["random", "genome", "planted"], default="random")
    ap.add_argument("--lncrnas", type=int, default=0, help="K > 0: K synthetic 3 kb lncRNAs as one batch (BASELINE config 4)")
    ap.add_argument("--rna", default="", help="comma-separated lncRNAs instead of H19: FASTA paths (relative to the repo root) or syn:N = "
                    "synth.random_rna(N, 515); several = one batch (BASELINE configs 3 and 5: tests/golden/MEG3.fa,tests/golden/MALAT1.fa,"
                    "tests/golden/NEAT1.fa / syn:10000 with --nt-max 1000)")
    ap.add_argument("--nt-max", type=int, default=0, help="-na of the reference (ntMax); 0 = default")
    ap.add_argument("--shard", action="store_true", help="one record of gpus x dna-mb, sharded by contiguous segment ranges")
    ap.add_argument("--cpu-sample-nt", type=int, default=250000)
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the all-core CPU baseline (0 = all cores this process may use, at most 64)")
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
    # Host threads sleep in stream synchronisation instead of polling (see fasim_engine_create: ten polling threads cost half
    # of a scan's CPU time).  The flag must be set before the device's context exists, i.e. before torch touches the device.
    sync_flags = "default (polling)"
    if os.environ.get("FASIM_BLOCKING_SYNC", "1") != "0":
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipSetDevice(local) == 0 and hip.hipSetDeviceFlags(0x4) == 0:      # hipDeviceScheduleBlockingSync
                sync_flags = "hipDeviceScheduleBlockingSync"
        except OSError:
            pass
    torch.cuda.set_device(local)
    xdev = "cuda" if args.backend == "nccl" else "cpu"      # where the exchanged record bytes live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    # N ranks share one node's host cores.  The host side of a batch (candidates, CIGAR -> record) is a burst that wants about
    # nine threads per batch in flight (ten batches per rank): with 16 threads per rank a 50 Mb step takes 3.05 s instead of 2.33 s,
    # 48 -> 2.49 s, 96 or 160 -> 2.33 s, measured on a box with a 16-core share (profiles/r02_ab_hostthreads.txt).  So every rank
    # keeps six threads per core of its share of the node (at most the single-rank default of 96) unless the user has set it.
    if world > 1 and "FASIM_HOST_THREADS" not in os.environ:
        os.environ["FASIM_HOST_THREADS"] = str(max(16, min(96, 6 * (host_cores_uncapped() // world))))
    mod = entry.load()
    eng = mod.Engine(local)
    if os.environ.get("FASIM_DEBUG_SYNCFLAG"):
        print("[bench] device schedule flags:", sync_flags, file=sys.stderr)
    rna_path = os.path.join(ROOT, "tests", "golden", "H19.fa")
    _, h19 = synth.read_fasta(rna_path)
    rnas = [synth.random_rna(3000, k + 1) for k in range(args.lncrnas)] if args.lncrnas > 0 else [h19]
    rna_names = [f"syn3k_{k + 1}" for k in range(args.lncrnas)] if args.lncrnas > 0 else ["H19"]
    if args.rna:
        rnas, rna_names = [], []
        for spec in args.rna.split(","):
            if spec.startswith("syn:"):
                rnas.append(synth.random_rna(int(spec[4:]), 515)); rna_names.append(f"syn{int(spec[4:])}")
            else:
                rnas.append(synth.read_fasta(spec if os.path.isabs(spec) else os.path.join(ROOT, spec))[1])
                rna_names.append(os.path.splitext(os.path.basename(spec))[0])
    multi = len(rnas) > 1 or args.lncrnas > 0

    def make_dna(n, k):
        """the k-th record of the workload (k = rank in the weak-scaling layout)"""
        if args.dna == "random":
            return mod.synth_dna(n, 12345 + k)
        if args.dna == "genome":
            return synth.genome_like(n, 12345 + k, soft_mask=False)
        return synth.planted_dna(n, 12345 + k, rnas[0])

    n = int(args.dna_mb * 1e6)
    p = mod.default_params(ntMax=args.nt_max) if args.nt_max > 0 else mod.default_params()
    if args.nt_max > 0:
        REF_EXTRA_ARGS.extend(["-na", str(args.nt_max)])
    if args.shard:
        dna = make_dna(n * world, 0)                       # every rank generates the same record and scans its shard of it
        seg_first, seg_count = mod.shard_segments(mod.segment_count(len(dna), p), rank, world)
    else:
        dna = make_dna(n, rank)
        seg_first, seg_count = 0, -1
    eng.load_dna(dna)                      # resident in HBM before the timed region
    if not multi:
        eng.set_query(rnas[0])

    def step():
        if multi:
            res = eng.scan_queries(rnas, None, p, seg_first, seg_count)
        else:
            res = [eng.scan(None, p, seg_first, seg_count)]
        # the path's one exchange step: gather every rank's records on rank 0 (RCCL over xGMI), per lncRNA
        nrec = 0
        for r in res:
            merged = mod.gather_results(r, dist, rank, world, xdev)
            nrec += merged.count if merged is not None else 0
        return res, nrec

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
        for r in res:
            st = r.stats
            if agg is None:
                agg = {k: (list(v) if isinstance(v, list) else v) for k, v in st.items()}
            else:
                for k, v in st.items():
                    if isinstance(v, list):
                        agg[k] = [a + b for a, b in zip(agg[k], v)]
                    else:
                        agg[k] += v
        del res
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
    cells = torch.tensor([float(agg["logical_cells"])], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cells, op=dist.ReduceOp.SUM)
    tmax = float(tmax.item())
    total_cells = float(cells.item())

    # untimed extra pass for the kernel-quality figures: ONE batch in flight, so every HIP-event duration is that of a
    # kernel that has the GPU to itself (exclusive); first lncRNA of the workload, first <= 1024 segments of the shard
    iso = None
    iso_all = []
    if rank == 0:
        eng.set_option("workers", 1)
        total_seg = mod.segment_count(len(dna), p)
        have = seg_count if seg_count >= 0 else total_seg
        for q, r in enumerate(rnas[:4]):
            # (about 14 M cells per unit for H19: a batch of 1024 segments; longer lncRNAs get proportionally fewer segments)
            nseg_iso = max(64, min(1024, int(1024 * 2812 / len(r))))
            eng.set_query(r)
            eng.set_option("seg_batch", nseg_iso)
            iso_all.append(eng.scan(None, p, seg_first, min(nseg_iso, have)).stats)
        iso = iso_all[0]
        eng.set_option("workers", 0)
        eng.set_option("seg_batch", 0)

    if rank == 0:
        m = len(rnas[0])
        units_per_step = agg["units"] / args.steps
        kms = agg["kernel_ms"]
        ik, il = iso["kernel_ms"], iso["kernel_launches"]
        n_avg = iso["cells_stage2"] / max(1, iso["units"]) / m        # mean segment length of the isolated pass
        fwd_cells = max(0, iso["cells_stage3"] - iso["band_cells"])   # executed by k_align_fwd (+ the few cells of the finish kernels)

        def kernel_roofline(idx):
            """HBM roofline of one of the two DP kernels from the isolated pass: algorithmic bytes per launch / exclusive
            HIP-event duration per launch (recorded on the stream the kernel is launched on)."""
            launches_k = max(1, il[idx])
            avg = ik[idx] / launches_k
            if idx == 0:   # k_scan, per unit: segment codes in (n) + u16 column maxima out (2n) + int8 profile in (5m)
                items = iso["units"] + iso["stage1_word_reruns"]
                per_item = 3 * n_avg + 5 * m
                alt = {"algorithmic_bytes_per_launch_3n_only": int(3 * n_avg * items / launches_k)}
            else:          # k_align_fwd, per full-height pass: L + 2 stream bytes in, 16-byte descriptor in, 24-byte result out
                # full-height passes = reverse passes (bounds for the band kernel) + the tries no band proved + 16-bit re-runs
                items = max(1, iso["rev_bound_passes"] + (iso["align_calls"] - iso["band_proven"]) + iso["align_word_reruns"])
                per_item = fwd_cells / m / items + 2 + 16 + 24
                alt = {"full_height_passes": int(items), "of_them_reverse_passes": int(iso["rev_bound_passes"])}
            alg = per_item * items / launches_k
            # measured HBM bytes per launch: bytes per item from the committed counter passes (two separate rocprofv3 --pmc runs folded
            # by tools/pmc_traffic.py; FETCH doubled for gfx950 as the guide prescribes) x the items of one launch
            traffic, tnote = None, "no counter file under profiles/"
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                per = pmc["k_scan" if idx == 0 else "k_align_fwd"]["hbm_bytes_per_item"]
                # (the counter file divides k_align_fwd's bytes by ALL window tries of its run, so the same denominator is used here)
                traffic = int(per * (items if idx == 0 else iso["align_calls"]) / launches_k)
                tnote = (f"profiles/{PMC_FILE}: {per:.0f} B per {'unit' if idx == 0 else 'window try'} (FETCH_SIZE x 2 + WRITE_SIZE of separate "
                         "rocprofv3 --pmc passes, tools/pmc_traffic.py) x the items of one launch of this run")
            except (OSError, KeyError, ValueError):
                pass
            d = {"bound": "hbm", "kernel": KERNEL_NAMES[idx], "achieved": round(alg / (avg * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic,
                 "traffic_note": tnote,
                 "avg_launch_ms": round(avg, 3), "launches": int(launches_k), "algorithmic_bytes_per_launch": int(alg),
                 "items_per_launch": int(items / launches_k), "source": "isolated pass: one batch in flight, exclusive HIP-event durations",
                 "note": "integer DP is VALU-bound by construction (SURVEY 8(d)); see valu"}
            d.update(alt)
            return d

        def valu(idx, key, cells_k):
            t = ik[idx] * 1e-3
            ops = OPS_PER_CELL[key]
            return {"kernel": KERNEL_NAMES[idx], "ms": round(ik[idx], 2), "gcells_per_s": round(cells_k / t / 1e9, 1),
                    "ops_per_cell": ops, "achieved_tops": round(cells_k * ops / t / 1e12, 3),
                    "peak_tops": round(VALU_PK16_PEAK_TOPS, 1), "frac": round(cells_k * ops / t / 1e12 / VALU_PK16_PEAK_TOPS, 4),
                    "frac_of_nominal_4clk_2p4ghz": round(cells_k * ops / t / 1e12 / VALU_PK16_NOMINAL_TOPS, 4),
                    "frac_of_plain_valu_nominal": round(cells_k * ops / t / 1e12 / VALU_PLAIN_PEAK_TOPS, 4)}

        executed = agg["cells_stage1"] + agg["cells_stage2"] + agg["cells_stage3"]
        qdesc = (f"{args.lncrnas} synthetic 3000-nt lncRNAs (one batch)" if args.lncrnas > 0 else
                 " + ".join(f"{n} ({len(r)} nt)" for n, r in zip(rna_names, rnas)) + (" as one batch" if len(rnas) > 1 else ""))
        workload = qdesc + (f", -na {args.nt_max}" if args.nt_max > 0 else "") + \
            f" x synthetic {args.dna} DNA, " + (f"ONE record of {args.dna_mb * world:g} Mb sharded over {world} GPU(s) by contiguous segment ranges"
                                                if args.shard else f"{args.dna_mb:g} Mb per GPU (seed 12345+rank)") + \
            ", default parameters, 48 rule encodings, records gathered to rank 0"
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
            "config": {"workload": workload, "units_per_step_per_gpu": int(units_per_step), "records_rank0": int(nrec),
                       "lncrnas": len(rnas), "sharded_record": bool(args.shard)},
            "mbp_per_s_per_lncrna": round(len(dna) * (1 if args.shard else world) * args.steps / tmax / 1e6, 3),
            "executed_gcells_per_s": round(executed / tmax / 1e9, 3),
            "executed_cells_per_step": {k: int(agg[k] / args.steps) for k in ("cells_stage1", "cells_stage2", "cells_stage3")},
            "phase_wall_s_summed_over_batches": {k: round(agg[k], 3) for k in ("t_stage1_s", "t_stage2_s", "t_stage3_s", "t_host_s")},
            "kernel_ms_overlapped": {KERNEL_NAMES[i]: round(kms[i], 2) for i in KERNEL_SHOWN},
            "kernel_launches": {KERNEL_NAMES[i]: int(agg["kernel_launches"][i]) for i in KERNEL_SHOWN},
            "counts": {k: int(agg[k]) for k in ("segments", "segments_skipped", "units", "candidates", "align_calls", "hazard_units", "rev_exact",
                                                "exact_replays", "tries_skipped", "align_word_reruns", "stage2_overflow_units", "stage1_word_reruns",
                                                "band_tries", "band_proven", "band_cells", "rev_bound_passes")},
            "per_unit": {"candidates": round(agg["candidates"] / max(1, agg["units"]), 2),
                         "hazard_units_pct": round(100.0 * agg["hazard_units"] / max(1, agg["units"]), 3),
                         "overflow_units_pct": round(100.0 * agg["stage2_overflow_units"] / max(1, agg["units"]), 3)},
            "roofline": kernel_roofline(0),
            "roofline_stage3": kernel_roofline(2),
            "valu": valu(0, "k_scan", iso["cells_stage2"]),
            "valu_stage3": valu(2, "k_align_fwd", fwd_cells),
            "valu_band": valu(8, "k_align_fwd", iso["band_cells"]) if ik[8] > 0 else None,
            "valu_note": "useful row work = packed 16-bit VALU instructions per DP cell x executed cells / EXCLUSIVE HIP-event time "
                         "(isolated pass); peak = the measured sustained issue rate of a pure stream of packed 16-bit ops on this chip at "
                         "the kernel's occupancy: 1024 SIMDs x 64 lanes / 1.81 ns (tools/valu_issue_bench.hip, "
                         "profiles/r02_valu_issue_bench.txt; power-limited, about 1.8 GHz); each packed lane-instruction processes two cells",
        }
        out["isolated_kernels"] = {
            "what": f"untimed pass over {iso['segments']} segments with ONE batch in flight (kernels run alone)",
            "units": iso["units"], "align_calls": iso["align_calls"], "hazard_units": iso["hazard_units"], "rev_exact": iso["rev_exact"],
            "ms": {KERNEL_NAMES[i]: round(ik[i], 2) for i in KERNEL_SHOWN},
            "band_tries": iso["band_tries"], "band_proven": iso["band_proven"], "band_cells": iso["band_cells"], "cells_stage3": iso["cells_stage3"],
            "dominant_kernel_ms_per_step_equivalent": round(ik[0] * units_per_step / max(1, iso["units"]), 1),
            # plan + checkpoint pass + all chunks of the stripe-faithful re-run (kernel family 1), scaled to one batch of 1024 segments
            "hazard_reruns_ms_per_49152_units": round(ik[1] * 49152 / max(1, iso["units"]), 2),
            "hazard_launches": int(il[1]),
        }
        if len(iso_all) > 1 or args.rna:
            # per lncRNA: the same one-batch pass (exclusive kernel times) and what it says about the query
            out["isolated_per_lncrna"] = [{
                "lncrna": rna_names[q], "m": len(rnas[q]), "segments": st["segments"], "units": st["units"], "t_total_s": round(st["t_total_s"], 4),
                "logical_gcells_per_s_one_batch_alone": round(st["logical_cells"] / max(1e-9, st["t_total_s"]) / 1e9, 1),
                "candidates_per_unit": round(st["candidates"] / max(1, st["units"]), 2), "align_calls": st["align_calls"],
                "band_tries": st["band_tries"], "band_proven": st["band_proven"], "rev_bound_passes": st["rev_bound_passes"],
                "rev_exact": st["rev_exact"], "hazard_units": st["hazard_units"], "stage2_overflow_units": st["stage2_overflow_units"],
                "cells_stage2": st["cells_stage2"], "cells_stage3": st["cells_stage3"],
                "ms": {KERNEL_NAMES[i]: round(st["kernel_ms"][i], 2) for i in KERNEL_SHOWN},
                "launches": {KERNEL_NAMES[i]: int(st["kernel_launches"][i]) for i in KERNEL_SHOWN},
            } for q, st in enumerate(iso_all)]
        if not args.no_cpu_baseline and world == 1:       # the CPU baseline is reported at N = 1 only
            cores = args.cpu_cores if args.cpu_cores > 0 else host_cores()
            tmp_rna = None
            cpu_rna = rna_path
            if args.lncrnas > 0 or args.rna:
                tmp_rna = tempfile.NamedTemporaryFile(suffix=".fa", delete=False)
                tmp_rna.write(b">" + rna_names[0].encode() + b"\n" + rnas[0] + b"\n")
                tmp_rna.close()
                cpu_rna = tmp_rna.name
            nn = args.cpu_sample_nt
            if args.dna == "random":      # slice 0 = the head of rank 0's record; the others: further records of the same stream family
                slices = [synth.random_dna(nn, 12345 + 1000 * k) for k in range(cores)]
            else:
                big = make_dna(nn * cores, 0)
                slices = [big[nn * k:nn * (k + 1)] for k in range(cores)]
            out["cpu_baseline"] = cpu_baseline(cpu_rna, m, slices)
            if tmp_rna:
                os.unlink(tmp_rna.name)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
