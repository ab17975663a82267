"""fasim-longtarget_amd: ctypes binding of libfasim_hip.so (include/fasim_hip.h).

The package directory name contains a hyphen (it is fixed by the project layout), so import it with

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("fasim_longtarget_amd", ".../fasim-longtarget_amd/__init__.py")
    mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod; spec.loader.exec_module(mod)

or use the `load()` helper in `__graft_entry__.py`.

Names mirror the reference's interface for this path: `calc_score_once` (stats.h:879), `ssw_pre_align`
(ssw.h:128), `ssw_align` (ssw.h:118), `pick_candidates` (Aligner::preAlign, ssw_cpp.cpp:427-572), `scan`
(= the body of LongTarget(), Fasim-LongTarget.cpp:379-598) and `tfosorted` (printResult(), :797-829).

There is no CPU fallback: if the shared library is missing or no HIP device is usable, construction of
`Engine` raises.
"""
from __future__ import annotations

import ctypes as C
import os

# the scan keeps ~10 batches in flight on separate HIP streams; give them more than the runtime's default of four
# hardware queues (must be in the environment before HIP initialises, i.e. before torch touches the GPU)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfasim_hip.so")


class FasimError(RuntimeError):
    pass


class Params(C.Structure):
    """struct fasim_params -- defaults of initEnv() (Fasim-LongTarget.cpp:284-303)."""
    _fields_ = [("rule", C.c_int32), ("cutLength", C.c_int32), ("strand", C.c_int32), ("overlapLength", C.c_int32),
                ("ntMin", C.c_int32), ("ntMax", C.c_int32), ("scoreMin", C.c_float), ("minIdentity", C.c_float),
                ("minStability", C.c_float), ("penaltyT", C.c_int32), ("penaltyC", C.c_int32), ("cDistance", C.c_int32),
                ("cLength", C.c_int32), ("classicSim", C.c_int32)]


class Alignment(C.Structure):
    _fields_ = [("sw_score", C.c_int32), ("ref_begin", C.c_int32), ("ref_end", C.c_int32), ("query_begin", C.c_int32),
                ("query_end", C.c_int32), ("cigar_len", C.c_int32), ("cigar", C.c_uint32 * 256)]

    def cigar_string(self) -> str:
        ops = "MIDNSHP=X"
        return "".join(f"{c >> 4}{'M' if (c & 15) > 8 else ops[c & 15]}" for c in self.cigar[: max(0, self.cigar_len)])


class Triplex(C.Structure):
    _fields_ = [("stari", C.c_int32), ("endi", C.c_int32), ("starj", C.c_int32), ("endj", C.c_int32), ("strand", C.c_int32),
                ("reverse", C.c_int32), ("rule", C.c_int32), ("nt", C.c_int32), ("score", C.c_float), ("identity", C.c_float),
                ("tri_score", C.c_float), ("seg", C.c_int32), ("enc", C.c_int32), ("genome_shift", C.c_int32),
                ("tfo_off", C.c_int64), ("tts_off", C.c_int64)]


class ScanStats(C.Structure):
    _fields_ = [("segments", C.c_int64), ("segments_skipped", C.c_int64), ("units", C.c_int64), ("candidates", C.c_int64),
                ("align_calls", C.c_int64), ("align_word_reruns", C.c_int64), ("stage2_overflow_units", C.c_int64),
                ("stage1_word_reruns", C.c_int64), ("logical_cells", C.c_int64), ("t_total_s", C.c_double),
                ("t_stage1_s", C.c_double), ("t_stage2_s", C.c_double), ("t_stage3_s", C.c_double), ("t_host_s", C.c_double),
                ("kernel_ms", C.c_double * 10), ("kernel_launches", C.c_int64 * 10), ("cells_stage1", C.c_int64),
                ("cells_stage2", C.c_int64), ("cells_stage3", C.c_int64), ("hazard_units", C.c_int64), ("rev_exact", C.c_int64),
                ("exact_replays", C.c_int64), ("tries_skipped", C.c_int64), ("band_tries", C.c_int64), ("band_proven", C.c_int64),
                ("band_cells", C.c_int64), ("rev_bound_passes", C.c_int64)]


class SimNode(C.Structure):
    """struct fasim_sim_node: one entry of SIM's node list (vertex, sim.h:47-58)."""
    _fields_ = [(k, C.c_int64) for k in ("score", "stari", "starj", "endi", "endj", "top", "bot", "left", "right")]

    def astuple(self):
        return tuple(getattr(self, k) for k, _ in self._fields_)


SIM_K = 50


class _Result(C.Structure):
    _fields_ = [("recs", C.POINTER(Triplex)), ("count", C.c_int64), ("pool", C.POINTER(C.c_char)), ("pool_len", C.c_int64),
                ("stats", ScanStats)]


EXPORTS = ["fasim_params_default", "fasim_engine_create", "fasim_engine_create_ex", "fasim_engine_destroy", "fasim_last_error", "fasim_set_option", "fasim_set_query",
           "fasim_calc_score_once", "fasim_ssw_pre_align", "fasim_ssw_colmax_word", "fasim_pick_candidates", "fasim_ssw_align", "fasim_pre_align_batch",
           "fasim_align_batch", "fasim_encode_unit", "fasim_sim_forward_batch", "fasim_sim_finish_unit", "fasim_scan", "fasim_scan_queries", "fasim_merge_results", "fasim_rebase_offsets", "fasim_load_dna", "fasim_result_free", "fasim_segment_count",
           "fasim_tfosorted", "fasim_tfoclass", "fasim_tfosorted_ex", "fasim_tfoclass_ex", "fasim_tail_outputs", "fasim_upper_case", "fasim_free",
           "fasim_synth_dna", "fasim_selfcheck_records",
           # the reference's own ssw.h ABI (include/ssw.h)
           "ssw_init", "init_destroy", "ssw_pre_align", "ssw_align", "align_destroy", "encoded_ops"]

_lib = None
_EMPTY_POOL = C.create_string_buffer(1)


def lib():
    """Load libfasim_hip.so (fails loudly when it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FasimError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    L.fasim_last_error.restype = C.c_char_p
    L.fasim_last_error.argtypes = [C.c_void_p]
    L.fasim_engine_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.fasim_engine_destroy.argtypes = [C.c_void_p]
    L.fasim_engine_destroy.restype = None
    L.fasim_params_default.argtypes = [C.POINTER(Params)]
    L.fasim_params_default.restype = None
    L.fasim_set_query.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.fasim_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.fasim_calc_score_once.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fasim_ssw_pre_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fasim_pick_candidates.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.c_int32, C.POINTER(C.c_int32)]
    L.fasim_ssw_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(Alignment)]
    L.fasim_pre_align_batch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.fasim_align_batch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32,
                                    C.POINTER(Alignment)]
    L.fasim_encode_unit.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p]
    L.fasim_sim_finish_unit.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(Params),
                                        C.POINTER(SimNode), C.c_int32, C.POINTER(C.POINTER(_Result))]
    L.fasim_sim_forward_batch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32,
                                          C.POINTER(C.c_int64), C.POINTER(SimNode), C.POINTER(C.c_int32)]
    L.fasim_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(Params),
                             C.POINTER(C.POINTER(_Result))]
    L.fasim_load_dna.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.fasim_scan_queries.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_int64,
                                     C.c_int64, C.c_int64, C.POINTER(Params), C.POINTER(C.POINTER(_Result))]
    L.fasim_merge_results.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                      C.c_int32, C.POINTER(C.POINTER(_Result))]
    L.fasim_rebase_offsets.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.fasim_ssw_colmax_word.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fasim_result_free.argtypes = [C.POINTER(_Result)]
    L.fasim_result_free.restype = None
    L.fasim_segment_count.argtypes = [C.c_int64, C.POINTER(Params)]
    L.fasim_segment_count.restype = C.c_int64
    L.fasim_tfosorted.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(Params),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.fasim_tfoclass.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p,
                                 C.POINTER(Params), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.fasim_tfosorted_ex.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(Params),
                                     C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.fasim_tfoclass_ex.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p,
                                    C.POINTER(Params), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.fasim_tail_outputs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p,
                                     C.POINTER(Params), C.c_int32] + [C.POINTER(C.c_void_p), C.POINTER(C.c_int64)] * 3
    L.fasim_upper_case.argtypes = [C.c_void_p, C.c_int64]
    L.fasim_upper_case.restype = None
    L.fasim_free.argtypes = [C.c_void_p]
    L.fasim_free.restype = None
    L.fasim_synth_dna.argtypes = [C.c_char_p, C.c_int64, C.c_uint64]
    L.fasim_synth_dna.restype = None
    _lib = L
    return L


def default_params(**kw) -> Params:
    p = Params()
    lib().fasim_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class ScanResult:
    """Records of one scan (or of one shard, or of a merge).  Either a view of a native `fasim_result` (what scan() and
    merge_results() return: nothing is copied) or two Python byte strings (`ScanResult(recs, pool, stats)`)."""

    def __init__(self, recs: bytes | None = b"", pool: bytes | None = b"", stats: dict | None = None, _native=None, _ext=None):
        self.stats = stats if stats is not None else {}
        self._native = _native            # POINTER(_Result) owned by this object
        self._ext = _ext                  # (recs address, count, pool address, pool_len, owner of that memory)
        lazy = _native is not None or _ext is not None
        self._recs = None if lazy else (recs or b"")
        self._pool = None if lazy else (pool or b"")

    def __del__(self):
        try:
            if self._native is not None:
                lib().fasim_result_free(self._native)
                self._native = None
        except Exception:
            pass

    @property
    def count(self) -> int:
        if self._native is not None or self._ext is not None:
            return self.pointers()[1]
        return len(self._recs) // C.sizeof(Triplex)

    @property
    def pool_len(self) -> int:
        return self.pointers()[3] if (self._native is not None or self._ext is not None) else len(self._pool)

    @property
    def recs(self) -> bytes:
        if self._recs is None:
            rp, cnt, _, _ = self.pointers()
            self._recs = C.string_at(rp, cnt * C.sizeof(Triplex)) if cnt else b""
        return self._recs

    @property
    def pool(self) -> bytes:
        if self._pool is None:
            _, _, pp, plen = self.pointers()
            self._pool = C.string_at(pp, plen) if plen else b""
        return self._pool

    def pointers(self):
        """(recs address, count, pool address, pool_len) for the C-ABI; valid while this object lives."""
        if self._ext is not None:
            return self._ext[:4]
        if self._native is not None:
            r = self._native.contents
            return C.cast(r.recs, C.c_void_p).value or 0, int(r.count), C.cast(r.pool, C.c_void_p).value or 0, int(r.pool_len)
        if getattr(self, "_keep", None) is None:      # created once: earlier pointers must stay valid
            self._keep = (C.create_string_buffer(self._recs, max(1, len(self._recs))), C.create_string_buffer(self._pool, max(1, len(self._pool))))
        return C.addressof(self._keep[0]), self.count, C.addressof(self._keep[1]), len(self._pool)

    def triplexes(self):
        recs, pool = self.recs, self.pool
        arr = (Triplex * self.count).from_buffer_copy(recs)
        out = []
        for t in arr:
            tfo = pool[t.tfo_off:pool.index(b"\0", t.tfo_off)]
            tts = pool[t.tts_off:pool.index(b"\0", t.tts_off)]
            out.append((t.stari, t.endi, t.starj, t.endj, t.strand, t.reverse, t.rule, t.nt, int(t.score),
                        C.c_uint32.from_buffer_copy(C.c_float(t.identity)).value,
                        C.c_uint32.from_buffer_copy(C.c_float(t.tri_score)).value, tfo, tts, t.seg, t.enc))
        return out


def _merge_pointers(ptrs):
    """ptrs: list of (recs address, count, pool address, pool_len) -> native-backed ScanResult (fasim_merge_results)."""
    L = lib()
    n = len(ptrs)
    recs = (C.c_void_p * n)(*[p[0] or None for p in ptrs])
    counts = (C.c_int64 * n)(*[p[1] for p in ptrs])
    pools = (C.c_void_p * n)(*[p[2] or None for p in ptrs])
    plens = (C.c_int64 * n)(*[p[3] for p in ptrs])
    out = C.POINTER(_Result)()
    rc = L.fasim_merge_results(recs, counts, pools, plens, n, C.byref(out))
    if rc != 0:
        raise FasimError(f"fasim_merge_results failed ({rc}): {L.fasim_last_error(None).decode()}")
    return ScanResult(stats={}, _native=out)


def merge_results(parts):
    """Concatenate shard results in rank order (shards are contiguous segment ranges, so this IS the canonical
    (segment, encoding, rank) order); pool offsets are rebased.  Native: fasim_merge_results, one pass per shard."""
    parts = list(parts)
    return _merge_pointers([r.pointers() for r in parts])      # `parts` keeps the source buffers alive during the call


def shard_segments(nseg: int, rank: int, world: int):
    """Contiguous block of segment indices for `rank` (blocks differ by at most one segment)."""
    base, rem = divmod(nseg, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


GATHER_CHUNK = 256 << 20      # bytes rank 0 stages on the device per transfer (two such buffers)


def _pieces(sizes, rsz, tot_r):
    """Transfer plan of gather_results: for every source rank k > 0 its records and its pool cut into pieces of at most
    GATHER_CHUNK bytes: (k, offset in the sender's staging buffer, offset in the merged buffer, length)."""
    out, roff, poff = [], 0, 0
    for k, (a, b) in enumerate(sizes):
        if k > 0:
            for src0, dst0, n in ((0, roff, a), (a, tot_r + poff, b)):
                o = 0
                while o < n:
                    ln = min(GATHER_CHUNK, n - o)
                    out.append((k, src0 + o, dst0 + o, ln))
                    o += ln
        roff += a
        poff += b
    return out


def gather_results(res: ScanResult, dist, rank: int, world: int, device: str = "cuda"):
    """The path's only exchange step: every rank contributes the records of its segment shard, rank 0 gets them
    merged in rank order (= canonical (segment, encoding, rank) order because shards are contiguous).
    One all_gather of (records bytes, pool bytes), then point-to-point transfers (RCCL send/recv over xGMI; gloo in
    the CPU tests) that put every shard's records and pool at their final positions of ONE pinned host buffer on rank 0;
    the pool offsets are then rebased in place (fasim_rebase_offsets).  No per-record work in Python and no second copy
    of the merged records.  On the GPU path rank 0 never stages more than 2 x GATHER_CHUNK bytes on its device: piece i + 1
    is received into one staging buffer while piece i leaves the other one for the host (at hg38 scale the merged records
    are 4-5 GB)."""
    import torch
    if world == 1:
        return res
    L = lib()
    rp, cnt, pp, plen = res.pointers()
    rsz = C.sizeof(Triplex)
    nr = cnt * rsz
    gathered = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([nr, plen], dtype=torch.int64, device=device))
    sizes = [(int(s[0]), int(s[1])) for s in gathered]
    tot_r = sum(a for a, _ in sizes)
    tot_p = sum(b for _, b in sizes)
    plan = _pieces(sizes, rsz, tot_r)
    gpu = device != "cpu"
    if rank != 0:
        stage = torch.empty(max(1, nr + plen), dtype=torch.uint8, pin_memory=gpu)
        if nr:
            C.memmove(stage.data_ptr(), rp, nr)
        if plen:
            C.memmove(stage.data_ptr() + nr, pp, plen)
        mine = stage.to(device, non_blocking=True) if gpu else stage
        for k, src, _dst, ln in plan:
            if k == rank:
                dist.send(mine[src:src + ln], 0)
        return None
    host = torch.empty(max(1, tot_r + tot_p), dtype=torch.uint8, pin_memory=gpu)       # [records of all shards][pools of all shards]
    base = host.data_ptr()
    if nr:
        C.memmove(base, rp, nr)                      # rank 0's own shard: host to host
    if plen:
        C.memmove(base + tot_r, pp, plen)
    if not gpu:
        for k, _src, dst, ln in plan:
            dist.recv(host[dst:dst + ln], k)
    elif plan:
        bufs = [torch.empty(min(GATHER_CHUNK, max(ln for _, _, _, ln in plan)), dtype=torch.uint8, device=device) for _ in range(2)]
        copy_stream = torch.cuda.Stream()
        freed = [None, None]                         # event: the buffer's previous piece has left for the host
        for i, (k, _src, dst, ln) in enumerate(plan):
            b = i & 1
            if freed[b] is not None:
                freed[b].synchronize()
            dist.recv(bufs[b][:ln], k)               # (RCCL recv is ordered on the current stream)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                host[dst:dst + ln].copy_(bufs[b][:ln], non_blocking=True)
                freed[b] = torch.cuda.Event()
                freed[b].record()
        copy_stream.synchronize()
    roff = poff = 0
    for a, b in sizes:
        if a and poff:
            L.fasim_rebase_offsets(base + roff, a // rsz, poff)
        roff += a
        poff += b
    return ScanResult(stats={}, _ext=(base, tot_r // rsz, base + tot_r, tot_p, host))


class Engine:
    """One engine per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device: int = 0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.fasim_engine_create(device, C.byref(h))
        if rc != 0:
            raise FasimError(f"fasim_engine_create({device}) failed ({rc}): {self._L.fasim_last_error(None).decode()}")
        self._h = h
        self.m = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fasim_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FasimError(f"libfasim_hip error {rc}: {self._L.fasim_last_error(self._h).decode()}")

    def set_option(self, key: str, value: int):
        self._check(self._L.fasim_set_option(self._h, key.encode(), value))

    def set_query(self, rna: bytes):
        self._check(self._L.fasim_set_query(self._h, rna, len(rna)))
        self.m = len(rna)

    # --- single-problem drop-ins ------------------------------------------------------------------
    def calc_score_once(self, target: bytes) -> int:
        s = C.c_int32()
        self._check(self._L.fasim_calc_score_once(self._h, target, len(target), C.byref(s)))
        return s.value

    def ssw_pre_align(self, target: bytes):
        out = (C.c_int32 * len(target))()
        self._check(self._L.fasim_ssw_pre_align(self._h, target, len(target), out))
        return list(out)

    def ssw_align(self, window: bytes) -> Alignment:
        a = Alignment()
        self._check(self._L.fasim_ssw_align(self._h, window, len(window), C.byref(a)))
        return a

    # --- batched ----------------------------------------------------------------------------------
    @staticmethod
    def _pack(seqs):
        blob = b"".join(seqs)
        n = len(seqs)
        offs = (C.c_int64 * n)()
        lens = (C.c_int32 * n)()
        o = 0
        for i, s in enumerate(seqs):
            offs[i] = o
            lens[i] = len(s)
            o += len(s)
        return blob, offs, lens

    def pre_align_batch(self, targets, want_cols=True, want_stage1=True):
        blob, offs, lens = self._pack(targets)
        cols = (C.c_int32 * len(blob))() if want_cols else None
        s1 = (C.c_int32 * len(targets))() if want_stage1 else None
        self._check(self._L.fasim_pre_align_batch(self._h, blob, offs, lens, len(targets), cols, s1))
        out_cols = None
        if want_cols:
            out_cols, o = [], 0
            for t in targets:
                out_cols.append(list(cols[o:o + len(t)]))
                o += len(t)
        return out_cols, (list(s1) if want_stage1 else None)

    def align_batch(self, windows):
        blob, offs, lens = self._pack(windows)
        out = (Alignment * len(windows))()
        self._check(self._L.fasim_align_batch(self._h, blob, offs, lens, len(windows), out))
        return list(out)

    def sim_forward(self, targets, min_scores):
        """Forward sweep of classic SIM (-F, sim.h:506-571) for a batch of targets: per target the node list it leaves, as
        tuples (score x10, stari, starj, endi, endj, top, bot, left, right) in list order."""
        blob, offs, lens = self._pack(targets)
        n = len(targets)
        mins = (C.c_int64 * n)(*min_scores)
        nodes = (SimNode * (n * SIM_K))()
        counts = (C.c_int32 * n)()
        self._check(self._L.fasim_sim_forward_batch(self._h, blob, offs, lens, n, mins, nodes, counts))
        return [[nodes[k * SIM_K + x].astuple() for x in range(counts[k])] for k in range(n)]

    # --- the LongTarget() body ----------------------------------------------------------------------
    def load_dna(self, dna: bytes):
        """Upload one DNA record and keep it resident in HBM; scan(None, ...) then scans it without H2D copies."""
        self._check(self._L.fasim_load_dna(self._h, dna, len(dna)))

    @staticmethod
    def _stats_dict(st) -> dict:
        stats = {}
        for k, _ in ScanStats._fields_:
            v = getattr(st, k)
            stats[k] = list(v) if hasattr(v, "__len__") else v
        return stats

    def scan(self, dna: bytes | None, params: Params | None = None, seg_first: int = 0, seg_count: int = -1) -> ScanResult:
        p = params or default_params()
        res = C.POINTER(_Result)()
        self._check(self._L.fasim_scan(self._h, dna, len(dna) if dna is not None else 0, seg_first, seg_count, C.byref(p),
                                       C.byref(res)))
        return ScanResult(stats=self._stats_dict(res.contents.stats), _native=res)

    def scan_queries(self, rnas, dna: bytes | None, params: Params | None = None, seg_first: int = 0, seg_count: int = -1):
        """Multi-lncRNA batch (fasim_scan_queries): one ScanResult per lncRNA, each identical to what scan() returns for
        that lncRNA alone; the DNA record stays resident and the lncRNAs share one work queue."""
        p = params or default_params()
        n = len(rnas)
        arr = (C.c_char_p * n)(*rnas)
        lens = (C.c_int32 * n)(*[len(r) for r in rnas])
        outs = (C.POINTER(_Result) * n)()
        self._check(self._L.fasim_scan_queries(self._h, arr, lens, n, dna, len(dna) if dna is not None else 0, seg_first,
                                               seg_count, C.byref(p), outs))
        self.m = len(rnas[-1])
        return [ScanResult(stats=self._stats_dict(outs[k].contents.stats), _native=outs[k]) for k in range(n)]


def sim_finish_unit(rna: bytes, seg: bytes, enc: int, dna_start: int, min_score: int, nodes, params: Params | None = None) -> "ScanResult":
    """Host half of the -F path for one unit (fasim_sim_finish_unit): `nodes` = 9-tuples of the forward sweep's node list."""
    L = lib()
    p = params or default_params()
    arr = (SimNode * max(1, len(nodes)))()
    for k, t in enumerate(nodes):
        for (name, _), v in zip(SimNode._fields_, t):
            setattr(arr[k], name, v)
    res = C.POINTER(_Result)()
    rc = L.fasim_sim_finish_unit(rna, len(rna), seg, len(seg), enc, dna_start, min_score, C.byref(p), arr, len(nodes), C.byref(res))
    if rc != 0:
        raise FasimError(f"fasim_sim_finish_unit failed ({rc}): {L.fasim_last_error(None).decode()}")
    return ScanResult(stats={}, _native=res)


def pick_candidates(cols, threshold):
    L = lib()
    n = len(cols)
    arr = (C.c_int32 * n)(*cols)
    s = (C.c_int32 * (n + 1))()
    p = (C.c_int32 * (n + 1))()
    k = C.c_int32()
    rc = L.fasim_pick_candidates(arr, n, threshold, s, p, n + 1, C.byref(k))
    if rc != 0:
        raise FasimError(L.fasim_last_error(None).decode())
    return [(s[i], p[i]) for i in range(k.value)]


def encode_unit(seg: bytes, enc: int):
    L = lib()
    t = C.create_string_buffer(len(seg) + 1)
    s = C.create_string_buffer(len(seg) + 1)
    rc = L.fasim_encode_unit(seg, len(seg), enc, t, s)
    if rc != 0:
        raise FasimError(L.fasim_last_error(None).decode())
    return t.raw[:len(seg)], s.raw[:len(seg)].rstrip(b"\0")


def segment_count(dna_len: int, params: Params | None = None) -> int:
    p = params or default_params()
    return lib().fasim_segment_count(dna_len, C.byref(p))


TAIL_CLAMP_CLUSTER = 1


def tfosorted(result: ScanResult, chr_name: str, start_genome: int, params: Params | None = None, flags: int = 0) -> bytes:
    """-TFOsorted bytes for the (merged) records; host-side tail of the path."""
    L = lib()
    p = params or default_params()
    text = C.c_void_p()
    n = C.c_int64()
    rp, cnt, pp, plen = result.pointers()
    rc = L.fasim_tfosorted_ex(rp or None, cnt, pp or C.addressof(_EMPTY_POOL), max(1, plen), chr_name.encode(),
                              start_genome, C.byref(p), flags, C.byref(text), C.byref(n))
    if rc != 0:
        raise FasimError(f"fasim_tfosorted failed ({rc}): {L.fasim_last_error(None).decode()}")
    try:
        return C.string_at(text, n.value)
    finally:
        L.fasim_free(text)


def tfoclass(result: ScanResult, level: int, chr_name: str, start_genome: int, dna_len: int, rna_name: str,
             params: Params | None = None, flags: int = 0) -> bytes:
    """-TFOclass<level>-<ds>-<lg> bedGraph bytes (print_cluster, Fasim-LongTarget.cpp:694) for the merged records."""
    L = lib()
    p = params or default_params()
    text = C.c_void_p()
    n = C.c_int64()
    rp, cnt, _, _ = result.pointers()
    rc = L.fasim_tfoclass_ex(rp or None, cnt, level, chr_name.encode(), start_genome, dna_len, rna_name.encode(), C.byref(p),
                             flags, C.byref(text), C.byref(n))
    if rc != 0:
        raise FasimError(f"fasim_tfoclass failed ({rc}): {L.fasim_last_error(None).decode()}")
    try:
        return C.string_at(text, n.value)
    finally:
        L.fasim_free(text)


def tail_outputs(result: ScanResult, chr_name: str, start_genome: int, dna_len: int, rna_name: str, params: Params | None = None,
                 flags: int = 0):
    """(-TFOsorted, -TFOclass1, -TFOclass2) bytes from one clustering of the records (fasim_tail_outputs)."""
    L = lib()
    p = params or default_params()
    texts = [C.c_void_p() for _ in range(3)]
    lens = [C.c_int64() for _ in range(3)]
    rp, cnt, pp, plen = result.pointers()
    args = []
    for t, n in zip(texts, lens):
        args += [C.byref(t), C.byref(n)]
    rc = L.fasim_tail_outputs(rp or None, cnt, pp or C.addressof(_EMPTY_POOL), max(1, plen), chr_name.encode(), start_genome, dna_len,
                              rna_name.encode(), C.byref(p), flags, *args)
    if rc != 0:
        raise FasimError(f"fasim_tail_outputs failed ({rc}): {L.fasim_last_error(None).decode()}")
    try:
        return tuple(C.string_at(t, n.value) for t, n in zip(texts, lens))
    finally:
        for t in texts:
            L.fasim_free(t)


def synth_dna(n: int, seed: int) -> bytes:
    buf = C.create_string_buffer(n)
    lib().fasim_synth_dna(buf, n, seed)
    return buf.raw[:n]


def parse_dna_header(header: str):
    """'>species|chr|start-end' -> (species, chr, start) like readDna() (Fasim-LongTarget.cpp:226-255)."""
    species = chro = start = ""
    tmp, j = "", 0
    for c in header.lstrip(">"):
        if c == "|" and j == 0:
            species, tmp, j = tmp, "", 1
        elif c == "|" and j == 1:
            chro, tmp, j = tmp, "", 2
        elif c == "-" and j == 2:
            start, tmp = tmp, ""
        else:
            tmp += c
    digits = ""
    for c in start.strip():
        if c.isdigit() or (c in "+-" and not digits):
            digits += c
        else:
            break
    try:
        st = int(digits)
    except ValueError:
        st = 0
    return species, chro, st
