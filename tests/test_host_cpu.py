"""CPU-only checks of the product's host side: the C-ABI library loads and exports every declared symbol,
and the pure-host entry points (no device needed) agree with the oracle / the reference fixtures."""
import os
import re

import pytest

import helpers
import synth
import __graft_entry__ as entry


def _mod():
    lib = os.path.join(entry.PKG_DIR, "libfasim_hip.so")
    if not os.path.exists(lib):
        entry.build()
    return entry.load()


def test_library_exports_every_declared_symbol():
    """Every function include/fasim_hip.h and include/ssw.h declare is exported by libfasim_hip.so (no compute calls)."""
    m = _mod()
    hdr = open(os.path.join(entry.ROOT, "include", "fasim_hip.h")).read()
    declared = set(re.findall(r"\b(fasim_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"fasim_engine", "fasim_params", "fasim_alignment", "fasim_triplex", "fasim_result", "fasim_scan_stats"}
    # the reference's own ABI (ssw.h:78-142): names as declared in include/ssw.h
    ssw = open(os.path.join(entry.ROOT, "include", "ssw.h")).read()
    ssw = re.sub(r"/\*.*?\*/", "", ssw, flags=re.S)
    ssw_declared = set(re.findall(r"\b([a-z_]+)\s*\(const s_profile\* prof|\b(ssw_init|init_destroy|align_destroy)\s*\(", ssw))
    ssw_declared = {a or b for a, b in ssw_declared} | {"encoded_ops"}
    assert ssw_declared == {"ssw_init", "init_destroy", "ssw_align", "ssw_pre_align", "align_destroy", "encoded_ops"}, ssw_declared
    L = m.lib()
    assert declared, "header parse failed"
    for name in sorted(declared | ssw_declared):
        assert hasattr(L, name), f"libfasim_hip.so does not export {name}"
    assert set(m.EXPORTS) == declared | ssw_declared


def test_ssw_h_layouts_match_the_reference_abi():
    """include/ssw.h must describe the same s_align layout as the reference's ssw.h:48-58 (x86-64: 40 bytes, cigar at 24)."""
    import ctypes as C

    class SAlign(C.Structure):
        _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                    ("read_begin1", C.c_int32), ("read_end1", C.c_int32), ("ref_end2", C.c_int32),
                    ("cigar", C.POINTER(C.c_uint32)), ("cigarLen", C.c_int32)]
    assert C.sizeof(SAlign) == 40 and SAlign.cigar.offset == 24 and SAlign.cigarLen.offset == 32
    # and the header spells the fields in that order
    ssw = open(os.path.join(entry.ROOT, "include", "ssw.h")).read()
    order = [ssw.index(f) for f in ("uint16_t score1", "uint16_t score2", "int32_t ref_begin1", "int32_t ref_end1",
                                    "int32_t read_begin1", "int32_t read_end1", "int32_t ref_end2", "uint32_t* cigar", "int32_t cigarLen")]
    assert order == sorted(order)


def test_integration_md_binding_is_the_compiled_text():
    """The LongTarget() binding printed in INTEGRATION.md section 2 is byte for byte what oracle/longtarget_binding.cpp
    compiles into oracle/_ref/fasim_ref_hipbind (reference driver + this binding; run on the GPU box by test_gpu_parity)."""
    md = open(os.path.join(entry.ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(entry.ROOT, "oracle", "longtarget_binding.cpp")).read()
    body = src.split("// >>> INTEGRATION.md section 2: begin\n")[1].split("// <<< INTEGRATION.md section 2: end")[0]
    assert "void LongTarget(struct para &paraList" in body
    assert "```cpp\n" + body + "```" in md, "INTEGRATION.md section 2 and oracle/longtarget_binding.cpp differ"
    if os.path.isdir("/root/reference"):
        assert os.access(os.path.join(entry.ROOT, "oracle", "_ref", "fasim_ref_hipbind"), os.X_OK), "run `make -C oracle ref`"


def test_classic_sim_host_half_against_reference_units(oracle_build, golden_dir):
    """Row f3, host half (csrc/host_sim.cpp: traceback, region re-sweeps, triplex records): fed with the node list of the
    forward sweep (here from the oracle; on the GPU box from k_sim_forward) it must reproduce the reference's own SIM() output
    for the unit (tests/golden/demoF.simscan.gz), identity / stability as float bits."""
    m = _mod()
    o = helpers.Oracle(oracle_build)
    _, rna = synth.read_fasta(os.path.join(golden_dir, "H19.fa"))
    _, dna = synth.read_fasta(os.path.join(golden_dir, "testDNA.fa"))
    units, cur = {}, None
    for line in helpers.gunzip(os.path.join(golden_dir, "demoF.simscan.gz")).decode().splitlines():
        f = line.split(" ")
        if f[0] == "V":
            cur = units.setdefault(int(f[2]), {"thr": int(f[9]), "x": []})
        elif f[0] == "X":
            cur["x"].append((int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[9]),
                             int(f[10], 16), int(f[11], 16), f[12].encode(), f[13].encode()))
    assert len(units) == 48
    p = m.default_params()
    checked = 0
    for enc in (0, 1, 12, 13, 26, 47):
        t, _ = o.encode_unit(dna, enc)
        thr = units[enc]["thr"]
        assert thr == int(o.stage1_max(rna, t) * 0.8)
        nodes = o.sim_forward_nodes(rna, t, thr)
        res = m.sim_finish_unit(rna, dna, enc, 0, thr, nodes, p)
        got = [x[:13] for x in res.triplexes()]
        assert got == units[enc]["x"], (enc, len(got), len(units[enc]["x"]))
        checked += len(got)
    assert checked > 50


def test_native_merge_rebases_offsets():
    """fasim_merge_results (host half of the exchange step): concatenation in the order given, pool offsets rebased."""
    import time
    import ctypes as C
    m = _mod()

    def part(n, tag):
        recs, pool = [], bytearray()
        for i in range(n):
            t = m.Triplex()
            t.stari, t.endi, t.seg, t.enc = i, i + 10, tag, i % 48
            t.tfo_off = len(pool)
            pool += f"TFO{tag}_{i}".encode() + b"\0"
            t.tts_off = len(pool)
            pool += f"TTS{tag}_{i}".encode() + b"\0"
            recs.append(bytes(t))
        return m.ScanResult(b"".join(recs), bytes(pool), {})

    parts = [part(5, 0), part(0, 1), part(7, 2)]
    merged = m.merge_results(parts)
    assert merged.count == 12 and merged.pool == b"".join(p.pool for p in parts)
    tr = merged.triplexes()
    assert [t[11] for t in tr] == [f"TFO0_{i}".encode() for i in range(5)] + [f"TFO2_{i}".encode() for i in range(7)]
    assert [t[12] for t in tr][5] == b"TTS2_0"
    assert m.merge_results([]).count == 0
    # size of the 8-GPU exchange of the 50 Mb bench (8 x 265 583 records): the gather puts the shards at their final
    # positions and only rebases the offsets in place (fasim_rebase_offsets) -- a few ms; the copying merge
    # (fasim_merge_results, used for local shards) is bound by the page faults of its fresh buffers
    big = part(1000, 3)
    n_rec, reps = 265583, 8
    recs = big.recs * (n_rec // 1000 + 1)
    pool = big.pool * (n_rec // 1000 + 1)
    shard = m.ScanResult(recs[:n_rec * C.sizeof(m.Triplex)], pool, {})
    shards = [shard] * reps
    ptrs = [s.pointers() for s in shards]
    t0 = time.perf_counter()
    out = m._merge_pointers(ptrs)
    dt_merge = time.perf_counter() - t0
    assert out.count == n_rec * reps
    data = recs[:n_rec * C.sizeof(m.Triplex)] * reps
    flat = C.create_string_buffer(data, len(data))
    t0 = time.perf_counter()
    for k in range(reps):
        assert m.lib().fasim_rebase_offsets(C.addressof(flat) + k * n_rec * C.sizeof(m.Triplex), n_rec, k * len(pool)) == 0
    dt_rebase = time.perf_counter() - t0
    assert flat.raw == out.recs, "in-place rebase and copying merge must agree"
    assert dt_rebase < 0.05, f"in-place rebase of 8 x 265583 records took {dt_rebase * 1e3:.1f} ms"
    print(f"8 x {n_rec} records: in-place rebase {dt_rebase * 1e3:.1f} ms, copying merge {dt_merge * 1e3:.1f} ms")


def test_no_device_fails_loudly():
    m = _mod()
    import torch
    if torch.cuda.is_available():
        return
    try:
        m.Engine(0)
    except m.FasimError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("Engine() must fail without a GPU")


def test_encodings_match_oracle(oracle_build):
    m = _mod()
    o = helpers.Oracle(oracle_build)
    seg = b"ACGTNACCGGTTNNAGCTTAGGCATCGX"[:27]
    for enc in range(48):
        assert m.encode_unit(seg, enc) == o.encode_unit(seg, enc), enc


def test_encodings_match_reference_fixture(golden_dir):
    # the demo scan fixture lists (strand, Para, rule) per canonical encoding index
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, "demo.scan.gz")))
    seen = {u["enc"]: (u["strand"], u["para"], u["rule"]) for u in units}
    assert len(seen) == 48
    for enc, (strand, para, rule) in seen.items():
        if enc < 12:
            assert (para, rule, strand) == (1, enc // 2 + 1, enc & 1)
        else:
            k = enc - 12
            assert (para, rule, strand) == (-1, k // 2 + 1, 0 if k & 1 else 1)


def test_pick_candidates_matches_reference_vectors(golden_dir):
    m = _mod()
    reqs = open(os.path.join(golden_dir, "batch.req")).read().splitlines()
    rsps = open(os.path.join(golden_dir, "batch.rsp")).read().splitlines()
    cols = None
    n = 0
    for rq, rs in zip(reqs, rsps):
        f, g = rq.split(" "), rs.split(" ")
        if f[0] == "P":
            cols = [int(x) for x in g[2:]]
        elif f[0] == "K":
            flat = [int(x) for x in g[2:]]
            assert m.pick_candidates(cols, int(f[3])) == list(zip(flat[0::2], flat[1::2]))
            n += 1
    assert n >= 50


def test_synth_generator_matches_python():
    m = _mod()
    for n, seed in ((1, 1), (31, 2), (32, 3), (33, 4), (1000, 12345)):
        assert m.synth_dna(n, seed) == synth.random_dna(n, seed)


def test_segment_count():
    m = _mod()
    for n in (1, 4899, 4900, 4901, 5000, 9800, 9801, 50_000_000):
        starts = list(range(0, n, 4900))
        assert m.segment_count(n) == len(starts)


@pytest.mark.parametrize("stem,hdr,n,lg,name", [
    ("demo_lg40", "hg19|chr11|2158478-2162843", 4366, 40, "H19"),
    ("demo_default", "hg19|chr11|2158478-2162843", 4366, 50, "H19"),
    ("planted40k", "syn|chrP|1001-41000", 40000, 40, "H19"),
    ("q2cat", "syn|chrQ|1-95000", 95000, 40, "H19"),
    ("malat1", "syn|chrL|1-10000", 10000, 40, None),
])
def test_tfoclass_writer_from_reference_rows(golden_dir, stem, hdr, n, lg, name):
    """Host tail only (no GPU): the rows of a reference -TFOsorted file, fed back through fasim_tfoclass, must give
    the reference's two bedGraph files byte for byte (print_cluster, Fasim-LongTarget.cpp:694-795)."""
    mod = _mod()
    if name is None:
        name = open(os.path.join(golden_dir, stem.upper() + ".fa")).readline().strip().replace(">", "")
    rows = open(os.path.join(golden_dir, stem + ".TFOsorted")).read().splitlines()[1:]
    arr = (mod.Triplex * len(rows))()
    for t, row in zip(arr, rows):
        f = row.split("\t")
        t.stari, t.endi, t.starj, t.endj = int(f[0]), int(f[1]), int(f[2]), int(f[3])
        t.tri_score, t.identity, t.rule, t.score, t.nt = float(f[8]), float(f[9]), int(f[11]), float(f[12]), int(f[13])
    res = mod.ScanResult(recs=bytes(arr), pool=b"", stats={})
    _, chro, start = mod.parse_dna_header(hdr)
    p = mod.default_params(cLength=lg)
    for level in (1, 2):
        got = mod.tfoclass(res, level, chro, start, n, name, p)
        assert got == open(os.path.join(golden_dir, f"{stem}.TFOclass{level}"), "rb").read()


def test_host_record_path_selfcheck():
    """The scan's numbers-only record conversion and its trivially copyable records through the dedup give what the
    reference-shaped string conversion and the full records give (float bits, order under the reference's comparators)."""
    import ctypes as C
    m = _mod()
    L = m.lib()
    L.fasim_selfcheck_records.restype = C.c_int
    L.fasim_selfcheck_records.argtypes = [C.c_uint64, C.c_int32, C.POINTER(C.c_int32)]
    for seed in (1, 2, 3):
        bad = C.c_int32(-1)
        assert L.fasim_selfcheck_records(seed, 20000, C.byref(bad)) == 0
        assert bad.value == 0, seed
