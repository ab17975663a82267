"""Parity of the HIP path (through the C-ABI) against the reference fixtures and the oracle.  GPU only."""
import os
import subprocess

import pytest

import helpers
import synth
import __graft_entry__ as entry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod():
    if not os.path.exists(os.path.join(entry.PKG_DIR, "libfasim_hip.so")):
        entry.build()
    return entry.load()


@pytest.fixture(scope="module")
def engine(mod):
    e = mod.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def h19(golden_dir):
    return synth.read_fasta(os.path.join(golden_dir, "H19.fa"))[1]


def test_raw_kernels_against_reference_vectors(mod, engine, golden_dir):
    """S = calc_score_once, P = ssw_pre_align, K = Aligner::preAlign's candidate list (fasim_pick_candidates on the
    column maxima of the HIP path), A = ssw_align on the reference's own outputs (batch.rsp)."""
    reqs = open(os.path.join(golden_dir, "batch.req")).read().splitlines()
    rsps = open(os.path.join(golden_dir, "batch.rsp")).read().splitlines()
    by_query = {}
    for rq, rs in zip(reqs, rsps):
        f = rq.split(" ")
        by_query.setdefault(f[1], []).append((f, rs.split(" ")))
    checked = {"S": 0, "P": 0, "K": 0, "A": 0}
    for q, items in by_query.items():
        engine.set_query(q.encode())
        targets = [f[2].encode() for f, _ in items if f[0] in "SPK"]
        if targets:
            uniq = sorted(set(targets))
            cols, s1 = engine.pre_align_batch(uniq)
            cmap = dict(zip(uniq, cols))
            smap = dict(zip(uniq, s1))
        wins = [f[2].encode() for f, _ in items if f[0] == "A"]
        amap = dict(zip(wins, engine.align_batch(wins))) if wins else {}
        for f, g in items:
            t = f[2].encode()
            if f[0] == "S":
                assert smap[t] == int(g[1]), ("S", q[:40], f[2][:40])
            elif f[0] == "P":
                assert cmap[t] == [int(x) for x in g[2:]], ("P", q[:40], f[2][:40])
            elif f[0] == "A":
                a = amap[t]
                exp = tuple(int(x) for x in g[1:6])
                if exp[0] == 0:
                    assert a.sw_score == 0
                else:
                    got = (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end)
                    assert got == exp and (a.cigar_string() or "*") == g[6], ("A", q[:40], f[2][:40], got, exp)
            elif f[0] == "K":
                got = mod.pick_candidates(cmap[t], int(f[3]))
                vals = [int(x) for x in g[2:]]
                assert len(got) == int(g[1]) and got == list(zip(vals[0::2], vals[1::2])), ("K", q[:40], f[2][:40])
            else:
                continue
            checked[f[0]] += 1
    assert min(checked.values()) >= 50, checked


def test_single_call_dropins(engine, h19, oracle_build):
    o = helpers.Oracle(oracle_build)
    engine.set_query(h19)
    for seed, n in ((1, 1), (2, 17), (3, 300), (4, 1234)):
        t = synth.random_dna(n, seed)
        assert engine.calc_score_once(t) == o.stage1_max(h19, t)
        assert engine.ssw_pre_align(t) == o.pre_align(h19, t)
    w = synth.random_dna(120, 5)
    a = engine.ssw_align(w)
    five, cig = o.align(h19, w)
    assert (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end) == five and a.cigar_string() == cig


def _expected_triplexes(units):
    exp = []
    for u in units:
        for x in u["triplexes"]:
            f = list(x)
            exp.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]),
                        int(f[9], 16), int(f[10], 16), f[11].encode(), f[12].encode(), u["seg"], u["enc"]))
    return exp


@pytest.mark.parametrize("scan_name,dna_name", [("demo.scan.gz", "testDNA.fa"), ("planted40k.scan.gz", "planted40k.fa")])
def test_scan_records_match_reference_fastsim(mod, engine, h19, golden_dir, scan_name, dna_name):
    """Every triplex the reference's fastSIM() emits, unit by unit, bit for bit (identity/stability as float bits)."""
    _, dna = synth.read_fasta(os.path.join(golden_dir, dna_name))
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, scan_name)))
    engine.set_query(h19)
    res = engine.scan(dna, mod.default_params(cLength=20))     # cLength == ntMin: LongTarget's tail filter == fastSIM's
    assert res.triplexes() == _expected_triplexes(units)
    assert res.stats["units"] == len(units)
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)


def _q2_units_that_matter(oracle_build, rna, dna):
    import ctypes
    o = helpers.Oracle(oracle_build)
    o.lib.fo_pre_align_noq2.restype = None
    o.lib.fo_pre_align_noq2.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    count = 0
    for seg in range(len(dna) // 5000):
        for enc in range(48):
            t, _ = o.encode_unit(dna[seg * 5000:(seg + 1) * 5000], enc)
            ref = o.pre_align(rna, t)
            buf = (ctypes.c_int * len(t))()
            o.lib.fo_pre_align_noq2(rna, len(rna), t, len(t), buf)
            if ref == list(buf):
                continue
            thr = int(o.stage1_max(rna, t) * 0.8)
            if [(c, v) for c, v in enumerate(ref) if v > thr] != [(c, v) for c, v in enumerate(buf) if v > thr]:
                count += 1
    return count


def test_q2_units_scan(mod, engine, h19, golden_dir, oracle_build):
    """Segments with units where the reference's signed lazy-F exit (Q2) changes results: the hazard detection of the
    systolic kernels must send them to the stripe-faithful kernels."""
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, "q2cat.fa"))
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, "q2cat.scan.gz")))
    engine.set_query(h19)
    res = engine.scan(dna, mod.default_params(cLength=20, overlapLength=0))
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    # every unit whose above-threshold columns really differ between the reference's signed exit and the textbook
    # recurrence must have been sent to the stripe-faithful kernel (differences below the threshold or behind the
    # overflow cut cannot matter, and the taint tracking of k_scan is allowed to ignore them)
    assert res.stats["hazard_units"] >= _q2_units_that_matter(oracle_build, h19, dna)
    p = mod.default_params(cLength=40, overlapLength=0)
    res = engine.scan(dna, p)
    _, chro, start = mod.parse_dna_header(hdr)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, "q2cat.TFOsorted"), "rb").read()


@pytest.mark.parametrize("fixture,qlen", [("q2cat.fa", 0), ("planted40k.fa", 0), ("planted40k.fa", 1200)])
def test_hazard_rerun_variants_agree(mod, h19, golden_dir, fixture, qlen):
    """The stripe-faithful re-run of the hazard units gives the same records however it is organised: one sequential run per
    unit (round 1), column chunks from checkpoints of a pass that starts at column 0, the same from the pipeline snapshots
    of the main scan (default), and with many small chunks (every chunk boundary is a place where a living Q2 deviation
    has to be carried on by the group that came from the left).  qlen 1200: a query whose stripes are too short for the
    row analysis of k_scan (coarse unit-level hazard test: the re-run starts at column 0)."""
    _, dna = synth.read_fasta(os.path.join(golden_dir, fixture))
    if qlen:
        h19 = h19[:qlen]
    p = mod.default_params(cLength=20, overlapLength=0)
    results = []
    for opts in ({"hazard_chunks": 0}, {"hazard_chunks": 1, "hazard_snapshots": 0}, {"hazard_chunks": 1, "hazard_snapshots": 1},
                 {"hazard_chunks": 1, "hazard_chunk_cols": 64, "hazard_hot_weight": 16}):
        e = mod.Engine(0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_query(h19)
        r = e.scan(dna, p)
        results.append((r.recs, r.pool, r.stats["hazard_units"], r.stats["candidates"]))
        e.close()
    assert results[0][2] > 0                      # the fixtures do hold hazard units
    for k in range(1, len(results)):
        assert results[k] == results[0], k


@pytest.mark.parametrize("query,dna_file", [("H19.fa", "planted40k.fa"), ("H19.fa", "q2cat.fa"), ("MALAT1.fa", "malat1_dna.fa"), ("NEAT1.fa", "neat1_dna.fa")])
def test_band_modes_agree(mod, golden_dir, query, dna_file):
    """Stage 3 gives the same records whether every window try sweeps the whole query (band 0, the reference's organisation), or
    row bands proven from k_scan's block maxima plus reverse passes (band 1, default), or the bands without reverse passes (band 2).
    MALAT1 / NEAT1: queries whose band profile is staged per row zone (3 and 8 zones)."""
    _, rna = synth.read_fasta(os.path.join(golden_dir, query))
    _, dna = synth.read_fasta(os.path.join(golden_dir, dna_file))
    p = mod.default_params(cLength=20)
    results, stats = [], []
    for band in (0, 1, 2):
        e = mod.Engine(0)
        e.set_option("band", band)
        e.set_query(rna)
        r = e.scan(dna, p)
        results.append((r.recs, r.pool, r.stats["candidates"], r.stats["align_calls"]))
        stats.append(r.stats)
        e.close()
    assert results[1] == results[0] and results[2] == results[0]
    assert stats[0]["band_tries"] == 0 and stats[0]["kernel_launches"][8] == 0
    assert stats[1]["band_proven"] > 0 and stats[1]["rev_bound_passes"] > 0 and stats[1]["kernel_launches"][8] > 0
    assert stats[2]["band_proven"] > 0 and stats[2]["rev_bound_passes"] == 0
    assert stats[1]["cells_stage3"] < stats[0]["cells_stage3"]


@pytest.mark.parametrize("name", ["meg3", "malat1", "neat1"])
def test_long_queries_scan(mod, engine, golden_dir, name):
    """MEG3 (1 582 nt, 1 query tile), MALAT1 (8 708 nt, 3 tiles) and NEAT1 (22 767 nt, 8 tiles): the systolic kernels run
    one launch per tile of 128 virtual lanes and hand the bottom row over through HBM."""
    _, rna = synth.read_fasta(os.path.join(golden_dir, name.upper() + ".fa"))
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, name + "_dna.fa"))
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, name + ".scan.gz")))
    engine.set_query(rna)
    res = engine.scan(dna, mod.default_params(cLength=20))
    assert res.stats["units"] == len(units)
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    assert res.stats["kernel_launches"][0] > 0, "the systolic scan kernel must have run"
    p = mod.default_params(cLength=40)
    res = engine.scan(dna, p)
    _, chro, start = mod.parse_dna_header(hdr)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, name + ".TFOsorted"), "rb").read()


def test_untidy_input_scan(mod, engine, h19, golden_dir):
    """N runs (a skipped all-N segment; units with N take the separate stage-1 pass), lower case, IUPAC codes."""
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, "messy.fa"))
    meta, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, "messy.scan.gz")))
    engine.set_query(h19)
    res = engine.scan(dna, mod.default_params(cLength=20))
    assert res.stats["segments_skipped"] == len(meta["skipped"]) > 0
    assert res.stats["units"] == len(units)
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    assert res.stats["stage1_word_reruns"] > 0, "units with N must have taken the separate stage-1 pass"
    p = mod.default_params(cLength=30)
    res = engine.scan(dna, p)
    _, chro, start = mod.parse_dna_header(hdr)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, "messy.TFOsorted"), "rb").read()


@pytest.mark.parametrize("name,systolic", [("h19_700", True), ("h19_100", False)])
def test_short_queries_scan(mod, golden_dir, name, systolic):
    """700 nt: systolic kernels with the coarse Q2 test (stripes shorter than 96 rows); 100 nt: fewer than 8 rows per
    stripe, everything runs on the stripe-faithful kernels."""
    _, rna = synth.read_fasta(os.path.join(golden_dir, name + ".fa"))
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, name + "_dna.fa"))
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, name + ".scan.gz")))
    e = mod.Engine(0)
    e.set_query(rna)
    res = e.scan(dna, mod.default_params(cLength=20))
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    if systolic:
        assert res.stats["kernel_launches"][0] > 0
    p = mod.default_params(cLength=25)
    res = e.scan(dna, p)
    _, chro, start = mod.parse_dna_header(hdr)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, name + ".TFOsorted"), "rb").read()
    for level in (1, 2):
        assert mod.tfoclass(res, level, chro, start, len(dna), name, p) == open(os.path.join(golden_dir, f"{name}.TFOclass{level}"), "rb").read()
    e.close()


def test_stage1_score_saturating_the_doubled_lanes(mod, golden_dir):
    """Stage-1 score 19 905 > 16 383: k_scan's doubled 16-bit lanes saturate, the unit must fall back to the 16-bit
    stripe-faithful kernel for its score and to the hazard path for its column maxima (same records as the reference)."""
    _, rna = synth.read_fasta(os.path.join(golden_dir, "satq.fa"))
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, "sat5k.fa"))
    _, units = helpers.parse_scan(helpers.gunzip(os.path.join(golden_dir, "sat5k.scan.gz")))
    assert max(u["stage1"] for u in units) > 16383
    e = mod.Engine(0)
    e.set_query(rna)
    res = e.scan(dna, mod.default_params(cLength=20))
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    assert res.stats["stage1_word_reruns"] >= 1 and res.stats["kernel_launches"][0] > 0
    t, _ = mod.encode_unit(dna, max(units, key=lambda u: u["stage1"])["enc"])
    assert e.calc_score_once(t) == max(u["stage1"] for u in units)
    p = mod.default_params(cLength=40)
    res = e.scan(dna, p)
    _, chro, start = mod.parse_dna_header(hdr)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, "sat5k.TFOsorted"), "rb").read()
    e.close()


def test_systolic_and_stripe_faithful_paths_agree(mod, h19, golden_dir, monkeypatch):
    """FASIM_SCAN_V1 / FASIM_ALIGN_V1 force the stripe-faithful kernels everywhere; records must be identical."""
    _, dna = synth.read_fasta(os.path.join(golden_dir, "planted40k.fa"))
    p = mod.default_params(cLength=20)
    e2 = mod.Engine(0)
    e2.set_query(h19)
    fast = e2.scan(dna, p)
    e2.close()
    monkeypatch.setenv("FASIM_SCAN_V1", "1")
    monkeypatch.setenv("FASIM_ALIGN_V1", "1")
    e1 = mod.Engine(0)
    e1.set_query(h19)
    slow = e1.scan(dna, p)
    e1.close()
    assert fast.recs == slow.recs and fast.pool == slow.pool
    assert fast.stats["kernel_launches"][0] > 0 and slow.stats["kernel_launches"][0] == 0


def test_scheduling_options_agree(mod, h19, golden_dir):
    """Batches in flight, segments per batch, the gate of the VALU-bound kernels, host threads, the taper and the NUMA pinning only
    shape the schedule: every setting gives the records of the default one."""
    _, dna = synth.read_fasta(os.path.join(golden_dir, "planted40k.fa"))
    p = mod.default_params(cLength=20)
    results = []
    for opts in ({}, {"workers": 1}, {"workers": 3, "seg_batch": 2}, {"workers": 16, "seg_batch": 1, "heavy_gate": 0},
                 {"seg_batch": 4, "heavy_gate": 1, "host_threads": 2, "taper": 50}, {"numa_affinity": 0, "host_threads": 64}):
        e = mod.Engine(0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_query(h19)
        r = e.scan(dna, p)
        results.append((r.recs, r.pool, r.stats["units"], r.stats["candidates"]))
        e.close()
    for k in range(1, len(results)):
        assert results[k] == results[0], k


@pytest.mark.parametrize("name,dna_name,kw", [
    ("demo_lg40.TFOsorted", "testDNA.fa", dict(cLength=40)),
    ("demo_default.TFOsorted", "testDNA.fa", dict()),
    ("demo_t1_r3.TFOsorted", "testDNA.fa", dict(cLength=30, strand=1, rule=3)),
    ("planted40k.TFOsorted", "planted40k.fa", dict(cLength=40)),
])
def test_tfosorted_identical(mod, engine, h19, golden_dir, name, dna_name, kw):
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, dna_name))
    _, chro, start = mod.parse_dna_header(hdr)
    engine.set_query(h19)
    p = mod.default_params(**kw)
    res = engine.scan(dna, p)
    assert mod.tfosorted(res, chro, start, p) == open(os.path.join(golden_dir, name), "rb").read()
    for level in (1, 2):
        gold = open(os.path.join(golden_dir, name.replace(".TFOsorted", f".TFOclass{level}")), "rb").read()
        assert mod.tfoclass(res, level, chro, start, len(dna), "H19", p) == gold


def test_rnd30k_tfosorted_and_sharding(mod, engine, h19, golden_dir):
    dna = synth.random_dna(30000, 12345)
    engine.set_query(h19)
    p = mod.default_params(cLength=40)
    whole = engine.scan(dna, p)
    gold = open(os.path.join(golden_dir, "rnd30k.TFOsorted"), "rb").read()
    assert mod.tfosorted(whole, "chrS", 1, p) == gold
    # the systolic path must carry the work: stripe-faithful re-runs / replays are the rare exception on random DNA
    # (a broken fast path is invisible in the records, because everything it cannot decide is replayed exactly)
    st = whole.stats
    assert st["kernel_launches"][0] > 0 and st["kernel_launches"][2] > 0
    assert st["hazard_units"] * 20 <= st["units"], st
    assert st["exact_replays"] * 50 <= st["candidates"], st
    # the same scan as 3 contiguous shards (what 3 ranks would do), merged in rank order
    nseg = mod.segment_count(len(dna), p)
    cuts = [0, nseg // 3, 2 * nseg // 3, nseg]
    parts = [engine.scan(dna, p, cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)]
    merged = mod.merge_results(parts)
    assert merged.recs == whole.recs and merged.pool == whole.pool
    assert mod.tfosorted(merged, "chrS", 1, p) == gold


def test_cli_driver_writes_identical_file(golden_dir, tmp_path):
    exe = os.path.join(entry.PKG_DIR, "fasim")
    for f in ("H19.fa", "testDNA.fa"):
        (tmp_path / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
    (tmp_path / "out").mkdir()
    subprocess.run([exe, "-f1", "testDNA.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL)
    got = (tmp_path / "out" / "hg19-H19-testDNA-TFOsorted").read_bytes()
    assert got == open(os.path.join(golden_dir, "demo_lg40.TFOsorted"), "rb").read()
    for level in (1, 2):
        got = (tmp_path / "out" / f"hg19-H19-testDNA-TFOclass{level}-15-40").read_bytes()
        assert got == open(os.path.join(golden_dir, f"demo_lg40.TFOclass{level}"), "rb").read()


def test_cli_all_records_streams_a_multi_record_file(golden_dir, tmp_path):
    """--all-records: every record of a multi-record DNA file is scanned on its own (one in memory at a time) and gets
    its own output files, each identical to what the reference writes for that record alone."""
    exe = os.path.join(entry.PKG_DIR, "fasim")
    both = open(os.path.join(golden_dir, "testDNA.fa"), "rb").read().rstrip(b"\n") + b"\n" + \
        open(os.path.join(golden_dir, "planted40k.fa"), "rb").read()
    (tmp_path / "two.fa").write_bytes(both)
    (tmp_path / "H19.fa").write_bytes(open(os.path.join(golden_dir, "H19.fa"), "rb").read())
    (tmp_path / "out").mkdir()
    subprocess.run([exe, "-f1", "two.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40", "--all-records"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL)
    for stem, gold in (("hg19-H19-two.chr11", "demo_lg40"), ("syn-H19-two.chrP", "planted40k")):
        assert (tmp_path / "out" / f"{stem}-TFOsorted").read_bytes() == open(os.path.join(golden_dir, gold + ".TFOsorted"), "rb").read()
        for level in (1, 2):
            got = (tmp_path / "out" / f"{stem}-TFOclass{level}-15-40").read_bytes()
            assert got == open(os.path.join(golden_dir, f"{gold}.TFOclass{level}"), "rb").read()
    # without the flag only the first record is scanned (and named as the reference names it)
    (tmp_path / "out1").mkdir()
    r = subprocess.run([exe, "-f1", "two.fa", "-f2", "H19.fa", "-O", "out1/", "-lg", "40"], cwd=tmp_path, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert b"more than one record" in r.stderr
    assert (tmp_path / "out1" / "hg19-H19-two-TFOsorted").read_bytes() == open(os.path.join(golden_dir, "demo_lg40.TFOsorted"), "rb").read()


def test_live_reference_probe_random_vectors(engine, oracle_build):
    """If the compiled reference travelled with the repo (oracle/_ref), compare fresh random vectors too."""
    if not helpers.have_ref_probe():
        pytest.skip("oracle/_ref/ref_probe not present")
    rng = synth._Rng(424242)
    q = synth.random_rna(900 + rng.below(50), 7).decode()
    engine.set_query(q.encode())
    reqs, targets, wins = [], [], []
    for k in range(40):
        t = synth.planted_dna(300 + rng.below(900), 1000 + k, q.encode(), every=150, max_len=120).decode()
        targets.append(t.encode())
        reqs += [f"S {q} {t}", f"P {q} {t}"]
        w = t[:60 + rng.below(130)]
        wins.append(w.encode())
        reqs.append(f"A {q} {w}")
    rsp = helpers.ref_batch(reqs)
    cols, s1 = engine.pre_align_batch(targets)
    als = engine.align_batch(wins)
    for k in range(40):
        assert s1[k] == int(rsp[3 * k].split(" ")[1])
        assert cols[k] == [int(x) for x in rsp[3 * k + 1].split(" ")[2:]]
        g = rsp[3 * k + 2].split(" ")
        a = als[k]
        if int(g[1]) == 0:
            assert a.sw_score == 0
        else:
            assert (a.sw_score, a.ref_begin, a.ref_end, a.query_begin, a.query_end) == tuple(int(x) for x in g[1:6])
            assert (a.cigar_string() or "*") == g[6]


def test_live_reference_10kb_query_ntmax(mod, engine, tmp_path):
    """BASELINE config 5 in miniature: a 10 kb synthetic lncRNA (4 query tiles), -na 1000, planted DNA so that byte
    overflows and 16-bit re-runs occur; compared with the compiled reference when it travelled with the repo."""
    if not helpers.have_ref_probe():
        pytest.skip("oracle/_ref/ref_probe not present")
    rna = synth.random_rna(10000, 515)
    dna = synth.planted_dna(12000, 516, rna, every=500, max_len=180, mut_pct=6)
    (tmp_path / "rna.fa").write_bytes(b">syn10k\n" + rna + b"\n")
    (tmp_path / "dna.fa").write_bytes(b">syn|chrT|1-12000\n" + dna + b"\n")
    out = subprocess.run([helpers.REF_PROBE, "scan", "rna.fa", "dna.fa", "-detail", "0", "-na", "1000"], cwd=tmp_path, check=True,
                         stdout=subprocess.PIPE).stdout
    _, units = helpers.parse_scan(out)
    engine.set_query(rna)
    res = engine.scan(dna, mod.default_params(cLength=20, ntMax=1000))
    assert res.stats["candidates"] == sum(u["ncand"] for u in units)
    assert res.triplexes() == _expected_triplexes(units)
    assert sum(u["stage1"] >= 251 for u in units) > 0, "the case must exercise byte overflow"


# ---- the reference's own ssw.h ABI (include/ssw.h) ---------------------------------------------------------------
SHIM_PROBE = os.path.join(entry.ROOT, "oracle", "_ref", "shim_probe")


def test_ssw_h_shim_runs_reference_wrapper(golden_dir):
    """oracle/_ref/shim_probe = the probe + the reference's UNCHANGED C++ wrapper (ssw_cpp.cpp), linked against the
    ssw_init / ssw_pre_align / ssw_align / init_destroy / align_destroy of libfasim_hip.so in place of the reference's
    sswNew.cpp (oracle/Makefile).  On the reference's own request file it must give the reference's own answers."""
    if not os.access(SHIM_PROBE, os.X_OK):
        pytest.skip("oracle/_ref/shim_probe not present (built by `make -C oracle ref` where /root/reference exists)")
    req = open(os.path.join(golden_dir, "batch.req"), "rb").read()
    out = subprocess.run([SHIM_PROBE, "batch"], input=req, check=True, stdout=subprocess.PIPE).stdout
    got = out.decode().splitlines()
    exp = open(os.path.join(golden_dir, "batch.rsp")).read().splitlines()
    assert len(got) == len(exp)
    kinds = {}
    for g, e in zip(got, exp):
        if e.startswith("A 0 "):        # nothing aligned: the reference's coordinates there come from reading ref[-1]
            assert g.split(" ")[1] == "0"
        else:
            assert g == e, (g[:120], e[:120])
        kinds[e[0]] = kinds.get(e[0], 0) + 1
    assert kinds["P"] >= 50 and kinds["K"] >= 50 and kinds["A"] >= 50, kinds


def test_ssw_h_abi_through_ctypes(mod, oracle_build):
    """The five symbols called directly with the reference's calling convention (ssw_cpp.cpp:394-440, 605-640):
    calloc'd profile borrowing read/mat, calloc'd int[refLen] freed by the caller, calloc'd s_align freed with
    align_destroy, NULL + stderr for a scoring the engine does not implement."""
    import ctypes as C
    L = mod.lib()

    class SAlign(C.Structure):
        _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                    ("read_begin1", C.c_int32), ("read_end1", C.c_int32), ("ref_end2", C.c_int32),
                    ("cigar", C.POINTER(C.c_uint32)), ("cigarLen", C.c_int32)]
    L.ssw_init.restype = C.c_void_p
    L.ssw_init.argtypes = [C.POINTER(C.c_int8), C.c_int32, C.POINTER(C.c_int8), C.c_int32, C.c_int8]
    L.init_destroy.argtypes = [C.c_void_p]
    L.init_destroy.restype = None
    L.ssw_pre_align.restype = C.POINTER(C.c_int)
    L.ssw_pre_align.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32, C.c_int]
    L.ssw_align.restype = C.POINTER(SAlign)
    L.ssw_align.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]
    L.align_destroy.argtypes = [C.POINTER(SAlign)]
    L.align_destroy.restype = None
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    o = helpers.Oracle(oracle_build)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    q = synth.random_rna(777, 21)
    t = synth.planted_dna(900, 22, q, every=200, max_len=110)
    w = t[:150]

    def codes(b):
        return (C.c_int8 * len(b))(*[code.get(c, 4) for c in b])
    mat = (C.c_int8 * 25)(*[(5 if (i == j and i < 4) else -4) for i in range(5) for j in range(5)])
    cq = codes(q)
    prof = L.ssw_init(cq, len(q), mat, 5, 2)
    assert prof
    cols = L.ssw_pre_align(prof, codes(t), len(t), 16, 4, 0x0f, 0, 32767, 15, 0)
    assert cols and [cols[i] for i in range(len(t))] == o.pre_align(q, t)
    libc.free(C.cast(cols, C.c_void_p))                         # the caller frees it (ssw_cpp.cpp:440)
    a = L.ssw_align(prof, codes(w), len(w), 16, 4, 0x0f, 0, 32767, 15)
    assert a
    five, cig = o.align(q, w)
    r = a.contents
    assert (r.score1, r.ref_begin1, r.ref_end1, r.read_begin1, r.read_end1) == five
    assert helpers.cigar_to_string([r.cigar[i] for i in range(r.cigarLen)]) == cig
    # sub-optimal score (sswNew.cpp:641-665): best column maximum outside +-maskLen of ref_end1
    pc = o.pre_align(q, w)
    best, at = 0, 0
    for i in list(range(0, max(0, r.ref_end1 - 15))) + list(range(min(len(w), r.ref_end1 + 15) + 1, len(w))):
        if pc[i] > best:
            best, at = pc[i], i
    assert (r.score2, r.ref_end2) == (best, at)
    L.align_destroy(a)
    # flag 0: scores and end positions only
    a = L.ssw_align(prof, codes(w), len(w), 16, 4, 0, 0, 32767, 15)
    assert a and a.contents.ref_begin1 == -1 and a.contents.cigarLen == 0 and not a.contents.cigar
    L.align_destroy(a)
    # a scoring the engine does not implement: NULL (+ message on stderr), the reference's error convention
    assert not L.ssw_align(prof, codes(w), len(w), 12, 4, 0x0f, 0, 32767, 15)
    assert not L.ssw_pre_align(prof, codes(t), len(t), 16, 2, 0x0f, 0, 32767, 15, 0)
    L.init_destroy(prof)


def test_reference_driver_with_longtarget_binding(golden_dir, tmp_path):
    """oracle/_ref/fasim_ref_hipbind = the reference's UNCHANGED driver (main, readDna, printResult, cluster_triplex,
    print_cluster) linked with the LongTarget() binding of INTEGRATION.md section 2 instead of its own: the reference's
    front and back end on top of libfasim_hip.so write the reference's files."""
    exe = os.path.join(entry.ROOT, "oracle", "_ref", "fasim_ref_hipbind")
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/fasim_ref_hipbind not present (built by `make -C oracle ref` where /root/reference exists)")
    for name, dna, extra in (("demo_lg40", "testDNA.fa", ["-lg", "40"]), ("planted40k", "planted40k.fa", ["-lg", "40"]),
                             ("demo_t1_r3", "testDNA.fa", ["-lg", "30", "-t", "1", "-r", "3"])):
        wd = tmp_path / name
        (wd / "out").mkdir(parents=True)
        for f in ("H19.fa", dna):
            (wd / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
        subprocess.run([exe, "-f1", dna, "-f2", "H19.fa", "-O", "out/"] + extra, cwd=wd, check=True, stdout=subprocess.DEVNULL)
        outs = sorted(os.listdir(wd / "out"))
        assert len(outs) == 3, outs
        for f in outs:
            data = (wd / "out" / f).read_bytes()
            if f.endswith("-TFOsorted"):
                assert data == open(os.path.join(golden_dir, name + ".TFOsorted"), "rb").read(), f
            else:
                level = f.split("-TFOclass")[1].split("-")[0]
                assert data == open(os.path.join(golden_dir, f"{name}.TFOclass{level}"), "rb").read(), f


# ---- BASELINE config 4: many lncRNAs x one genome record, segments sharded -----------------------------------------
def _gz(golden_dir, name):
    return helpers.gunzip(os.path.join(golden_dir, name))


def _cfg4_inputs():
    dna_masked = synth.genome_like(240000, 4004)
    rnas = [synth.random_rna(3000, k) for k in (1, 2, 3, 4)]
    return dna_masked, rnas


def test_config4_multi_lncrna_sharded_scan(mod, golden_dir):
    """4 synthetic 3 kb lncRNAs x a 240 kb genome-like record (N gaps incl. whole segments, soft-masking upper-cased as
    `--upper` does, purine tracts), scanned as 3 contiguous segment shards with fasim_scan_queries over the RESIDENT
    record and merged natively: byte-identical to the reference CLI run once per lncRNA (tests/golden/cfg4_q*.gz)."""
    dna_masked, rnas = _cfg4_inputs()
    dna = dna_masked.upper()
    assert dna != dna_masked and b"N" * 5000 in dna
    p = mod.default_params(cLength=40)
    e = mod.Engine(0)
    e.load_dna(dna)
    nseg = mod.segment_count(len(dna), p)
    cuts = [0, nseg // 3, 2 * nseg // 3, nseg]
    shards = [e.scan_queries(rnas, None, p, cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)]
    whole = e.scan_queries(rnas, None, p)
    for q in range(4):
        merged = mod.merge_results([shards[i][q] for i in range(3)])
        assert merged.recs == whole[q].recs and merged.pool == whole[q].pool
        assert mod.tfosorted(merged, "chrG", 1, p) == _gz(golden_dir, f"cfg4_q{q + 1}.TFOsorted.gz")
        for level in (1, 2):
            assert mod.tfoclass(merged, level, "chrG", 1, len(dna), f"syn3k_{q + 1}", p) == _gz(golden_dir, f"cfg4_q{q + 1}.TFOclass{level}.gz")
        assert whole[q].stats["segments_skipped"] > 0 and whole[q].stats["units"] > 0
    # the batch returns, per lncRNA, exactly what a scan of that lncRNA alone returns (here: host buffer, streamed ingest)
    e.set_query(rnas[2])
    alone = e.scan(dna, p)
    assert alone.recs == whole[2].recs and alone.pool == whole[2].pool
    assert sum(alone.stats[k] for k in ("units", "candidates")) == sum(whole[2].stats[k] for k in ("units", "candidates"))
    e.close()


def test_cli_multi_lncrna_devices_upper(golden_dir, tmp_path):
    """The driver end to end on config 4 in miniature: soft-masked DNA + --upper, four lncRNAs in one -f2 file, the
    record cut into three device shards (--devices 0,0,0: three engines on the one GPU of the test box): the three
    output files of every lncRNA equal the reference's."""
    exe = os.path.join(entry.PKG_DIR, "fasim")
    dna_masked, rnas = _cfg4_inputs()
    synth.write_fasta(str(tmp_path / "dna.fa"), f"syn|chrG|1-{len(dna_masked)}", dna_masked)
    with open(tmp_path / "four.fa", "wb") as f:
        for k, r in enumerate(rnas):
            f.write(f">syn3k_{k + 1}\n".encode() + r[:1700] + b"\n" + r[1700:] + b"\n")
    (tmp_path / "out").mkdir()
    r = subprocess.run([exe, "-f1", "dna.fa", "-f2", "four.fa", "-O", "out/", "-lg", "40", "--upper", "--devices", "0,0,0", "--stats"],
                       cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert b"end to end" in r.stderr and b"3 device shard" in r.stderr
    for k in range(1, 5):
        assert (tmp_path / "out" / f"syn-syn3k_{k}-dna-TFOsorted").read_bytes() == _gz(golden_dir, f"cfg4_q{k}.TFOsorted.gz")
        for level in (1, 2):
            assert (tmp_path / "out" / f"syn-syn3k_{k}-dna-TFOclass{level}-15-40").read_bytes() == _gz(golden_dir, f"cfg4_q{k}.TFOclass{level}.gz")


def test_cli_devices_sharded_identical_to_unsharded(golden_dir, tmp_path):
    exe = os.path.join(entry.PKG_DIR, "fasim")
    for f in ("H19.fa", "planted40k.fa"):
        (tmp_path / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
    (tmp_path / "out").mkdir()
    subprocess.run([exe, "-f1", "planted40k.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40", "--devices", "0,0,0"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL)
    assert (tmp_path / "out" / "syn-H19-planted40k-TFOsorted").read_bytes() == open(os.path.join(golden_dir, "planted40k.TFOsorted"), "rb").read()
    for level in (1, 2):
        got = (tmp_path / "out" / f"syn-H19-planted40k-TFOclass{level}-15-40").read_bytes()
        assert got == open(os.path.join(golden_dir, f"planted40k.TFOclass{level}"), "rb").read()


def test_cli_accumulate_records_is_bug_compatible(golden_dir, tmp_path):
    """--accumulate-records reproduces the reference's reader on a multi-record DNA file (defect B1): record 2 is scanned
    as record 1 + record 2, its header is parsed with the stale field counter (start 0, chr of record 1), and all
    triplexes land in one output set named after record 1."""
    exe = os.path.join(entry.PKG_DIR, "fasim")
    for f in ("H19.fa", "b1_two.fa"):
        (tmp_path / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
    (tmp_path / "out").mkdir()
    subprocess.run([exe, "-f1", "b1_two.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40", "--accumulate-records"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL)
    assert (tmp_path / "out" / "hg19-H19-b1_two-TFOsorted").read_bytes() == open(os.path.join(golden_dir, "b1_two.TFOsorted"), "rb").read()
    for level in (1, 2):
        got = (tmp_path / "out" / f"hg19-H19-b1_two-TFOclass{level}-15-40").read_bytes()
        assert got == open(os.path.join(golden_dir, f"b1_two.TFOclass{level}"), "rb").read()


def test_cluster_guard_flag(mod, engine, h19, golden_dir):
    """-lg below ~2*-ds lets a triplex mid-point fall within -ds of the query start, where the reference's clustering
    never terminates: refused by default, defined result with TAIL_CLAMP_CLUSTER; inputs the reference handles are
    unaffected by the flag."""
    hdr, dna = synth.read_fasta(os.path.join(golden_dir, "planted40k.fa"))
    _, chro, start = mod.parse_dna_header(hdr)
    engine.set_query(h19)
    p = mod.default_params(cLength=40)
    res = engine.scan(dna, p)
    gold = open(os.path.join(golden_dir, "planted40k.TFOsorted"), "rb").read()
    assert mod.tfosorted(res, chro, start, p, mod.TAIL_CLAMP_CLUSTER) == gold
    p2 = mod.default_params(cLength=20, cDistance=40)
    res2 = engine.scan(dna, p2)
    near_start = [t for t in res2.triplexes() if t[7] > 20 and (t[0] + t[1]) // 2 < 40]
    if not near_start:
        pytest.skip("no triplex within -ds of the query start in this fixture")
    with pytest.raises(mod.FasimError):
        mod.tfosorted(res2, chro, start, p2)
    text = mod.tfosorted(res2, chro, start, p2, mod.TAIL_CLAMP_CLUSTER)
    assert text.count(b"\n") > 10


# ---- BASELINE configs 3 and 5 on a chromosome-like record ------------------------------------------------------------
@pytest.mark.parametrize("name,rna_src,kw", [
    ("big_meg3", "MEG3.fa", dict()),
    ("big_malat1", "MALAT1.fa", dict()),
    ("big_neat1", "NEAT1.fa", dict()),
    ("big_syn10k", None, dict(ntMax=1000)),
])
def test_genome_like_1mb_long_queries(mod, golden_dir, name, rna_src, kw):
    """MEG3 / MALAT1 / NEAT1 (configs[2]) and a 10 kb lncRNA with -na 1000 (configs[4]) x a 1.2 Mb chromosome-like record
    (telomere and whole-segment N runs, soft-masked repeats upper-cased, microsatellites, purine tracts): the three output
    files equal the compiled reference's (tests/golden/big_*.gz, made by make_golden.py big)."""
    dna = synth.genome_like(1200000, 3003, soft_mask=False)
    rna = synth.read_fasta(os.path.join(golden_dir, rna_src))[1] if rna_src else synth.random_rna(10000, 515)
    rna_name = rna_src[:-3] if rna_src else "syn10k"
    p = mod.default_params(**kw)
    e = mod.Engine(0)
    e.set_query(rna)
    e.load_dna(dna)
    res = e.scan(None, p)
    st = res.stats
    assert st["segments_skipped"] > 0 and st["stage1_word_reruns"] > 0      # all-N segments skipped, segments with N take the stage-1 pass
    assert mod.tfosorted(res, "chrG", 1, p) == _gz(golden_dir, name + ".TFOsorted.gz")
    for level in (1, 2):
        assert mod.tfoclass(res, level, "chrG", 1, len(dna), rna_name, p) == _gz(golden_dir, f"{name}.TFOclass{level}.gz")
    print(f"{name}: {st['units']} units, {st['candidates']} candidates, hazard {st['hazard_units']}, overflow {st['stage2_overflow_units']}, "
          f"rev_exact {st['rev_exact']}, replays {st['exact_replays']}, word reruns {st['align_word_reruns']}, {res.count} records, {st['t_total_s']:.2f} s")
    e.close()


# ---- row f3: forward sweep of classic SIM (-F) ---------------------------------------------------------------------------
def test_sim_forward_sweep_node_lists(mod, engine, h19, golden_dir, oracle_build):
    """k_sim_forward + the host replay of addnode: the K = 50 node list SIM() holds after its first sweep (sim.h:506-571), for
    demo units of all four strand / direction classes and a planted 3 kb target, against the oracle's restatement (which is
    pinned end to end by the reference's own -F outputs, tests/test_oracle_golden.py)."""
    o = helpers.Oracle(oracle_build)
    _, dna = synth.read_fasta(os.path.join(golden_dir, "testDNA.fa"))
    engine.set_query(h19)
    targets, mins = [], []
    for enc in (0, 1, 12, 13, 30, 47):
        t, _ = o.encode_unit(dna, enc)
        targets.append(t)
        mins.append(int(o.stage1_max(h19, t) * 0.8))
    small = synth.planted_dna(3000, 77, h19, every=400)
    t, _ = o.encode_unit(small, 5)
    targets.append(t)
    mins.append(int(o.stage1_max(h19, t) * 0.8))
    targets.append(b"ACGT" * 5)          # shorter than one strip of rows; nothing above the threshold
    mins.append(100000)
    got = engine.sim_forward(targets, mins)
    for k, (t, ms) in enumerate(zip(targets, mins)):
        exp = o.sim_forward_nodes(h19, t, ms)
        assert got[k] == exp, (k, len(got[k]), len(exp), got[k][:2], exp[:2])
    assert len(got[0]) == 50 and got[-1] == []


# ---- BASELINE config 2 at its full size ----------------------------------------------------------------------------------
def test_full_size_50mb_reference_validated_digest_and_sharding_invariance(mod, h19):
    """The bench's whole 50 Mb record (splitmix64 seed 12345), H19, default parameters.
    (1) Cut into the same 13 slices at segment boundaries as tests/parity/parity_sharded.py, every slice scanned and its three
    output files hashed in that script's order: the digest must be the one recorded in profiles/r01_parity_big_vs_reference.log
    for the run in which all 39 files were byte-identical to the compiled reference on the GPU box (260 649 triplex lines).
    (2) Size-independent property: the whole record scanned as 3 contiguous segment shards and merged natively equals the
    unsharded scan, record for record."""
    import hashlib
    total, nsl = 50_000_000, 13
    dna = mod.synth_dna(total, 12345)
    p = mod.default_params()
    e = mod.Engine(0)
    e.set_query(h19)
    nseg = (total - 5000) // 4900 + 1
    per = (nseg + nsl - 1) // nsl
    h = hashlib.sha256()
    lines = 0
    for k in range(nsl):
        a, b = k * per, min(nseg, (k + 1) * per)
        if a >= b:
            break
        lo, hi = a * 4900, min(total, (b - 1) * 4900 + 5000)
        res = e.scan(dna[lo:hi], p)
        tfo, c1, c2 = mod.tail_outputs(res, "chrB", lo + 1, hi - lo, "H19", p)
        for text in (c1, c2, tfo):                      # file-name order: ...-TFOclass1-15-50, ...-TFOclass2-15-50, ...-TFOsorted
            h.update(text)
        lines += tfo.count(b"\n") - 1
    assert lines == 260649
    assert h.hexdigest()[:16] == "3868117966c38486"
    e.load_dna(dna)
    whole = e.scan(None, p)
    assert whole.count == 265583 and whole.stats["units"] == 489840
    nall = mod.segment_count(total, p)
    cuts = [0, nall // 3, 2 * nall // 3, nall]
    parts = [e.scan(None, p, cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)]
    merged = mod.merge_results(parts)
    assert merged.count == whole.count and merged.recs == whole.recs and merged.pool == whole.pool
    e.close()


def _simscan_expected(golden_dir, name, c_length):
    """fixture X lines of every unit, with LongTarget()'s tail filter applied (Fasim-LongTarget.cpp:589-597; SIM() itself only
    filters on nt), as scan() tuples"""
    import struct
    out, seg, enc = [], 0, 0
    for line in helpers.gunzip(os.path.join(golden_dir, name)).decode().splitlines():
        f = line.split(" ")
        if f[0] == "V":
            seg, enc = int(f[1]), int(f[2])
        elif f[0] == "X":
            ident = struct.unpack("<f", struct.pack("<I", int(f[10], 16)))[0]
            tri = struct.unpack("<f", struct.pack("<I", int(f[11], 16)))[0]
            if float(int(f[9])) >= 0.0 and ident >= 60.0 and tri >= 1.0 and int(f[8]) >= c_length:
                out.append((int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[9]),
                            int(f[10], 16), int(f[11], 16), f[12].encode(), f[13].encode(), seg, enc))
    return out


def test_classic_sim_scan_end_to_end(mod, h19, golden_dir):
    """-F through fasim_scan (params.classicSim): forward sweep and the re-sweeps between the K rounds on the GPU (k_sim_forward,
    k_sim_resweep), traceback / triplex records on the host; every record of every unit equals the reference's own SIM()
    (ref_probe simscan fixtures), demo and planted 12 kb."""
    e = mod.Engine(0)
    e.set_query(h19)
    p = mod.default_params(classicSim=1, cLength=20)
    _, dna = synth.read_fasta(os.path.join(golden_dir, "testDNA.fa"))
    res = e.scan(dna, p)
    exp = _simscan_expected(golden_dir, "demoF.simscan.gz", 20)
    assert len(exp) > 100 and res.triplexes() == exp
    assert res.stats["kernel_launches"][7] > 0, "k_sim_forward must have run"
    dna2 = synth.planted_dna(12000, 909, h19, every=700)
    res = e.scan(dna2, p)
    assert res.triplexes() == _simscan_expected(golden_dir, "simF12k.simscan.gz", 20)
    e.close()


def test_classic_sim_resweep_self_check(mod, h19, golden_dir, monkeypatch):
    """The device re-sweeps of -F (k_sim_resweep, resumable line sweeps + batch node-list replay) against the host restatement
    of sim.h:884-1141 inside the engine, round by round (FASIM_SIM_RESWEEP=check: node lists and `min` of every unit after
    every round must be identical, or the scan fails), with a small step budget so that units are suspended and resumed; and
    host-only re-sweeps give the same records."""
    dna = synth.planted_dna(12000, 909, h19, every=700)
    p = mod.default_params(classicSim=1, cLength=20)
    exp = _simscan_expected(golden_dir, "simF12k.simscan.gz", 20)
    e = mod.Engine(0)
    e.set_query(h19)
    monkeypatch.setenv("FASIM_SIM_RESWEEP", "check")
    monkeypatch.setenv("FASIM_SIM_BUDGET", "700")
    assert e.scan(dna, p).triplexes() == exp
    monkeypatch.delenv("FASIM_SIM_BUDGET")
    monkeypatch.setenv("FASIM_SIM_RESWEEP", "host")
    assert e.scan(dna, p).triplexes() == exp
    monkeypatch.delenv("FASIM_SIM_RESWEEP")
    res = e.scan(dna, p)
    assert res.triplexes() == exp and res.stats["kernel_launches"][7] > 1
    # several batches in flight on the worker engines (their re-sweep launches overlap; the kernel variant follows the units in
    # flight over all of them)
    e.set_option("seg_batch", 1)
    assert e.scan(dna, p).triplexes() == exp
    e.close()


def test_classic_sim_long_query(mod, golden_dir):
    """-F with MALAT1 (8 708 nt): start rows beyond the 13-bit fields of the first version of k_sim_forward (16-bit fields now);
    every record equals the reference's own SIM() (`make_golden.py simF_long`)."""
    _, malat = synth.read_fasta(os.path.join(golden_dir, "MALAT1.fa"))
    _, dna = synth.read_fasta(os.path.join(golden_dir, "malat1F_dna.fa"))
    e = mod.Engine(0)
    e.set_query(malat)
    res = e.scan(dna, mod.default_params(classicSim=1, cLength=20))
    exp = _simscan_expected(golden_dir, "malat1F.simscan.gz", 20)
    assert len(exp) > 0 and res.triplexes() == exp
    assert res.stats["kernel_launches"][7] > 0
    e.close()


def test_cli_classic_sim_writes_reference_files(golden_dir, tmp_path):
    """`fasim -F -lg 40` on the demo, and the reference's own driver with the LongTarget() binding and -F: both write the files
    `fasim_ref -F` writes."""
    exes = [os.path.join(entry.PKG_DIR, "fasim")]
    hb = os.path.join(entry.ROOT, "oracle", "_ref", "fasim_ref_hipbind")
    if os.access(hb, os.X_OK):
        exes.append(hb)
    for k, exe in enumerate(exes):
        wd = tmp_path / f"run{k}"
        (wd / "out").mkdir(parents=True)
        for f in ("H19.fa", "testDNA.fa"):
            (wd / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
        subprocess.run([exe, "-f1", "testDNA.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40", "-F"], cwd=wd, check=True, stdout=subprocess.DEVNULL)
        assert (wd / "out" / "hg19-H19-testDNA-TFOsorted").read_bytes() == open(os.path.join(golden_dir, "demoF_lg40.TFOsorted"), "rb").read(), exe
        for level in (1, 2):
            got = (wd / "out" / f"hg19-H19-testDNA-TFOclass{level}-15-40").read_bytes()
            assert got == open(os.path.join(golden_dir, f"demoF_lg40.TFOclass{level}"), "rb").read(), exe


# ---- edge cases of the round-2 paths ---------------------------------------------------------------------------------------
def test_scan_queries_mixed_query_classes(mod, golden_dir, h19):
    """One batch with a 100-nt query (stripe-faithful kernels only), H19 (one systolic tile) and MALAT1 (three tiles): the
    workers switch query and kernel variant (striped, systolic, banded) from item to item; every result equals the scan of that
    query alone."""
    _, q100 = synth.read_fasta(os.path.join(golden_dir, "h19_100.fa"))
    _, malat = synth.read_fasta(os.path.join(golden_dir, "MALAT1.fa"))
    _, dna = synth.read_fasta(os.path.join(golden_dir, "planted40k.fa"))
    p = mod.default_params(cLength=30)
    e = mod.Engine(0)
    e.set_option("seg_batch", 3)           # several items per query
    rnas = [q100, h19, malat, h19[:1600]]
    batch = e.scan_queries(rnas, dna, p)
    for q, rna in enumerate(rnas):
        e.set_query(rna)
        alone = e.scan(dna, p)
        assert batch[q].recs == alone.recs and batch[q].pool == alone.pool, q
        assert batch[q].stats["units"] == alone.stats["units"] and batch[q].stats["candidates"] == alone.stats["candidates"]
    e.close()


def test_tiny_and_boundary_records(mod, engine, h19, oracle_build, tmp_path):
    """DNA records of 20, 4 900, 4 901 and 5 000 nt (segment boundary cases of cutSequence, fastsim.h:71-90) against the oracle."""
    engine.set_query(h19)
    p = mod.default_params(cLength=20)
    rna_fa = tmp_path / "rna.fa"
    rna_fa.write_bytes(b">H19\n" + h19 + b"\n")
    for n in (20, 4900, 4901, 5000):
        dna = synth.planted_dna(max(n, 200), 4000 + n, h19, every=300)[:n]
        res = engine.scan(dna, p)
        text = mod.tfosorted(res, "chrT", 1, p)
        fa = tmp_path / f"d{n}.fa"
        fa.write_bytes(b">syn|chrT|1-%d\n" % n + dna + b"\n")
        exp = helpers.oracle_cli(oracle_build, "tfosorted", str(rna_fa), str(fa), "-lg", "20")
        assert text == exp, n
        assert res.stats["segments"] == mod.segment_count(n, p)


def test_cli_more_device_shards_than_segments(golden_dir, tmp_path):
    exe = os.path.join(entry.PKG_DIR, "fasim")
    for f in ("H19.fa", "testDNA.fa"):
        (tmp_path / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
    (tmp_path / "out").mkdir()
    subprocess.run([exe, "-f1", "testDNA.fa", "-f2", "H19.fa", "-O", "out/", "-lg", "40", "--devices", "0,0,0,0"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL)
    assert (tmp_path / "out" / "hg19-H19-testDNA-TFOsorted").read_bytes() == open(os.path.join(golden_dir, "demo_lg40.TFOsorted"), "rb").read()
