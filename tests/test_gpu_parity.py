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
    """S = calc_score_once, P = ssw_pre_align, A = ssw_align on the reference's own outputs (batch.rsp)."""
    reqs = open(os.path.join(golden_dir, "batch.req")).read().splitlines()
    rsps = open(os.path.join(golden_dir, "batch.rsp")).read().splitlines()
    by_query = {}
    for rq, rs in zip(reqs, rsps):
        f = rq.split(" ")
        by_query.setdefault(f[1], []).append((f, rs.split(" ")))
    checked = {"S": 0, "P": 0, "A": 0}
    for q, items in by_query.items():
        engine.set_query(q.encode())
        targets = [f[2].encode() for f, _ in items if f[0] in "SP"]
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
