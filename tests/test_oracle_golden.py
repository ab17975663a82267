"""The CPU restatement (oracle/) against the reference's own outputs (tests/golden/, made by
tests/golden/make_golden.py from the compiled reference).  CPU only; pins the oracle."""
import os

import pytest

import helpers
import synth


def _write(tmp_path, name, data):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


def test_demo_scan_identical(oracle_build, golden_dir):
    out = helpers.oracle_cli(oracle_build, "scan", os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "testDNA.fa"))
    assert out == helpers.gunzip(os.path.join(golden_dir, "demo.scan.gz"))


@pytest.mark.parametrize("name,opts", [
    ("demo_lg40.TFOsorted", ["-lg", "40"]),
    ("demo_default.TFOsorted", []),
    ("demo_t1_r3.TFOsorted", ["-lg", "30", "-t", "1", "-r", "3"]),
])
def test_demo_tfosorted_identical(oracle_build, golden_dir, name, opts):
    out = helpers.oracle_cli(oracle_build, "tfosorted", os.path.join(golden_dir, "H19.fa"),
                             os.path.join(golden_dir, "testDNA.fa"), *opts)
    assert out == open(os.path.join(golden_dir, name), "rb").read()


@pytest.mark.parametrize("stem,dna_name,opts", [
    ("demo_lg40", "testDNA.fa", ["-lg", "40"]),
    ("demo_default", "testDNA.fa", []),
    ("demo_t1_r3", "testDNA.fa", ["-lg", "30", "-t", "1", "-r", "3"]),
    ("planted40k", "planted40k.fa", ["-lg", "40"]),
    ("q2cat", "q2cat.fa", ["-o", "0", "-lg", "40"]),
])
def test_tfoclass_bedgraph_identical(oracle_build, golden_dir, stem, dna_name, opts):
    """print_cluster (Fasim-LongTarget.cpp:694): both bedGraph files the reference CLI wrote next to the -TFOsorted file."""
    for level in (1, 2):
        out = helpers.oracle_cli(oracle_build, "tfoclass", os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, dna_name),
                                 "-level", str(level), "-threads", "8", *opts)
        assert out == open(os.path.join(golden_dir, f"{stem}.TFOclass{level}"), "rb").read()


def test_planted40k_scan_and_tfosorted(oracle_build, golden_dir):
    rna, dna = os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "planted40k.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-threads", "8")
    gold = helpers.gunzip(os.path.join(golden_dir, "planted40k.scan.gz"))
    assert out == gold
    _, units = helpers.parse_scan(gold)
    assert sum(u["stage1"] >= 251 for u in units) >= 20, "fixture must exercise the byte-overflow (Q1) path"
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "40", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "planted40k.TFOsorted"), "rb").read()


def test_rnd30k(oracle_build, golden_dir, tmp_path):
    dna = _write(tmp_path, "rnd30k.fa", b">syn|chrS|1-30000\n" + synth.random_dna(30000, 12345) + b"\n")
    rna = os.path.join(golden_dir, "H19.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-detail", "0", "-threads", "8")
    assert out == helpers.gunzip(os.path.join(golden_dir, "rnd30k.scan.gz"))
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "40", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "rnd30k.TFOsorted"), "rb").read()


def test_batch_vectors(oracle_build, golden_dir):
    o = helpers.Oracle(oracle_build)
    reqs = open(os.path.join(golden_dir, "batch.req")).read().splitlines()
    rsps = open(os.path.join(golden_dir, "batch.rsp")).read().splitlines()
    assert len(reqs) == len(rsps)
    kinds = set()
    for rq, rs in zip(reqs, rsps):
        f, g = rq.split(" "), rs.split(" ")
        q, t = f[1].encode(), f[2].encode()
        kinds.add(f[0])
        if f[0] == "S":
            assert o.stage1_max(q, t) == int(g[1]), rq[:80]
        elif f[0] == "P":
            assert o.pre_align(q, t) == [int(x) for x in g[2:]], rq[:80]
        elif f[0] == "K":
            cands = o.candidates(o.pre_align(q, t), int(f[3]))
            flat = [int(x) for x in g[2:]]
            assert cands == list(zip(flat[0::2], flat[1::2])), rq[:80]
        elif f[0] == "A":
            five, cig = o.align(q, t)
            exp = tuple(int(x) for x in g[1:6])
            if exp[0] == 0:
                assert five[0] == 0
            else:
                assert five == exp and (cig or "*") == g[6], rq[:80]
    assert kinds == {"S", "P", "K", "A"}


def test_q2_units_fixture(oracle_build, golden_dir):
    """19 whole segments, each holding a unit where the signed lazy-F exit (Q2) changes the column maxima."""
    rna, dna = os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "q2cat.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-o", "0", "-detail", "0", "-threads", "8")
    assert out == helpers.gunzip(os.path.join(golden_dir, "q2cat.scan.gz"))
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-o", "0", "-lg", "40", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "q2cat.TFOsorted"), "rb").read()
    # the fixture really contains Q2 units: the unsigned-exit variant of the oracle gives different columns
    import ctypes
    o = helpers.Oracle(oracle_build)
    o.lib.fo_pre_align_noq2.restype = None
    o.lib.fo_pre_align_noq2.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    _, q = synth.read_fasta(rna)
    _, d = synth.read_fasta(dna)
    t, _ = o.encode_unit(d[:5000], 44)
    b = (ctypes.c_int * len(t))()
    o.lib.fo_pre_align_noq2(q, len(q), t, len(t), b)
    assert o.pre_align(q, t) != list(b)


@pytest.mark.parametrize("name", ["meg3", "malat1"])
def test_long_queries(oracle_build, golden_dir, name):
    """The reference's example lncRNAs MEG3 (1 582 nt) and MALAT1 (8 708 nt) against planted synthetic DNA."""
    rna = os.path.join(golden_dir, name.upper() + ".fa")
    dna = os.path.join(golden_dir, name + "_dna.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-detail", "0", "-threads", "8")
    assert out == helpers.gunzip(os.path.join(golden_dir, name + ".scan.gz"))
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "40", "-threads", "8")
    assert out == open(os.path.join(golden_dir, name + ".TFOsorted"), "rb").read()


def test_stage1_score_beyond_16383(oracle_build, golden_dir):
    """A unit whose exact stage-1 score is 19 905 (only the 16-bit pass of calc_score_once can hold it)."""
    rna, dna = os.path.join(golden_dir, "satq.fa"), os.path.join(golden_dir, "sat5k.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-detail", "0", "-threads", "8")
    gold = helpers.gunzip(os.path.join(golden_dir, "sat5k.scan.gz"))
    assert out == gold
    _, units = helpers.parse_scan(gold)
    assert max(u["stage1"] for u in units) > 16383
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "40", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "sat5k.TFOsorted"), "rb").read()


@pytest.mark.parametrize("name", ["h19_700", "h19_100"])
def test_short_queries(oracle_build, golden_dir, name):
    rna, dna = os.path.join(golden_dir, name + ".fa"), os.path.join(golden_dir, name + "_dna.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-detail", "0", "-threads", "8")
    assert out == helpers.gunzip(os.path.join(golden_dir, name + ".scan.gz"))
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "25", "-threads", "8")
    assert out == open(os.path.join(golden_dir, name + ".TFOsorted"), "rb").read()


def test_untidy_input(oracle_build, golden_dir):
    """N runs (one whole segment is skipped), lower-case letters and IUPAC codes in the DNA."""
    rna, dna = os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "messy.fa")
    out = helpers.oracle_cli(oracle_build, "scan", rna, dna, "-detail", "0", "-threads", "8")
    gold = helpers.gunzip(os.path.join(golden_dir, "messy.scan.gz"))
    assert out == gold
    meta, _ = helpers.parse_scan(gold)
    assert meta["skipped"], "fixture must contain a skipped all-N segment"
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "30", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "messy.TFOsorted"), "rb").read()


# ---- row f3: the -F path (classic SIM, sim.h:410-1143) ------------------------------------------------------------------
def test_classic_sim_demo_units_identical(oracle_build, golden_dir):
    """oracle `simscan` (oracle/fasim_sim_oracle.cpp) against the reference's own SIM() for all 48 units of the demo:
    stage-1 score, threshold and every triplex (coordinates, nt, score, identity / stability as float bits, both strings)."""
    out = helpers.oracle_cli(oracle_build, "simscan", os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "testDNA.fa"), "-threads", "8")
    gold = helpers.gunzip(os.path.join(golden_dir, "demoF.simscan.gz"))
    assert out == gold
    assert gold.count(b"\nX ") > 500


def test_classic_sim_planted12k_identical(oracle_build, golden_dir, tmp_path):
    _, rna = synth.read_fasta(os.path.join(golden_dir, "H19.fa"))
    dna = _write(tmp_path, "simF12k.fa", b">syn|chrF|1-12000\n" + synth.planted_dna(12000, 909, rna, every=700) + b"\n")
    out = helpers.oracle_cli(oracle_build, "simscan", os.path.join(golden_dir, "H19.fa"), dna, "-threads", "8")
    assert out == helpers.gunzip(os.path.join(golden_dir, "simF12k.simscan.gz"))


def test_classic_sim_cli_files_identical(oracle_build, golden_dir):
    """`fasim_ref -F -lg 40` on the demo: -TFOsorted and both -TFOclass files from the oracle's -F path."""
    rna, dna = os.path.join(golden_dir, "H19.fa"), os.path.join(golden_dir, "testDNA.fa")
    out = helpers.oracle_cli(oracle_build, "tfosorted", rna, dna, "-lg", "40", "-F", "1", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "demoF_lg40.TFOsorted"), "rb").read()
    out = helpers.oracle_cli(oracle_build, "tfoclass", rna, dna, "-lg", "40", "-F", "1", "-level", "1", "-threads", "8")
    assert out == open(os.path.join(golden_dir, "demoF_lg40.TFOclass1"), "rb").read()
