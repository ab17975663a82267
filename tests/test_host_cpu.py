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
    m = _mod()
    hdr = open(os.path.join(entry.ROOT, "include", "fasim_hip.h")).read()
    declared = set(re.findall(r"\b(fasim_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"fasim_engine", "fasim_params", "fasim_alignment", "fasim_triplex", "fasim_result", "fasim_scan_stats"}
    L = m.lib()
    assert declared, "header parse failed"
    for name in sorted(declared):
        assert hasattr(L, name), f"libfasim_hip.so does not export {name}"
    assert set(m.EXPORTS) == declared


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
