"""The N>1 path on CPU: two ranks over gloo.  Each rank owns a contiguous block of segments (what bench.py / a
multi-GPU driver does with fasim_scan(seg_first, seg_count)); records are gathered with the package's
gather_results() and rank 0 runs the host tail.  The per-rank records are rebuilt from the reference's own
fastSIM() output (tests/golden/planted40k.scan.gz), so the merged -TFOsorted must equal the reference file."""
import ctypes as C
import os
import struct
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
import __graft_entry__ as entry

ROOT = entry.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")


def _bits_to_float(h):
    return struct.unpack("<f", struct.pack("<I", int(h, 16)))[0]


def records_for_segments(mod, units, seg_lo, seg_hi, c_length=40):
    """ScanResult holding what fasim_scan(seg_first=seg_lo, seg_count=seg_hi-seg_lo) returns: fastSIM() records that
    pass LongTarget()'s tail filter (Fasim-LongTarget.cpp:589-597), in canonical order."""
    recs, pool = [], bytearray()
    for u in units:
        if not (seg_lo <= u["seg"] < seg_hi):
            continue
        for x in u["triplexes"]:
            stari, endi, starj, endj, strand, reverse, rule, nt, score = (int(v) for v in x[:9])
            identity, tri = _bits_to_float(x[9]), _bits_to_float(x[10])
            if not (score >= 0 and identity >= 60.0 and tri >= 1.0 and nt >= c_length):
                continue
            t = mod.Triplex()
            t.stari, t.endi, t.starj, t.endj, t.strand, t.reverse, t.rule, t.nt = stari, endi, starj, endj, strand, reverse, rule, nt
            t.score, t.identity, t.tri_score, t.seg, t.enc = float(score), identity, tri, u["seg"], u["enc"]
            t.tfo_off = len(pool)
            pool += x[11].encode() + b"\0"
            t.tts_off = len(pool)
            pool += x[12].encode() + b"\0"
            recs.append(bytes(t))
    return mod.ScanResult(b"".join(recs), bytes(pool), {})


def _worker(rank, world, port, out_path, bounds=None, chunk=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mod = entry.load()
        if chunk:
            mod.GATHER_CHUNK = chunk                    # many small pieces per shard: the chunked transfer plan
        _, units = helpers.parse_scan(helpers.gunzip(os.path.join(GOLD, "planted40k.scan.gz")))
        nseg = max(u["seg"] for u in units) + 1
        if bounds is None:
            first, count = mod.shard_segments(nseg, rank, world)
        else:                                           # hand-made shards, some of them empty
            first, count = bounds[rank], bounds[rank + 1] - bounds[rank]
        mine = records_for_segments(mod, units, first, first + count)
        merged = mod.gather_results(mine, dist, rank, world, "cpu")
        if rank == 0:
            whole = records_for_segments(mod, units, 0, nseg)
            assert merged.recs == whole.recs and merged.pool == whole.pool, "merged shards differ from the unsharded order"
            p = mod.default_params(cLength=40)
            text = mod.tfosorted(merged, "chrP", 1001, p)
            open(out_path, "wb").write(text)
        else:
            assert merged is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scan_gathers_to_reference_output(tmp_path, world):
    if not os.path.exists(os.path.join(entry.PKG_DIR, "libfasim_hip.so")):
        entry.build()
    out = tmp_path / "merged.TFOsorted"
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(out)), nprocs=world, join=True)
    assert out.read_bytes() == open(os.path.join(GOLD, "planted40k.TFOsorted"), "rb").read()


def test_eight_ranks_uneven_and_empty_shards_chunked_gather(tmp_path):
    """World 8 (the node size of BASELINE config 4) with hand-made shards of 0, 0, 3, 0, 1, 5, 0, 0 segments and a 4 KB transfer
    chunk (every shard goes in many pieces): the merged records and the -TFOsorted text equal the unsharded ones."""
    if not os.path.exists(os.path.join(entry.PKG_DIR, "libfasim_hip.so")):
        entry.build()
    out = tmp_path / "merged8.TFOsorted"
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(8, port, str(out), [0, 0, 0, 3, 3, 4, 9, 9, 9], 4096), nprocs=8, join=True)
    assert out.read_bytes() == open(os.path.join(GOLD, "planted40k.TFOsorted"), "rb").read()


def test_shard_segments_cover_everything():
    mod = entry.load()
    for nseg in (1, 7, 8, 9, 10205):
        for world in (1, 2, 3, 8):
            blocks = [mod.shard_segments(nseg, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == nseg
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
