"""Shared helpers for the test-suite (probe line protocol, oracle ctypes binding, inputs)."""
import ctypes
import gzip
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
REF_FASIM = os.path.join(ROOT, "oracle", "_ref", "fasim_ref")


def gunzip(path):
    with gzip.open(path, "rb") as f:
        return f.read()


def oracle_cli(build_dir, *args):
    exe = os.path.join(build_dir, "fasim_oracle")
    return subprocess.run([exe, *args], check=True, stdout=subprocess.PIPE).stdout


class Oracle:
    """ctypes view of oracle/_build/libfasim_oracle.so (CHECKER ONLY)."""

    def __init__(self, build_dir):
        self.lib = ctypes.CDLL(os.path.join(build_dir, "libfasim_oracle.so"))
        L = self.lib
        L.fo_stage1_max.restype = ctypes.c_int
        L.fo_stage1_max.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.fo_pre_align.restype = None
        L.fo_pre_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.fo_pick_candidates.restype = ctypes.c_int
        L.fo_pick_candidates.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.fo_align.restype = ctypes.c_int
        L.fo_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                               ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.fo_sim_forward_nodes.restype = ctypes.c_int
        L.fo_sim_forward_nodes.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_long,
                                           ctypes.POINTER(ctypes.c_long), ctypes.c_int]
        L.fo_encode_unit.restype = None
        L.fo_encode_unit.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]

    def stage1_max(self, q: bytes, t: bytes) -> int:
        return self.lib.fo_stage1_max(q, len(q), t, len(t))

    def pre_align(self, q: bytes, t: bytes):
        out = (ctypes.c_int * len(t))()
        self.lib.fo_pre_align(q, len(q), t, len(t), out)
        return list(out)

    def candidates(self, cols, thr):
        n = len(cols)
        arr = (ctypes.c_int * n)(*cols)
        s = (ctypes.c_int * (n + 1))()
        p = (ctypes.c_int * (n + 1))()
        k = self.lib.fo_pick_candidates(arr, n, thr, s, p, n + 1)
        return [(s[i], p[i]) for i in range(k)]

    def align(self, q: bytes, w: bytes):
        out5 = (ctypes.c_int * 5)()
        cig = (ctypes.c_uint32 * 4096)()
        k = self.lib.fo_align(q, len(q), w, len(w), out5, cig, 4096)
        assert k >= 0
        return tuple(out5), cigar_to_string(cig[:k])

    def sim_forward_nodes(self, q: bytes, t: bytes, min_score: int):
        """node list after the first sweep of SIM() (oracle/fasim_sim_oracle.cpp), as 9-tuples in list order"""
        out = (ctypes.c_long * (9 * 50))()
        k = self.lib.fo_sim_forward_nodes(q, len(q), t, len(t), min_score, out, 50)
        return [tuple(out[9 * x + y] for y in range(9)) for x in range(k)]

    def encode_unit(self, seg: bytes, enc: int):
        t = ctypes.create_string_buffer(len(seg))
        s = ctypes.create_string_buffer(len(seg) + 1)
        self.lib.fo_encode_unit(seg, len(seg), enc, t, s)
        return t.raw[:len(seg)], s.raw[:len(seg)].rstrip(b"\0")


def cigar_to_string(cigar):
    ops = "MIDNSHP=X"
    return "".join(f"{c >> 4}{'M' if (c & 15) > 8 else ops[c & 15]}" for c in cigar)


def parse_scan(text: bytes):
    """ref_probe/fasim_oracle `scan` protocol -> list of unit dicts (see oracle/ref_probe.cpp)."""
    units, cur, cand = [], None, None
    meta = {}
    for line in text.decode().splitlines():
        f = line.split(" ")
        k = f[0]
        if k == "Q":
            meta = {"m": int(f[1]), "dna_len": int(f[2]), "nseg": int(f[3]), "skipped": []}
        elif k == "K":
            meta["skipped"].append(int(f[1]))
        elif k == "U":
            cur = {"seg": int(f[1]), "enc": int(f[2]), "dna_start": int(f[3]), "strand": int(f[4]), "para": int(f[5]),
                   "rule": int(f[6]), "n": int(f[7]), "stage1": int(f[8]), "thr": int(f[9]), "colhash": f[10],
                   "nhits": int(f[11]), "ncand": int(f[12]), "hits": [], "cands": [], "triplexes": []}
            units.append(cur)
        elif k == "H":
            cur["hits"].append((int(f[1]), int(f[2])))
        elif k == "C":
            cand = {"score": int(f[1]), "pos": int(f[2]), "tries": []}
            cur["cands"].append(cand)
        elif k == "T":
            cand["tries"].append({"it": int(f[1]), "L": int(f[2]), "score": int(f[3]), "ref_begin": int(f[4]),
                                  "ref_end": int(f[5]), "q_begin": int(f[6]), "q_end": int(f[7]), "cigar": f[8]})
        elif k == "X":
            cur["triplexes"].append(tuple(f[1:]))
    return meta, units


def fnv1a_ints(vals):
    h = 1469598103934665603
    for v in vals:
        x = v & 0xFFFFFFFF
        for b in range(4):
            h ^= (x >> (8 * b)) & 0xFF
            h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def have_ref_probe():
    return os.access(REF_PROBE, os.X_OK)


def ref_batch(requests):
    """Run the compiled reference probe (if present) on S/P/K/A request lines."""
    req = ("\n".join(requests) + "\n").encode()
    out = subprocess.run([REF_PROBE, "batch"], input=req, check=True, stdout=subprocess.PIPE).stdout
    return out.decode().splitlines()
