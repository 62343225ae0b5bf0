"""Shared test plumbing: fixture loading, the ctypes face of the CPU oracle, and a Python model of the
reference's record loop that drives the oracle the way the reference's main() drives its hot path."""
import ctypes
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
INPUTS = os.path.join(GOLD, "inputs")


def golden_meta():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


def golden_arrays(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def input_path(name):
    """Resolve REF_TEST/x.fastq and EDGE/x names used in golden.json to fixture files."""
    base = name.split("/", 1)[1]
    p = os.path.join(INPUTS, base)
    return p if os.path.exists(p) else p + ".gz"


def read_bytes(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def records(path, fastq=True):
    """[(line0, line1, ...)] with 4 lines per FASTQ record, 2 per FASTA record (the reference reads
    records strictly by line count, normalise_kmers_multi_large.c:1572,1616-1627)."""
    lines = read_bytes(path).split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    n = 4 if fastq else 2
    return [tuple(lines[i:i + n]) for i in range(0, len(lines) - n + 1, n)]


def mask_n(seq):
    return seq.replace(b"N", b"A")


class Oracle:
    """ctypes wrapper of oracle/nkml_oracle.so (test infrastructure only)."""
    _lib = None
    EXACT, QUIRK = 0, 1

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "nkml_oracle.so"))
            vp, ci, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
            ip = ctypes.POINTER(ctypes.c_int)
            lib.nkml_oracle_create.restype = vp
            lib.nkml_oracle_create.argtypes = [ci, ci, ci, ci, ctypes.c_uint64]
            lib.nkml_oracle_destroy.argtypes = [vp]
            lib.nkml_oracle_read.argtypes = [vp, cp, ci, ip, ip]
            lib.nkml_oracle_pair.argtypes = [vp, cp, ci, cp, ci, ip, ip, ip, ip]
            lib.nkml_oracle_seed_read.argtypes = [vp, cp, ci]
            lib.nkml_oracle_batch.argtypes = [vp, vp, vp, ctypes.c_uint32, vp, vp]
            lib.nkml_oracle_keys.argtypes = [ci, ci, vp, vp, ctypes.c_uint32, vp]
            lib.nkml_oracle_count_keys.argtypes = [vp, vp, ctypes.c_uint64, vp]
            lib.nkml_oracle_count_keys.restype = None
            lib.nkml_oracle_unique.restype = ctypes.c_uint64
            lib.nkml_oracle_unique.argtypes = [vp]
            lib.nkml_oracle_count_of.restype = ctypes.c_uint32
            lib.nkml_oracle_count_of.argtypes = [vp, ctypes.c_uint64]
            lib.nkml_oracle_keep_pair.argtypes = [ci, ci, ci, ci, ctypes.c_float]
            lib.nkml_oracle_keep_single.argtypes = [ci, ci, ctypes.c_float]
            lib.nkml_oracle_encode.restype = ctypes.c_uint64
            lib.nkml_oracle_encode.argtypes = [cp, ci]
            lib.nkml_oracle_encode_canonical.restype = ctypes.c_uint64
            lib.nkml_oracle_encode_canonical.argtypes = [cp, ci]
            cls._lib = lib
        return cls._lib

    def __init__(self, k, depth, canonical=False, model=0, capacity=0):
        self.k, self.depth = k, depth
        self.h = self.lib().nkml_oracle_create(k, depth, int(bool(canonical)), model, capacity)
        if not self.h:
            raise MemoryError("nkml_oracle_create failed")

    def close(self):
        if self.h:
            self.lib().nkml_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def read(self, seq):
        h, t = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib().nkml_oracle_read(self.h, seq, len(seq), ctypes.byref(h), ctypes.byref(t))
        return rc, h.value, t.value

    def pair(self, fwd, rev):
        v = [ctypes.c_int(0) for _ in range(4)]
        rc = self.lib().nkml_oracle_pair(self.h, fwd, len(fwd), rev, len(rev), *[ctypes.byref(x) for x in v])
        return (rc,) + tuple(x.value for x in v)

    def seed(self, seq):
        return self.lib().nkml_oracle_seed_read(self.h, seq, len(seq))

    def batch(self, seq, offsets):
        """seq: uint8 array of concatenated reads, offsets: uint64[n+1] -> (high int32[n], total int32[n])."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        high = np.zeros(n, dtype=np.int32)
        total = np.zeros(n, dtype=np.int32)
        rc = self.lib().nkml_oracle_batch(self.h, seq.ctypes.data, offsets.ctypes.data, n,
                                          high.ctypes.data, total.ctypes.data)
        if rc != 0:
            raise ValueError("invalid base in batch")
        return high, total

    @classmethod
    def keys(cls, k, canonical, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        keys = np.zeros(len(seq), dtype=np.uint64)
        rc = cls.lib().nkml_oracle_keys(k, int(bool(canonical)), seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1, keys.ctypes.data)
        if rc != 0:
            raise ValueError("invalid base")
        return keys

    def count_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        flags = np.zeros(len(keys), dtype=np.uint8)
        self.lib().nkml_oracle_count_keys(self.h, keys.ctypes.data, len(keys), flags.ctypes.data)
        return flags

    @property
    def unique(self):
        return self.lib().nkml_oracle_unique(self.h)

    def count_of(self, code):
        return self.lib().nkml_oracle_count_of(self.h, code)

    @classmethod
    def keep_pair(cls, hf, tf, hr, tr, cov):
        return bool(cls.lib().nkml_oracle_keep_pair(hf, tf, hr, tr, cov))

    @classmethod
    def keep_single(cls, h, t, cov):
        return bool(cls.lib().nkml_oracle_keep_single(h, t, cov))

    @classmethod
    def encode(cls, kmer, canonical=False):
        f = cls.lib().nkml_oracle_encode_canonical if canonical else cls.lib().nkml_oracle_encode
        return f(kmer, len(kmer))


def parse_args(args):
    """The subset of the reference's command line used by golden.json -> dict."""
    cfg = dict(fwd=[], rev=[], k=15, depth=100, cov=0.9, canonical=False, single=False, fastq_in=True,
               fastq_out=True, memory=0)
    i = 0
    while i < len(args):
        a = args[i]
        if a in ("-f", "--forward", "-r", "--reverse"):
            key = "fwd" if a in ("-f", "--forward") else "rev"
            i += 1
            while i < len(args) and not args[i].startswith("-"):
                cfg[key].append(args[i])
                i += 1
            continue
        if a in ("-k", "--ksize"):
            cfg["k"] = int(args[i + 1]); i += 2; continue
        if a in ("-d", "--depth"):
            cfg["depth"] = int(args[i + 1]); i += 2; continue
        if a in ("-g", "--coverage"):
            cfg["cov"] = float(args[i + 1]); i += 2; continue
        if a in ("-m", "--memory_start"):
            cfg["memory"] = int(args[i + 1]); i += 2; continue
        if a in ("-t", "--filetype"):
            cfg["fastq_in"] = args[i + 1].lower() in ("fq", "fastq", "fsq"); i += 2; continue
        if a in ("-o", "--outformat"):
            cfg["fastq_out"] = args[i + 1].lower() in ("fq", "fastq", "fsq"); i += 2; continue
        if a in ("-c", "--canonical"):
            cfg["canonical"] = True; i += 1; continue
        if a in ("-s", "--single"):
            cfg["single"] = True; i += 1; continue
        raise ValueError("unhandled arg " + a)
    return cfg


def reference_capacity(k, memory_gb):
    """initial_hash_size of the reference at --cpu 1 (normalise_kmers_multi_large.c:416-422, 676-684)."""
    if memory_gb > 0:
        per = int(np.float32(np.float32(memory_gb) * np.float32(1073741824.0)) / np.float32(16))
        cap = per + 1 if per % 2 == 0 else per
    else:
        cap = 67108879
    return min(cap, 4 ** k)


def drive_oracle(cfg, model=0, seed=True):
    """Run the oracle the way the reference's main() runs its hot path at --cpu 1: seeding pass over every
    file (normalise_kmers_multi_large.c:2242-2250, 1322-1373), then every file pair / single file in order
    with one table for the whole run.  Returns per-processed-record arrays like tests/golden/make_golden.py."""
    k, depth, cov = cfg["k"], cfg["depth"], np.float32(cfg["cov"])
    o = Oracle(k, depth, cfg["canonical"], model, reference_capacity(k, cfg["memory"]) if model else 0)
    fwd = [records(input_path(p), cfg["fastq_in"]) for p in cfg["fwd"]]
    rev = [records(input_path(p), cfg["fastq_in"]) for p in cfg["rev"]]
    if seed:
        n_seed = 1 + int(3e6 / len(fwd))
        for i in range(len(fwd)):
            for recs in ([fwd[i]] + ([rev[i]] if i < len(rev) else [])):
                done = 0
                for r in recs:
                    if o.seed(r[1]) == 1:
                        done += 1
                        if done == n_seed:
                            break
    out = dict(kept=[], high_fwd=[], high_rev=[], total_fwd=[], total_rev=[], per_file=[])
    printed = skipped = 0
    for i in range(len(fwd)):
        if i < len(rev):
            for rf, rr in zip(fwd[i], rev[i]):
                rc, hf, tf, hr, tr = o.pair(rf[1], rr[1])
                assert rc >= 0
                if rc == 0:
                    continue
                keep = Oracle.keep_pair(hf, tf, hr, tr, cov)
                out["kept"].append(keep); out["high_fwd"].append(hf); out["high_rev"].append(hr)
                out["total_fwd"].append(tf); out["total_rev"].append(tr)
                printed += keep; skipped += (not keep)
        else:
            for rf in fwd[i]:
                rc, hf, tf = o.read(rf[1])
                assert rc >= 0
                if rc == 0:
                    continue
                keep = Oracle.keep_single(hf, tf, cov)
                out["kept"].append(keep); out["high_fwd"].append(hf); out["high_rev"].append(-1)
                out["total_fwd"].append(tf); out["total_rev"].append(-1)
                printed += keep; skipped += (not keep)
        out["per_file"].append([len(out["kept"]), printed, skipped, o.unique])
    res = {key: np.array(v, dtype=np.int32) for key, v in out.items() if key != "per_file"}
    res["kept"] = res["kept"].astype(np.uint8)
    res["per_file"] = out["per_file"]
    res["unique"] = o.unique
    o.close()
    return res
