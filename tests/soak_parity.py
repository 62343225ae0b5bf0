"""Randomised parity soak on the GPU (a checker run, not a product path): many small random configurations - k, depth,
canonical, read mix (random, low-complexity, tandem repeats, N runs, poly-G floods), random batch cuts - through the
C ABI on the partitioned and the atomic counting path, each compared read by read with the CPU oracle.
python tests/soak_parity.py [cases] [seed]   (not collected by pytest: a long run for the GPU box)"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import Oracle
from nomalise_kmers_multi_large_amd import binding
from nomalise_kmers_multi_large_amd.normaliser import KmerNormaliser

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_read(k, pool):
    kind = rng.integers(0, 7)
    n = int(rng.integers(k, 330))
    if kind == 0:
        r = ACGT[rng.integers(0, 4, n)]
    elif kind == 1:   # a piece of a shared "genome": repeats across reads, counts cross the depth
        at = int(rng.integers(0, len(pool) - n))
        r = pool[at:at + n].copy()
        if rng.random() < 0.3:
            r[rng.integers(0, n)] = ACGT[rng.integers(0, 4)]
    elif kind == 2:   # tandem repeat
        unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 7)))]
        r = np.tile(unit, n // len(unit) + 1)[:n]
    elif kind == 3:   # N runs
        r = ACGT[rng.integers(0, 4, n)]
        a = int(rng.integers(0, n)); r[a:a + int(rng.integers(1, 40))] = ord("N")
    elif kind == 4:   # homopolymer with a defect
        r = np.full(n, ACGT[rng.integers(0, 4)], dtype=np.uint8)
        r[rng.integers(0, n)] = ACGT[rng.integers(0, 4)]
    elif kind == 5:   # reverse complement of a genome piece
        at = int(rng.integers(0, len(pool) - n))
        comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
        r = comp[pool[at:at + n]][::-1].copy()
    else:             # exactly k long
        r = ACGT[rng.integers(0, 4, k)]
    return r.tobytes()


bad = 0
for case in range(cases):
    k = int(rng.choice([5, 7, 9, 11, 13, 14, 15, 16, 17, 21, 25, 31]))
    depth = int(rng.choice([2, 3, 5, 20, 100, 127, 300, 1000]))
    canonical = bool(rng.integers(0, 2))
    pool = ACGT[rng.integers(0, 4, int(rng.choice([400, 3000, 40000])))]
    reads = [make_read(k, pool) for _ in range(int(rng.integers(1, 700)))]
    o = Oracle(k, depth, canonical)
    want = [o.read(r) for r in reads]
    want_h, want_t = np.array([w[1] for w in want]), np.array([w[2] for w in want])
    modes = [binding.NKML_F_COUNT_ATOMIC, binding.NKML_F_COUNT_PARTITIONED] if k != 16 else [0]
    for flags in modes:
        # geometry of the hashed table's partitioned path: tiny buckets and LDS tables make it grow, cut buckets into
        # sub-ranges and (with a low limit) move to the atomic path in the middle of the stream
        for var in ("NKML_HP_LOG2_BUCKET", "NKML_HP_LOG2_LOCAL", "NKML_HP_LEAVE_SLOTS"):
            os.environ.pop(var, None)
        if k > 16 and flags == binding.NKML_F_COUNT_PARTITIONED:
            os.environ["NKML_HP_LOG2_BUCKET"] = str(int(rng.choice([3, 4, 5, 6])))
            if rng.random() < 0.7:
                os.environ["NKML_HP_LOG2_LOCAL"] = str(int(rng.choice([3, 4, 5, 6])))
            if rng.random() < 0.3:
                os.environ["NKML_HP_LEAVE_SLOTS"] = str(65536 << int(rng.choice([3, 4, 5])))
        g = KmerNormaliser(k, depth, 0.9, canonical, max_batch_reads=1024, max_batch_bases=1 << 19, flags=flags,
                           chunk_bases=int(rng.choice([0, 64, 4096])))
        at, hs, ts = 0, [], []
        while at < len(reads):
            n = int(rng.integers(1, 300))
            seq, off = KmerNormaliser._concat(reads[at:at + n])
            h, t = g.process_batch(seq, off)
            hs.append(h); ts.append(t); at += n
        h, t = np.concatenate(hs), np.concatenate(ts)
        ok = np.array_equal(h, want_h) and np.array_equal(t, want_t) and g.unique_kmers == o.unique
        if not ok:
            bad += 1
            print("MISMATCH case %d k=%d depth=%d canonical=%s flags=%d reads=%d: first read %s, unique %d vs %d" % (
                case, k, depth, canonical, flags, len(reads), np.nonzero((h != want_h) | (t != want_t))[0][:3], g.unique_kmers, o.unique), flush=True)
        g.close()
    if case % 25 == 24:
        print("%d cases done, %d mismatches" % (case + 1, bad), flush=True)
print("soak: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
