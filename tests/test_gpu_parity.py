"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the vectors captured from the
compiled reference.  Integer work: every comparison is bit-exact."""
import os

import numpy as np
import pytest

from tests.helpers import (Oracle, drive_oracle, golden_arrays, golden_meta, input_path, parse_args, records)

pytestmark = pytest.mark.gpu

META = golden_meta()
FIELDS = ("high_fwd", "high_rev", "total_fwd", "total_rev")


@pytest.fixture(scope="module")
def nk():
    import nomalise_kmers_multi_large_amd as m
    return m


def drive_gpu(nk, cfg, seed=True, **kw):
    """The reference's main() loop (seeding pass, then every file pair / single file in order on one table,
    normalise_kmers_multi_large.c:2242-2250, 2310-2393) with the HIP path doing the per-read work."""
    k, cov = cfg["k"], np.float32(cfg["cov"])
    fwd = [records(input_path(p), cfg["fastq_in"]) for p in cfg["fwd"]]
    rev = [records(input_path(p), cfg["fastq_in"]) for p in cfg["rev"]]
    out = dict(kept=[], high_fwd=[], high_rev=[], total_fwd=[], total_rev=[], per_file=[])
    with nk.KmerNormaliser(k, cfg["depth"], cfg["cov"], cfg["canonical"], **kw) as g:
        if seed:
            n_seed = 1 + int(3e6 / len(fwd))
            for i in range(len(fwd)):
                for recs in ([fwd[i]] + ([rev[i]] if i < len(rev) else [])):
                    reads = [r[1] for r in recs if len(r[1]) > k][:n_seed]
                    if reads:
                        g.seed_batch(*g._concat(reads))
        printed = skipped = 0
        for i in range(len(fwd)):
            if i < len(rev):
                n = min(len(fwd[i]), len(rev[i]))
                ok, hf, tf, hr, tr = g.process_sequence_pair([r[1] for r in fwd[i][:n]], [r[1] for r in rev[i][:n]])
                for j in np.nonzero(ok)[0]:
                    keep = g.keep_pair(hf[j], tf[j], hr[j], tr[j])
                    out["kept"].append(keep); out["high_fwd"].append(hf[j]); out["high_rev"].append(hr[j])
                    out["total_fwd"].append(tf[j]); out["total_rev"].append(tr[j])
                    printed += keep; skipped += (not keep)
            else:
                ok, hf, tf = g.process_sequence_single([r[1] for r in fwd[i]])
                for j in np.nonzero(ok)[0]:
                    keep = g.keep_single(hf[j], tf[j])
                    out["kept"].append(keep); out["high_fwd"].append(hf[j]); out["high_rev"].append(-1)
                    out["total_fwd"].append(tf[j]); out["total_rev"].append(-1)
                    printed += keep; skipped += (not keep)
            out["per_file"].append([len(out["kept"]), printed, skipped, g.unique_kmers])
        res = {key: np.array(v, dtype=np.int32) for key, v in out.items() if key != "per_file"}
        res["kept"] = res["kept"].astype(np.uint8)
        res["per_file"] = out["per_file"]
        res["unique"] = g.unique_kmers
        res["stats"] = g.stats()
    return res


@pytest.mark.parametrize("name", sorted(META))
def test_golden_cases(nk, oracle_lib, name):
    """Every captured reference run: the HIP path equals the exact-count oracle counter for counter, and equals
    the reference's own keep/skip decisions (and its counters where its table had no collisions)."""
    cfg = parse_args(META[name]["args"])
    gold = golden_arrays(name)
    want = drive_oracle(cfg, model=Oracle.EXACT)
    got = drive_gpu(nk, cfg)
    for f in ("kept",) + FIELDS:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    assert got["unique"] == want["unique"]
    assert got["per_file"] == want["per_file"]
    np.testing.assert_array_equal(got["kept"], gold["kept"])
    assert int(got["kept"].sum()) == META[name]["final"]["printed"]


def random_reads(rng, n, lo, hi, alphabet=b"ACGT", n_rate=0.0):
    reads = []
    al = np.frombuffer(alphabet, dtype=np.uint8)
    for _ in range(n):
        r = al[rng.integers(0, len(al), rng.integers(lo, hi + 1))].copy()
        if n_rate:
            r[rng.random(len(r)) < n_rate] = ord("N")
        reads.append(r.tobytes())
    return reads


@pytest.mark.parametrize("k,depth,canonical,chunk", [
    (5, 2, False, 0), (5, 3, True, 64), (7, 5, False, 1024), (9, 100, True, 4096), (15, 3, False, 64),
    (16, 4, True, 0), (17, 3, False, 128), (21, 2, True, 0), (25, 4, True, 1024), (31, 3, False, 64), (31, 50, True, 0),
])
def test_random_reads_small_keyspace(nk, oracle_lib, k, depth, canonical, chunk):
    """Low-complexity reads (2-3 letter alphabets, repeats) make the same k-mers recur inside reads and
    across chunks, so depth crossings land inside reads, at chunk edges and in the middle of batches."""
    rng = np.random.default_rng(1000 + k * 7 + depth)
    reads = random_reads(rng, 600, k, 4 * k + 40, b"AC") + random_reads(rng, 600, k, 200, b"ACGT", 0.01) \
        + random_reads(rng, 300, k, 90, b"GT") + [b"A" * 50, b"T" * 77, b"G" * 64, b"C" * 33, b"N" * 40] * 3
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    o = oracle_lib(k, depth, canonical)
    want = [o.read(r) for r in reads]
    assert all(w[0] == 1 for w in want)
    with nk.KmerNormaliser(k, depth, 0.9, canonical, chunk_bases=chunk, max_batch_reads=512, max_batch_bases=40000) as g:
        ok, high, total = g.process_sequence_single(reads)
        assert ok.all()
        np.testing.assert_array_equal(high, [w[1] for w in want])
        np.testing.assert_array_equal(total, [w[2] for w in want])
        assert g.unique_kmers == o.unique
        st = g.stats()
        assert st["kmers_counted"] == int(total.sum())


@pytest.mark.parametrize("k,depth,canonical,log2_bucket,per_batch", [
    (17, 3, False, None, 700), (21, 2, True, None, 512), (25, 4, True, None, 2000), (31, 3, False, None, 300), (31, 50, True, None, 700),
    (25, 100, True, None, 2000), (21, 200, False, None, 2000),
    (25, 4, True, 3, 2000),    # eight slots (one set) per bucket to begin with: the table doubles between batches and inside a batch
    (31, 3, False, 3, 150), (19, 5, True, 3, 5000),
    # a workgroup's LDS table of 8 (16, 32) slots: buckets of several sub-ranges, their keys regrouped by sub-range first
    # (2-8 sub-ranges: ballots; more: LDS counters), one bucket with more keys than 16-bit counters hold (the general kernel's)
    (25, 4, True, (3, 3), 2000), (31, 3, False, (3, 4), 5000), (21, 100, True, (3, 3), 5000), (23, 2, False, (3, 5), 300),
])
@pytest.mark.parametrize("on", [True, False])
def test_hashed_table_both_counting_paths(nk, oracle_lib, k, depth, canonical, log2_bucket, per_batch, on):
    """k > 16 through the partitioned path of the hashed table (buckets counted in LDS) and through the atomic path:
    per-read counters and distinct k-mers against the oracle, table dump against the oracle's counts."""
    if not on and log2_bucket is not None:
        pytest.skip("table geometry of the partitioned path only")
    log2_local = None
    if isinstance(log2_bucket, tuple):
        log2_bucket, log2_local = log2_bucket
    rng = np.random.default_rng(7000 + k * 7 + depth)
    reads = random_reads(rng, 600, k, 4 * k + 40, b"AC") + random_reads(rng, 900, k, 200, b"ACGT", 0.01) \
        + random_reads(rng, 300, k, 90, b"GT") + [b"A" * 50, b"T" * 77, b"G" * 64, b"C" * 33, b"N" * 40] * 3 \
        + [b"G" * 150] * (700 if log2_local else 400)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth, canonical)
    seeds = [r for r in reads[:100] if len(r) > k]
    for r in seeds:
        o.seed(r)
    want = [o.read(r) for r in reads]
    old = os.environ.get("NKML_HP_LOG2_BUCKET")
    old_local = os.environ.get("NKML_HP_LOG2_LOCAL")
    if log2_bucket is not None:
        os.environ["NKML_HP_LOG2_BUCKET"] = str(log2_bucket)
    if log2_local is not None:
        os.environ["NKML_HP_LOG2_LOCAL"] = str(log2_local)
    try:
        with partitioned_mode(on), nk.KmerNormaliser(k, depth, 0.9, canonical, max_batch_reads=per_batch, max_batch_bases=per_batch * 200) as g:
            st = g.stats()
            assert st["partitioned"] == (1 if on else 0)
            cap0 = st["capacity"]
            seq, off = nk.KmerNormaliser._concat(seeds)
            g.seed_batch(seq, off)
            ok, high, total = g.process_sequence_single(reads)
            np.testing.assert_array_equal(high, [w[1] for w in want])
            np.testing.assert_array_equal(total, [w[2] for w in want])
            assert g.unique_kmers == o.unique
            st = g.stats()
            assert st["kmers_counted"] == int(total.sum())
            if log2_bucket is not None:
                assert cap0 == 65536 << log2_bucket and st["resizes"] > 0 and st["capacity"] > cap0
            codes, counts, n = g.dump_table()
            assert n == o.unique
            # the partitioned path's counts stop growing at 32767, the atomic path freezes the counts of saturated k-mers
            for code, cnt in list(zip(codes.tolist(), counts.tolist()))[:3000]:
                true = o.count_of(code)
                assert cnt == true or (cnt >= depth - 1 and true >= cnt), (code, cnt, true)
    finally:
        for var, val in (("NKML_HP_LOG2_BUCKET", old), ("NKML_HP_LOG2_LOCAL", old_local)):
            if val is None:
                os.environ.pop(var, None)
            else:
                os.environ[var] = val


def test_hashed_table_moves_from_the_partitioned_to_the_atomic_path(nk, oracle_lib):
    """The hashed table starts on the partitioned path and, once it has outgrown the batches (here: after its first
    doubling, by NKML_HP_LEAVE_SLOTS), carries on on the atomic path with every k-mer and count it had: counters of every
    read, distinct k-mers and the dumped counts against the oracle, across the move."""
    k, depth = 25, 3
    rng = np.random.default_rng(4242)
    reads = random_reads(rng, 2600, 60, 200, b"ACGT", 0.01) + random_reads(rng, 400, k, 90, b"AC") + [b"G" * 150] * 50
    reads = [reads[i] for i in rng.permutation(len(reads))]
    reads = reads + reads[:700]          # seen again after the move: counts and saturation must have come along
    o = oracle_lib(k, depth, True)
    want = [o.read(r) for r in reads]
    saved = {v: os.environ.get(v) for v in ("NKML_HP_LOG2_BUCKET", "NKML_HP_LEAVE_SLOTS", "NKML_PARTITIONED")}
    os.environ["NKML_HP_LOG2_BUCKET"] = "3"
    os.environ["NKML_HP_LEAVE_SLOTS"] = str(65536 << 3)
    os.environ.pop("NKML_PARTITIONED", None)
    try:
        with nk.KmerNormaliser(k, depth, 0.9, True, max_batch_reads=400, max_batch_bases=400 * 200) as g:
            assert g.stats()["partitioned"] == 1 and g.stats()["capacity"] == 65536 << 3
            ok, high, total = g.process_sequence_single(reads)
            st = g.stats()
            assert st["partitioned"] == 0 and st["capacity"] > 65536 << 3
            np.testing.assert_array_equal(high, [w[1] for w in want])
            np.testing.assert_array_equal(total, [w[2] for w in want])
            assert g.unique_kmers == o.unique
            assert st["kmers_counted"] == int(total.sum())
            codes, counts, n = g.dump_table()
            assert n == o.unique
            for code, cnt in list(zip(codes.tolist(), counts.tolist()))[:3000]:
                true = o.count_of(code)
                assert cnt == true or (cnt >= depth - 1 and true >= cnt), (code, cnt, true)
    finally:
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old


def test_batch_cut_invariance(nk):
    """The stream may be cut into batches anywhere: one batch, many batches and one read per batch agree."""
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 400, 15, 120, b"ACG")
    res = []
    for max_reads, chunk in ((4096, 0), (37, 64), (1, 64)):
        with nk.KmerNormaliser(11, 3, 0.9, True, chunk_bases=chunk, max_batch_reads=max_reads, max_batch_bases=1 << 16) as g:
            res.append(g.process_sequence_single(reads)[1:] + (g.unique_kmers,))
    for r in res[1:]:
        np.testing.assert_array_equal(r[0], res[0][0])
        np.testing.assert_array_equal(r[1], res[0][1])
        assert r[2] == res[0][2]


def test_short_reads_and_bad_bases(nk):
    from nomalise_kmers_multi_large_amd import NkmlError
    with nk.KmerNormaliser(15, 2) as g:
        ok, hf, tf, hr, tr = g.process_sequence_pair([b"ACGTACGTACGTACGT", b"ACGT", b"ACGTACGTACGTACGT"],
                                                     [b"ACGTACGTACGTACGTA", b"ACGTACGTACGTACGTA", b"ACG"])
        assert ok.tolist() == [True, False, False]          # either mate < k: record not processed, nothing counted
        assert (tf[0], tr[0]) == (2, 3)
        assert g.unique_kmers == 3                            # the period-4 repeat has only 3 distinct 15-mers here
        with pytest.raises(NkmlError) as e:                   # the reference exits on a non-ACGT base
            g.process_sequence_single([b"ACGTACGTACGTACGTx"])
        assert e.value.code == -4
        with pytest.raises(NkmlError):
            g.process_batch(np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), np.array([0, 10], dtype=np.uint64))
        # N is read as A: 6 all-A windows are skipped, the one ending in C counts
        ok, h, t = g.process_sequence_single([b"N" * 20 + b"C"])
        assert (int(h[0]), int(t[0])) == (0, 1)


def test_keep_skip_edges(nk, oracle_lib):
    with nk.KmerNormaliser(15, 2, 0.9) as g:
        assert g.keep_single(0, 0) and not g.keep_single(9, 10) and g.keep_single(8, 10)
        assert not g.keep_pair(0, 10, 10, 10) and g.keep_pair(8, 10, 0, 0)
    rng = np.random.default_rng(11)
    import torch
    tot = rng.integers(0, 140, 20000).astype(np.int32)
    hi = (rng.random(20000) * (tot + 1)).astype(np.int32)
    for cov in (0.9, 0.96, 0.5, 0.001, 1.0):
        with nk.KmerNormaliser(15, 2, cov) as g:
            d_h = torch.from_numpy(hi).cuda(); d_t = torch.from_numpy(tot).cuda()
            keep_p = torch.zeros(10000, dtype=torch.uint8, device="cuda")
            keep_s = torch.zeros(20000, dtype=torch.uint8, device="cuda")
            g.decide_device(d_h.data_ptr(), d_t.data_ptr(), 10000, True, keep_p.data_ptr())
            g.decide_device(d_h.data_ptr(), d_t.data_ptr(), 20000, False, keep_s.data_ptr())
            g.synchronize()
            c = np.float32(cov)
            want_s = [oracle_lib.keep_single(int(h), int(t), c) for h, t in zip(hi, tot)]
            want_p = [oracle_lib.keep_pair(int(hi[2 * i]), int(tot[2 * i]), int(hi[2 * i + 1]), int(tot[2 * i + 1]), c)
                      for i in range(10000)]
            np.testing.assert_array_equal(keep_s.cpu().numpy().astype(bool), want_s)
            np.testing.assert_array_equal(keep_p.cpu().numpy().astype(bool), want_p)
            assert [g.keep_single(int(h), int(t)) for h, t in zip(hi[:2000], tot[:2000])] == want_s[:2000]


def test_synthetic_pairs_device_path(nk, oracle_lib):
    """150 bp synthetic pairs through the device-buffer entry point, in three batches, against the oracle;
    also the staged entry points (kmerise -> count -> reduce) used by the sharded path give the same numbers."""
    import torch
    from nomalise_kmers_multi_large_amd import synth
    n = 60000
    blocks = list(synth.pair_blocks(n, seed=3, transcripts=60, block=20000))
    o = oracle_lib(15, 100)
    with nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=40000, max_batch_bases=40000 * 150, chunk_bases=1 << 20) as g, \
            nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=40000, max_batch_bases=40000 * 150) as g2:
        for c1, c2 in blocks:
            seq, off = synth.interleave(c1, c2)
            want_h, want_t = o.batch(seq, off)
            d_seq = torch.from_numpy(seq).cuda(); d_off = torch.from_numpy(off.astype(np.int64)).cuda()
            nr = len(off) - 1
            d_h = torch.zeros(nr, dtype=torch.int32, device="cuda"); d_t = torch.zeros_like(d_h)
            torch.cuda.synchronize()
            g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), d_h.data_ptr(), d_t.data_ptr())
            g.synchronize()
            np.testing.assert_array_equal(d_h.cpu().numpy(), want_h)
            np.testing.assert_array_equal(d_t.cpu().numpy(), want_t)
            npos = (len(seq) + 15) // 16 * 16
            d_keys = torch.zeros(npos, dtype=torch.int32, device="cuda")
            d_flags = torch.zeros(npos, dtype=torch.uint8, device="cuda")
            d_h2 = torch.zeros_like(d_h); d_t2 = torch.zeros_like(d_h)
            torch.cuda.synchronize()
            g2.kmerise_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), d_keys.data_ptr())
            g2.count_keys_device(d_keys.data_ptr(), npos, d_flags.data_ptr())
            g2.reduce_reads_device(d_keys.data_ptr(), d_flags.data_ptr(), d_off.data_ptr(), nr, d_h2.data_ptr(), d_t2.data_ptr())
            g2.synchronize()
            np.testing.assert_array_equal(d_h2.cpu().numpy(), want_h)
            np.testing.assert_array_equal(d_t2.cpu().numpy(), want_t)
        assert g.unique_kmers == o.unique == g2.unique_kmers
        st = g.stats()
        assert st["crossing_kmers"] > 0 and st["kmers_saturated"] > 0  # both branches of the counting pass ran


def test_table_dump_matches_oracle_counts(nk, oracle_lib):
    rng = np.random.default_rng(9)
    reads = random_reads(rng, 300, 20, 100, b"ACGT")
    for k in (13, 19):
        o = oracle_lib(k, 1000)
        for r in reads:
            o.read(r)
        with nk.KmerNormaliser(k, 1000) as g:   # depth never reached: counts are exact occurrence counts
            g.process_sequence_single(reads)
            codes, counts, n = g.dump_table()
            assert n == o.unique == len(codes)
            assert all(o.count_of(int(c)) == int(v) for c, v in zip(codes[:5000], counts[:5000]))


def test_sharded_stages_on_one_gpu(nk, oracle_lib):
    """The device stages of the k-mer-sharded path (k-merise, stable routing by owner, count at the owner, flags
    back to their positions, per-read reduce) with three shards emulated in one process on one GPU: the
    all-to-all is replaced by slicing, everything else is the product code.  Equals the single-stream oracle."""
    import torch
    from nomalise_kmers_multi_large_amd.sharded import HipShardEngine
    W = 3
    for k, depth, canonical in ((13, 3, True), (23, 2, False)):
        engines = [HipShardEngine(k, depth, 0.9, canonical, 0, max_reads=4096, max_bases=1 << 18, slack=3.0) for _ in range(W)]
        o = oracle_lib(k, depth, canonical)
        rng = np.random.default_rng(77 + k)
        for step in range(3):
            per_rank = [random_reads(rng, 300 + 50 * r, k, 120, b"ACG") for r in range(W)]
            routed = []
            for r in range(W):
                seq, off = engines[r].nk._concat(per_rank[r])
                d_seq = torch.from_numpy(seq.copy()).cuda()
                d_off = torch.from_numpy(off.astype(np.int64)).cuda()
                keys = engines[r].kmerise(d_seq, d_off, len(per_rank[r]), len(seq))
                sk, ss, counts = engines[r].route(keys, W)
                assert sum(counts) == int((keys != 0).sum())
                routed.append((d_off, keys.numel(), sk.clone(), ss.clone(), counts))
            back = [[None] * W for _ in range(W)]
            for owner in range(W):          # "all-to-all": owner receives, in source order, each source's slice
                parts = []
                for src in range(W):
                    c = routed[src][4]
                    a = sum(c[:owner])
                    parts.append(routed[src][2][a:a + c[owner]])
                recv = engines[owner].recv_buffer(sum(p.numel() for p in parts))
                recv.copy_(torch.cat(parts))
                fl = engines[owner].count(recv).clone()
                a = 0
                for src in range(W):
                    n = parts[src].numel()
                    back[src][owner] = fl[a:a + n]
                    a += n
            for r in range(W):
                d_off, npos, sk, ss, counts = routed[r]
                bf = engines[r].back_buffer(sum(counts))
                bf.copy_(torch.cat(back[r]))
                flags = engines[r].unroute(bf, ss, npos)
                high, total = engines[r].reduce(flags, d_off, len(per_rank[r]))
                engines[r].synchronize()
                want = [o.read(x) for x in per_rank[r]]
                np.testing.assert_array_equal(high.cpu().numpy(), [w[1] for w in want])
                np.testing.assert_array_equal(total.cpu().numpy(), [w[2] for w in want])
        assert sum(e.stats()["unique_kmers"] for e in engines) == o.unique
        for e in engines:
            e.nk.close()


# ---- the partitioned counting path (k <= 15, depth <= 127; NKML_PARTITIONED=1) against the same checks -----------
class partitioned_mode:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        import os
        self.old = os.environ.get("NKML_PARTITIONED")
        os.environ["NKML_PARTITIONED"] = "1" if self.on else "0"

    def __exit__(self, *exc):
        import os
        if self.old is None:
            os.environ.pop("NKML_PARTITIONED", None)
        else:
            os.environ["NKML_PARTITIONED"] = self.old


@pytest.mark.parametrize("k,depth,canonical", [(5, 2, False), (7, 5, True), (9, 100, True), (9, 127, False), (13, 3, False), (15, 3, False),
                                               (15, 100, True), (9, 128, False), (7, 255, True), (5, 1000, False), (13, 200, False)])
def test_partitioned_random_reads(nk, oracle_lib, k, depth, canonical):
    rng = np.random.default_rng(4000 + k * 7 + depth)
    reads = random_reads(rng, 600, k, 4 * k + 40, b"AC") + random_reads(rng, 600, k, 200, b"ACGT", 0.01) \
        + random_reads(rng, 300, k, 90, b"GT") + [b"A" * 50, b"T" * 77, b"G" * 64, b"C" * 33, b"N" * 40] * 3
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth, canonical)
    want = [o.read(r) for r in reads]
    with partitioned_mode(True), nk.KmerNormaliser(k, depth, 0.9, canonical, max_batch_reads=700, max_batch_bases=60000) as g:
        ok, high, total = g.process_sequence_single(reads)
        np.testing.assert_array_equal(high, [w[1] for w in want])
        np.testing.assert_array_equal(total, [w[2] for w in want])
        assert g.unique_kmers == o.unique
        st = g.stats()
        assert st["kmers_counted"] == int(total.sum())
        # one byte per k-mer up to depth 127, two beyond: the partitioned path is the one that ran
        assert st["partitioned"] == 1 and st["table_bytes"] == st["capacity"] * (1 if depth <= 127 else 2)


@pytest.mark.parametrize("n_reads,per_batch", [(300, 400), (700, 400), (1200, 2000)])
def test_partitioned_largest_depth(nk, oracle_lib, n_reads, per_batch):
    """depth 16383, the largest the partitioned path takes (two-byte table entries): two k-mers cross it in the middle of
    a batch; with 1200 reads in one batch their bucket holds more than 65535 keys (the other bucket kernel)."""
    k, depth = 5, 16383
    rng = np.random.default_rng(n_reads)
    reads = [b"AC" * 75] * n_reads + random_reads(rng, 50, k, 120, b"ACGT")
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth)
    want = [o.read(r) for r in reads]
    with partitioned_mode(True), nk.KmerNormaliser(k, depth, 0.9, max_batch_reads=per_batch, max_batch_bases=per_batch * 150) as g:
        ok, high, total = g.process_sequence_single(reads)
        np.testing.assert_array_equal(high, [w[1] for w in want])
        np.testing.assert_array_equal(total, [w[2] for w in want])
        assert g.unique_kmers == o.unique and g.stats()["partitioned"] == 1
        assert int(high.sum()) > 0


@pytest.mark.parametrize("on", [False, True])
def test_heavy_bucket_one_kmer_many_times(nk, oracle_lib, on):
    """Low-complexity flood: one k-mer hundreds of thousands of times in a batch (more than 16-bit counters hold),
    its threshold crossed in the middle of a read, plus near-copies landing in the same table bucket."""
    k, depth = 15, 100
    reads = [b"G" * 150] * 1500 + [b"G" * 120 + b"C" + b"G" * 29] * 300 + [b"G" * 40 + b"T" + b"G" * 60] * 200 + [b"ACGT" * 30] * 50
    rng = np.random.default_rng(8)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth)
    want = [o.read(r) for r in reads]
    with partitioned_mode(on), nk.KmerNormaliser(k, depth, 0.9, max_batch_reads=4096, max_batch_bases=1 << 20) as g:
        ok, high, total = g.process_sequence_single(reads)
        np.testing.assert_array_equal(high, [w[1] for w in want])
        np.testing.assert_array_equal(total, [w[2] for w in want])
        assert g.unique_kmers == o.unique


def test_partitioned_equals_atomic_path_on_synthetic_pairs(nk):
    """Both counting paths on the same 150 bp synthetic pairs in several batches: identical per-read counters."""
    from nomalise_kmers_multi_large_amd import synth
    res = []
    for on in (False, True):
        with partitioned_mode(on), nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=100000, max_batch_bases=100000 * 150) as g:
            out = []
            for c1, c2 in synth.pair_blocks(150000, seed=5, transcripts=100, block=50000):
                out.append(g.process_batch(*synth.interleave(c1, c2)))
            res.append((np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), g.unique_kmers))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]


def test_sharded_normaliser_over_rccl_world_of_one(nk):
    """ShardedNormaliser end to end over the real RCCL backend (world size 1 on this box): k=15, so the owner counts
    with the partitioned path from a key array; the exchange is forced into several rounds per step.  Per-read
    counters must equal those of the single-table path on the same stream."""
    import socket
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.sharded import HipShardEngine, ShardedNormaliser
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bp = 20000
        eng = HipShardEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, stay_home=True)
        sn = ShardedNormaliser(eng, max_message_bytes=1 << 20)
        with nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
            h = torch.zeros(2 * bp, dtype=torch.int32, device="cuda")
            t = torch.zeros_like(h)
            host = [synth.interleave(c1, c2) for c1, c2 in synth.pair_blocks(4 * bp, seed=5, transcripts=40, block=bp)]
            dev = [(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)) for seq, off in host]
            for i, (d_seq, d_off, nr, nb) in enumerate(dev):
                # steps 0 and 1 announce their successor (its keys travel while this step's are counted), step 2 does not
                high, total, gathered = sn.process_step(d_seq, d_off, nr, nb, then=dev[i + 1] if i < 2 else None)
                g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, nb, h.data_ptr(), t.data_ptr())
                sn.synchronize()
                g.synchronize()
                np.testing.assert_array_equal(high.cpu().numpy(), h[:nr].cpu().numpy())
                np.testing.assert_array_equal(total.cpu().numpy(), t[:nr].cpu().numpy())
                np.testing.assert_array_equal(gathered[0][0].cpu().numpy(), h[:nr].cpu().numpy())
            assert int(h.sum()) > 0 and eng.stats()["unique_kmers"] == g.unique_kmers
            assert eng.ranked and eng.stay_home and eng.stats()["kmers_counted"] == g.stats()["kmers_counted"]
    finally:
        dist.destroy_process_group()


def test_k25_decisions_against_the_reference_at_scale(nk):
    """k=25 --canonical on 2 M synthetic pairs against the decisions of the reference program itself (recorded by
    tests/refdiff_k25.py, one bitmap per reference table size).  The reference's table is never collision-free at k > 16, so
    a few decisions at the threshold differ from exact counting: 348 of 2 M with its default table; fewer the larger its
    table, which is what pins the differences on its collisions.  The product runs with --memory_start 4 (a k > 16 context
    with memory_gb > 0) and grows its table from there."""
    from tests import refdiff_k25
    maps = refdiff_k25.load_bitmaps()
    assert maps, "tests/golden/refdiff_k25_*.npz missing"
    keep, unique = refdiff_k25.product_keep(maps[0]["pairs"])
    assert int(keep.sum()) == 883957 and unique == 33851135   # the exact model's answer (CPU oracle)
    diffs = {}
    for m in maps:
        assert m["pairs"] == maps[0]["pairs"]
        diffs[m["memory_gb"]] = int((keep != m["keep"]).sum())
    assert diffs[0] == 348 and m["printed"] > 0
    for gb, d in diffs.items():
        if gb:
            assert d < diffs[0], diffs   # a larger reference table collides less and disagrees less


@pytest.mark.parametrize("k,canonical,batch_pairs,want_kept,want_unique", [
    (15, False, 2000000, 1011077, 28393181),   # one batch: 544 M k-mers through the partitioned path at its working size
    (15, False, 700000, 1011077, 28393181),    # ragged batches: the answer does not depend on the cuts
    (25, True, 1000000, 883957, 33851135),     # hashed table, canonical k-mers, growth from the default capacity
])
def test_known_answer_two_million_synthetic_pairs(nk, k, canonical, batch_pairs, want_kept, want_unique):
    """The 2 M-pair synthetic dataset (seed 1, 2000 transcripts) whose answer the reference gives with a
    collision-free table (-m 16 at k=15) and the CPU oracle reproduces: kept pairs and distinct k-mers."""
    import torch
    from nomalise_kmers_multi_large_amd import synth
    n = 2000000
    kept = 0
    with nk.KmerNormaliser(k, 100, 0.9, canonical, max_batch_reads=2 * batch_pairs, max_batch_bases=2 * batch_pairs * 150) as g:
        h = torch.zeros(2 * batch_pairs, dtype=torch.int32, device="cuda")
        t = torch.zeros_like(h)
        keep = torch.zeros(batch_pairs, dtype=torch.uint8, device="cuda")
        # the dataset is defined with the generator's default block size (its random stream depends on it)
        blocks = list(synth.pair_blocks(n, seed=1, transcripts=2000))
        a1, a2 = np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks])
        del blocks
        for at in range(0, n, batch_pairs):
            seq, off = synth.interleave(a1[at:at + batch_pairs], a2[at:at + batch_pairs])
            d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
            nr = len(off) - 1
            g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), h.data_ptr(), t.data_ptr())
            g.decide_device(h.data_ptr(), t.data_ptr(), nr // 2, True, keep.data_ptr())
            g.synchronize()
            kept += int(keep[: nr // 2].sum())
        assert kept == want_kept
        assert g.unique_kmers == want_unique


def _two_rank_worker(rank, world, port, steps, bp, q, ranked=None, stay_home=None):
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.sharded import HipShardEngine, ShardedNormaliser
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = HipShardEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, slack=2.0, world=world, ranked=ranked, stay_home=stay_home)
        assert eng.ranked == (ranked is not False) and eng.stay_home == bool(ranked is not False and stay_home)
        sn = ShardedNormaliser(eng, stage_through_host=True, max_message_bytes=4 << 20)
        blocks = list(synth.pair_blocks(steps * bp, seed=11 + rank, transcripts=30, block=bp))
        out, batches = [], []
        for c1, c2 in blocks:
            seq, off = synth.interleave(c1, c2)
            batches.append((torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)))
        for i, b in enumerate(batches):   # the first step announces the second (keys sent ahead), the others do not
            high, total, gathered = sn.process_step(*b, then=batches[1] if i == 0 else None)
            sn.synchronize()
            if rank == 0:
                out.append([g.cpu().numpy().copy() for g in gathered])
        st = eng.stats()
        tallies = torch.tensor([st["kmers_counted"], st["kmers_saturated"], st["unique_kmers"], eng.home_total], dtype=torch.int64)
        dist.all_reduce(tallies)
        if rank == 0:
            q.put((out, tallies.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ranked,stay_home", [(2, False, False), (2, True, False), (2, True, True), (3, True, True)])
def test_two_ranks_share_one_gpu_equal_single_table(nk, world, ranked, stay_home):
    """Two (three) ranks with the product engine (each its own context and table shard on this GPU), exchanging over gloo
    through host memory: the per-read counters gathered on rank 0 equal those of one table reading the global
    stream (per step: rank 0's reads, then rank 1's, ...).  With source positions per sent key, with ranked routing, and with
    saturated k-mers staying home (then fewer occurrences reach a table than were read, and the statistics still add up)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from nomalise_kmers_multi_large_amd import synth
    steps, bp = 3, 15000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, steps, bp, q, ranked, stay_home)) for r in range(world)]
    for p in procs:
        p.start()
    got, tallies = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    data = [list(synth.pair_blocks(steps * bp, seed=11 + r, transcripts=30, block=bp)) for r in range(world)]
    with nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
        h = torch.zeros(2 * bp, dtype=torch.int32, device="cuda")
        t = torch.zeros_like(h)
        crossed = 0
        for step in range(steps):
            for r in range(world):
                seq, off = synth.interleave(*data[r][step])
                d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
                nr = len(off) - 1
                g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), h.data_ptr(), t.data_ptr())
                g.synchronize()
                np.testing.assert_array_equal(got[step][r][0], h[:nr].cpu().numpy())
                np.testing.assert_array_equal(got[step][r][1], t[:nr].cpu().numpy())
                crossed += int(h[:nr].sum())
        assert crossed > 0
        st = g.stats()
        # summed over the ranks: every occurrence read is accounted for, whether it reached a table or stayed home
        # (the "saturated" tally depends on where the batches are cut - an occurrence counts if its k-mer was at depth - 1 when
        # its batch began, and a step here is one batch of all ranks' reads - so it is only close to the single table's)
        assert tallies[0] == st["kmers_counted"] and tallies[2] == st["unique_kmers"]
        assert 0 < tallies[1] <= st["kmers_saturated"] and tallies[1] > 0.99 * st["kmers_saturated"]
        # with the bitmap a good part of the saturated occurrences never left the rank that read them
        assert (tallies[3] > 0.3 * st["kmers_saturated"]) if (ranked and stay_home) else (tallies[3] == 0)


def _bucket_rank_worker(rank, world, port, steps, bp, k, depth, stay_home, q, list_room=None):
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.bucketed import BucketedNormaliser, HipBucketEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if list_room:   # a list of newly saturated k-mers that holds next to nothing: what does not fit is never announced
            HipBucketEngine.MAX_NEW_SATURATED = list_room
        eng = HipBucketEngine(k, depth, 0.9, False, 0, 2 * bp, 2 * bp * 150, world, rank, slack=2.0, stay_home=stay_home)
        sn = BucketedNormaliser(eng, stage_through_host=True, max_message_bytes=1 << 20)
        out, batches = [], []
        for step, (c1, c2) in enumerate(synth.pair_blocks(steps * bp, seed=11 + rank, transcripts=30, block=bp)):
            if rank == world - 1 and step == 1:   # the last rank brings nothing to the second step
                seq, off = np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64)
            else:
                seq, off = synth.interleave(c1, c2)
            batches.append((torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)))
        for step, batch in enumerate(batches):
            # the first step announces the second: that one is partitioned and its k-mers travel before the first is counted
            high, total, gathered = sn.process_step(*batch, then=batches[1] if step == 0 else None)
            sn.synchronize()
            if rank == 0:
                out.append([g.cpu().numpy().copy() for g in gathered])
        st = eng.stats()
        tallies = torch.tensor([st["kmers_counted"], st["unique_kmers"], eng.home_total], dtype=torch.int64)
        dist.all_reduce(tallies)
        if rank == 0:
            q.put((out, tallies.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,depth,stay_home,list_room", [(2, 15, 100, True, None), (3, 15, 100, True, None), (2, 15, 200, False, None),
                                                                (3, 13, 20, True, None), (4, 14, 100, True, None), (2, 15, 100, True, 50)])
def test_bucket_ordered_exchange_ranks_share_one_gpu_equal_single_table(nk, world, k, depth, stay_home, list_room):
    """The bucket-ordered exchange with the product engine: two to four ranks (each its own context and its share of the table's
    buckets on this GPU) exchanging over gloo through host memory.  The per-read counters gathered on rank 0 equal those of
    one table reading the global stream (per step: rank 0's reads, then rank 1's, ...); one rank brings nothing to one step;
    with the saturation bitmap a good part of the occurrences never leaves the rank that read them, and every occurrence
    is still accounted for.  One-byte and two-byte table entries, several key-space sizes; in the last case the list of newly
    saturated k-mers holds 50 per step, so most are never announced to the other rank and keep travelling - same results."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from nomalise_kmers_multi_large_amd import synth
    steps, bp = 3, 12000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_rank_worker, args=(r, world, port, steps, bp, k, depth, stay_home, q, list_room)) for r in range(world)]
    for p in procs:
        p.start()
    got, tallies = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    data = [list(synth.pair_blocks(steps * bp, seed=11 + r, transcripts=30, block=bp)) for r in range(world)]
    with nk.KmerNormaliser(k, depth, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
        h = torch.zeros(2 * bp, dtype=torch.int32, device="cuda")
        t = torch.zeros_like(h)
        crossed = 0
        for step in range(steps):
            for r in range(world):
                if r == world - 1 and step == 1:
                    assert got[step][r].shape[1] == 0
                    continue
                seq, off = synth.interleave(*data[r][step])
                d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
                nr = len(off) - 1
                g.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, len(seq), h.data_ptr(), t.data_ptr())
                g.synchronize()
                np.testing.assert_array_equal(got[step][r][0], h[:nr].cpu().numpy())
                np.testing.assert_array_equal(got[step][r][1], t[:nr].cpu().numpy())
                crossed += int(h[:nr].sum())
        assert crossed > 0
        st = g.stats()
        assert tallies[0] == st["kmers_counted"] and tallies[1] == st["unique_kmers"]
        if list_room:   # (each rank still knows its own buckets' saturated k-mers: some occurrences stay home)
            assert 0 < tallies[2] < 0.9 * st["kmers_saturated"]
        else:
            assert (tallies[2] > 0.3 * st["kmers_saturated"]) if stay_home else (tallies[2] == 0)


@pytest.mark.parametrize("k", [14, 15])
def test_partitioned_equals_atomic_on_odd_batch_sizes(nk, k):
    """Batches of 1, 13, 14, 27, 28, 55 and 1000 pairs in a row (tile and word boundaries of the partition passes fall
    inside and between them): the partitioned path gives the per-read counters of the atomic path."""
    import torch
    from nomalise_kmers_multi_large_amd import synth
    sizes = [1, 13, 14, 27, 28, 55, 1000, 1, 2731]
    c1, c2 = next(synth.pair_blocks(sum(sizes), seed=21, transcripts=3, block=sum(sizes)))
    got = {}
    for on in (False, True):
        with partitioned_mode(on), nk.KmerNormaliser(k, 7, 0.9, max_batch_reads=8192, max_batch_bases=8192 * 150) as g:
            out, at = [], 0
            for n in sizes:
                seq, off = synth.interleave(c1[at:at + n], c2[at:at + n])
                at += n
                out.append(g.process_batch(seq, off))
            got[on] = (out, g.unique_kmers, g.stats()["partitioned"])
    assert got[True][2] == 1 and got[False][2] == 0
    assert got[True][1] == got[False][1]
    for (ha, ta), (hp, tp) in zip(got[False][0], got[True][0]):
        np.testing.assert_array_equal(hp, ha)
        np.testing.assert_array_equal(tp, ta)
    assert sum(int(h.sum()) for h, _ in got[True][0]) > 0


def test_table_shards_add_up_to_the_single_table(nk):
    """Three contexts on one GPU, each counting only its table shard of the same batches: per-read high and total
    summed over the shards, and the shards' distinct k-mers, equal the unsharded context's."""
    from nomalise_kmers_multi_large_amd import binding, synth
    n_shards, bp = 3, 20000
    mk = lambda: nk.KmerNormaliser(15, 20, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150, flags=binding.NKML_F_COUNT_PARTITIONED)
    whole, parts = mk(), [mk() for _ in range(n_shards)]
    try:
        for s, g in enumerate(parts):
            g.set_shard(s, n_shards)
        seen = 0
        for c1, c2 in synth.pair_blocks(3 * bp, seed=9, transcripts=25, block=bp):
            seq, off = synth.interleave(c1, c2)
            h, t = whole.process_batch(seq, off)
            got = [g.process_batch(seq, off) for g in parts]
            np.testing.assert_array_equal(sum(x[0] for x in got), h)
            np.testing.assert_array_equal(sum(x[1] for x in got), t)
            assert all(x[1].sum() > 0.2 * t.sum() for x in got)   # every shard carries its share
            seen += int(h.sum())
        assert seen > 0
        assert sum(g.unique_kmers for g in parts) == whole.unique_kmers
        with pytest.raises(binding.NkmlError):
            parts[0].set_shard(3, 3)
    finally:
        for g in [whole] + parts:
            g.close()
    with nk.KmerNormaliser(25, 20, 0.9, True, max_batch_reads=64, max_batch_bases=6400) as g:   # hashed table: atomic path
        with pytest.raises(binding.NkmlError):
            g.set_shard(0, 2)
        g.set_shard(0, 1)


@pytest.mark.parametrize("k,canonical", [(15, False), (14, True), (25, True), (11, False)])
def test_packed_input_equals_sequence_input(nk, k, canonical):
    """nkml_pack_device + nkml_process_packed_device against nkml_process_device on the same batches (reads with N,
    lengths that are no multiple of 16): the same counters, and the packed words are the 2-bit codes, first base on top."""
    import torch
    rng = np.random.default_rng(k)
    a, b = (nk.KmerNormaliser(k, 3, 0.9, canonical, max_batch_reads=4096, max_batch_bases=1 << 20) for _ in range(2))
    try:
        for step in range(3):
            reads = random_reads(rng, 900 + step, k, 211, n_rate=0.01)
            seq, off = nk.KmerNormaliser._concat(reads)
            nr, nb = len(reads), len(seq)
            d_seq, d_off = torch.from_numpy(seq.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
            words = torch.full(((nb + 15) // 16,), -1, dtype=torch.int32, device="cuda")
            out = torch.zeros((4, nr), dtype=torch.int32, device="cuda")
            a.process_device(d_seq.data_ptr(), d_off.data_ptr(), nr, nb, out[0].data_ptr(), out[1].data_ptr())
            b.pack_device(d_seq.data_ptr(), nb, words.data_ptr())
            b.process_packed_device(words.data_ptr(), d_off.data_ptr(), nr, nb, out[2].data_ptr(), out[3].data_ptr())
            a.synchronize()
            b.synchronize()
            o = out.cpu().numpy()
            np.testing.assert_array_equal(o[2], o[0])
            np.testing.assert_array_equal(o[3], o[1])
            assert o[1].sum() > 0
            code = np.zeros(256, dtype=np.uint32)
            code[list(b"ACGT")] = [0, 1, 2, 3]
            padded = np.zeros(len(words) * 16, dtype=np.uint32)
            padded[:nb] = code[seq]
            want = (padded.reshape(-1, 16) << np.arange(30, -2, -2, dtype=np.uint32)).sum(axis=1, dtype=np.uint64).astype(np.uint32)
            np.testing.assert_array_equal(words.cpu().numpy().view(np.uint32), want)
        assert a.unique_kmers == b.unique_kmers
    finally:
        a.close()
        b.close()


def _replica_worker(rank, world, port, steps, bp, q):
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.replicated import HipReplicaEngine, ReplicatedReadsNormaliser
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = HipReplicaEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, world=world, rank=rank)
        rn = ReplicatedReadsNormaliser(eng, stage_through_host=True)
        # rank 1 brings a third less per step, so its region ends in filler
        mine = bp if rank == 0 else bp * 2 // 3
        blocks = list(synth.pair_blocks(steps * mine, seed=11 + rank, transcripts=30, block=mine))
        out, batches = [], []
        for c1, c2 in blocks:
            seq, off = synth.interleave(c1, c2)
            batches.append((torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)))
        for i, b in enumerate(batches):   # every step but the last announces the next one
            high, total, gathered = rn.process_step(*b, then=batches[i + 1] if i + 1 < len(batches) else None)
            rn.synchronize()
            if rank == 0:
                out.append([g.cpu().numpy().copy() for g in gathered])
        uniq = torch.tensor([rn.stats()["unique_kmers"]], dtype=torch.int64)
        dist.all_reduce(uniq)
        if rank == 0:
            q.put((out, int(uniq.item())))
    finally:
        dist.destroy_process_group()


def test_two_replica_ranks_share_one_gpu_equal_single_table(nk):
    """The replicated-reads mode with the product engine: two ranks (each its own context and table shard on this GPU),
    collectives over gloo through host memory.  The counters on rank 0 equal those of one table reading the global stream."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from nomalise_kmers_multi_large_amd import synth
    world, steps, bp = 2, 3, 15000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, steps, bp, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, uniq = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sizes = [bp, bp * 2 // 3]
    data = [list(synth.pair_blocks(steps * sizes[r], seed=11 + r, transcripts=30, block=sizes[r])) for r in range(world)]
    with nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
        crossed = 0
        for step in range(steps):
            for r in range(world):
                h, t = g.process_batch(*synth.interleave(*data[r][step]))
                np.testing.assert_array_equal(got[step][r][0], h)
                np.testing.assert_array_equal(got[step][r][1], t)
                crossed += int(h.sum())
        assert crossed > 0 and uniq == g.unique_kmers


def test_replicated_normaliser_over_rccl_world_of_one(nk):
    """ReplicatedReadsNormaliser end to end over the real RCCL backend (world size 1 on this box): all-gather and
    sum are the real collectives; counters equal the plain single-table path."""
    import socket
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.replicated import HipReplicaEngine, ReplicatedReadsNormaliser
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bp = 20000
        eng = HipReplicaEngine(15, 100, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, world=1, rank=0)
        rn = ReplicatedReadsNormaliser(eng)
        with nk.KmerNormaliser(15, 100, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
            seen = 0
            host = [synth.interleave(c1, c2) for c1, c2 in synth.pair_blocks(4 * bp, seed=5, transcripts=40, block=bp)]
            dev = [(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(off) - 1, len(seq)) for seq, off in host]
            for i, (seq, off) in enumerate(host):
                # steps 0 and 1 announce their successor (its all-gather overlaps the counting), step 2 does not
                high, total, gathered = rn.process_step(*dev[i], then=dev[i + 1] if i < 2 else None)
                h, t = g.process_batch(seq, off)
                np.testing.assert_array_equal(high.cpu().numpy(), h)
                np.testing.assert_array_equal(total.cpu().numpy(), t)
                np.testing.assert_array_equal(gathered[0][1].cpu().numpy(), t)
                seen += int(h.sum())
            assert seen > 0 and eng.stats()["unique_kmers"] == g.unique_kmers
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [15, 13, 25])
def test_one_batch_of_more_than_two_to_the_31_positions(nk, k):
    """2.4 G base positions (8 M pairs) in ONE batch - positions, key arrays (9.6 GB) and sorted-position arrays well past
    2^31 entries / 2^32 bytes - against the same reads handed over in four batches of 2 M pairs: the per-read counters
    and the number of distinct k-mers must not depend on the cut (the small batches are covered against the oracle and the
    known answer elsewhere)."""
    import torch
    from nomalise_kmers_multi_large_amd import binding, synth
    quarter, copies = 2000000, 4
    blocks = list(synth.pair_blocks(quarter, seed=1, transcripts=2000))
    seq, off = synth.interleave(np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks]))
    del blocks
    nr, nb = len(off) - 1, len(seq)
    # four different quarters from one: quarter j is the first one rotated by j reads (mates stay together only for j even, which
    # does not matter to the per-read counters)
    d_seq1, d_off1 = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    assert (np.diff(off) == 150).all()
    parts = [torch.roll(d_seq1.view(nr, 150), shifts=j * 7919, dims=0).reshape(-1) for j in range(copies)]
    d_seq = torch.cat(parts)
    d_off = torch.arange(0, copies * nr + 1, dtype=torch.int64, device="cuda") * 150
    assert d_seq.numel() == copies * nb > (1 << 31)
    # k=15: partitioned path in both; k=13: atomic path on the direct table; k=25: hashed table, canonical, growing,
    # partitioned as well (buckets of 37 K keys: several chunks per bucket workgroup)
    big = nk.KmerNormaliser(k, 100, 0.9, k == 25, max_batch_reads=copies * nr, max_batch_bases=copies * nb,
                            flags=binding.NKML_F_COUNT_PARTITIONED if k == 15 else 0)
    small = nk.KmerNormaliser(k, 100, 0.9, k == 25, max_batch_reads=nr, max_batch_bases=nb)
    try:
        assert big.stats()["partitioned"] == small.stats()["partitioned"] == (0 if k == 13 else 1)
        out_big = torch.zeros((2, copies * nr), dtype=torch.int32, device="cuda")
        out_small = torch.zeros((2, copies * nr), dtype=torch.int32, device="cuda")
        big.process_device(d_seq.data_ptr(), d_off.data_ptr(), copies * nr, copies * nb, out_big[0].data_ptr(), out_big[1].data_ptr())
        for j in range(copies):
            small.process_device(parts[j].data_ptr(), d_off1.data_ptr(), nr, nb, out_small[0, j * nr:].data_ptr(), out_small[1, j * nr:].data_ptr())
        big.synchronize()
        small.synchronize()
        assert torch.equal(out_big, out_small)
        assert int(out_big[0].sum()) > 0 and int(out_big[0].sum()) < int(out_big[1].sum())
        assert big.unique_kmers == small.unique_kmers
    finally:
        big.close()
        small.close()


@pytest.mark.parametrize("mode", ["replicated", "exchange", "bucketed"])
def test_multi_gpu_step_to_which_a_rank_brings_nothing(nk, mode):
    """The multi-GPU normalisers over RCCL (world of one): a step without any read between two ordinary steps returns
    empty counters and leaves the table alone; the ordinary steps equal the single-table path."""
    import socket
    import torch
    import torch.distributed as dist
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.replicated import make_multi_gpu_normaliser
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bp = 8000
        mg, engine, chosen = make_multi_gpu_normaliser(15, 20, 0.9, False, 0, max_reads=2 * bp, max_bases=2 * bp * 150, mode=mode)
        assert chosen == mode
        host = [synth.interleave(c1, c2) for c1, c2 in synth.pair_blocks(2 * bp, seed=3, transcripts=10, block=bp)]
        empty = (torch.zeros(0, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"), 0, 0)
        steps = [host[0], None, host[1]]
        dev = [empty if h is None else (torch.from_numpy(h[0]).cuda(), torch.from_numpy(h[1].astype(np.int64)).cuda(), len(h[1]) - 1, len(h[0])) for h in steps]
        with nk.KmerNormaliser(15, 20, 0.9, max_batch_reads=2 * bp, max_batch_bases=2 * bp * 150) as g:
            for i, h in enumerate(steps):
                high, total, gathered = mg.process_step(*dev[i], then=dev[i + 1] if i + 1 < len(dev) else None)
                if h is None:
                    assert high.numel() == 0 and total.numel() == 0 and gathered[0].shape[1] == 0
                    continue
                want_h, want_t = g.process_batch(*h)
                np.testing.assert_array_equal(high.cpu().numpy(), want_h)
                np.testing.assert_array_equal(total.cpu().numpy(), want_t)
            assert mg.stats()["unique_kmers"] == g.unique_kmers and int(want_h.sum()) > 0
    finally:
        dist.destroy_process_group()


# ---- one table over several devices, driven from one process (nkml_multi_*, what the command line uses with NKML_DEVICES) ----
@pytest.mark.parametrize("devices,transport,k,depth,canonical", [
    ([0, 0], "copy", 15, 3, False),        # partitioned counting behind the exchange, two shards sharing the GPU
    ([0, 0, 0], "copy", 15, 100, True),
    ([0, 0], "copy", 11, 4, True),         # direct table, atomic path
    ([0, 0], "copy", 25, 4, True),         # hashed table
    ([0, 0, 0, 0], "copy", 31, 3, False),
    ([0], "copy", 15, 3, False),           # a "multi" of one
    ([0], "rccl", 21, 3, True),            # the RCCL send/receive transport with a world of one (all a one-GPU box can run)
])
def test_multi_device_table_equals_single_stream(nk, oracle_lib, devices, transport, k, depth, canonical):
    from nomalise_kmers_multi_large_amd.normaliser import MultiDeviceNormaliser
    rng = np.random.default_rng(900 + k + len(devices))
    reads = random_reads(rng, 500, k, 4 * k + 40, b"AC") + random_reads(rng, 700, k, 200, b"ACGT", 0.01) \
        + random_reads(rng, 300, k, 90, b"GT") + [b"A" * 50, b"T" * 77, b"G" * 64, b"C" * 33, b"N" * 40] * 3
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth, canonical)
    seeds = [r for r in reads[:200] if len(r) > k]   # the reference seeds with reads longer than k only
    for r in seeds:
        o.seed(r)
    want = [o.read(r) for r in reads]
    old = os.environ.get("NKML_TRANSPORT")
    os.environ["NKML_TRANSPORT"] = transport
    try:
        with MultiDeviceNormaliser(devices, k, depth, 0.9, canonical, max_batch_reads=700, max_batch_bases=90000) as g:
            seq, off = nk.KmerNormaliser._concat(seeds)
            g.seed_batch(seq, off)
            assert g.unique_kmers == o_unique_after_seeding(oracle_lib, k, depth, canonical, seeds)
            high, total = [], []
            for at in range(0, len(reads), 450):   # several ragged batches
                seq, off = nk.KmerNormaliser._concat(reads[at:at + 450])
                h, t = g.process_batch(seq, off)
                high.append(h)
                total.append(t)
            np.testing.assert_array_equal(np.concatenate(high), [w[1] for w in want])
            np.testing.assert_array_equal(np.concatenate(total), [w[2] for w in want])
            assert g.unique_kmers == o.unique
            codes, counts, n = g.dump_table()
            assert n == o.unique and len(set(codes.tolist())) == n
            # a read with a byte outside ACGTN stops the batch with the reference's error class, whatever shard it falls to
            bad = reads[:40] + [b"ACGT" * 10 + b"x" + b"ACGT" * 10] + reads[40:80]
            seq, off = nk.KmerNormaliser._concat(bad)
            with pytest.raises(nk.NkmlError) as e:
                g.process_batch(seq, off)
            assert e.value.code == -4
            # some shards have counted their part of that batch: the tables are no longer those of a stream prefix, and the
            # driver says so until it is reset
            with pytest.raises(nk.NkmlError) as e2:
                g.process_batch(*nk.KmerNormaliser._concat(reads[:10]))
            assert e2.value.code == -5
            g.reset()
            h, t = g.process_batch(*nk.KmerNormaliser._concat(reads[:10]))
            assert len(h) == 10 and g.unique_kmers > 0
    finally:
        if old is None:
            os.environ.pop("NKML_TRANSPORT", None)
        else:
            os.environ["NKML_TRANSPORT"] = old


@pytest.mark.parametrize("piece,pairs", [(1 << 16, 20000), (0, 500000)])
def test_rccl_transport_moves_large_segments_in_pieces(nk, piece, pairs):
    """The C host's RCCL send/receive transport (a world of one: all a one-GPU box can run) against the single table:
    with pieces of 64 KiB every segment takes many sends, and with the default piece a batch of 0.5 M pairs at k=25
    sends 1.2 GB of k-mers to the one shard - more than the 1 GiB beyond which this stack delivers half a message."""
    from nomalise_kmers_multi_large_amd import synth
    from nomalise_kmers_multi_large_amd.normaliser import MultiDeviceNormaliser
    blocks = list(synth.pair_blocks(pairs, seed=3, transcripts=max(50, pairs // 1000)))
    seq, off = synth.interleave(np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks]))
    off = off.astype(np.uint64)
    nr, nb = len(off) - 1, len(seq)
    saved = {v: os.environ.get(v) for v in ("NKML_TRANSPORT", "NKML_RCCL_PIECE_BYTES")}
    os.environ["NKML_TRANSPORT"] = "rccl"
    if piece:
        os.environ["NKML_RCCL_PIECE_BYTES"] = str(piece)
    try:
        with MultiDeviceNormaliser([0], 25, 100, 0.9, True, max_batch_reads=nr, max_batch_bases=nb) as g, \
                nk.KmerNormaliser(25, 100, 0.9, True, max_batch_reads=nr, max_batch_bases=nb) as one:
            for _ in range(2):   # the second batch finds the k-mers of the first
                h, t = g.process_batch(seq, off)
                h1, t1 = one.process_batch(seq, off)
                np.testing.assert_array_equal(h, h1)
                np.testing.assert_array_equal(t, t1)
            assert g.unique_kmers == one.unique_kmers and int(h.sum()) > 0
    finally:
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old


def o_unique_after_seeding(oracle_lib, k, depth, canonical, seeds):
    o = oracle_lib(k, depth, canonical)
    for r in seeds:
        o.seed(r)
    return o.unique


@pytest.mark.gpu
def test_a_device_that_fails_the_lds_order_check_counts_on_the_atomic_path():
    """The partitioned paths rank with returning LDS adds; nkml_create checks the device's order once.  On a device that
    fails, the default context quietly becomes an atomic-path context (one line on stderr) and agrees with the oracle, at
    k = 15 and at k = 25; only a context that asks for the partitioned path by flag is refused.  NKML_TEST_LDS_DISORDER
    makes the check fail; a process of its own, as the result is kept per device."""
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import binding as B, synth
from tests.helpers import Oracle
c1, c2 = next(synth.pair_blocks(1000, seed=3, transcripts=4, block=1000))
seq, off = synth.interleave(c1, c2)
for k, canonical in ((15, False), (25, True)):
    o = Oracle(k, 20, canonical)
    want_h, want_t = o.batch(seq, off)
    with nk.KmerNormaliser(k, 20, 0.9, canonical, max_batch_reads=4096, max_batch_bases=4096 * 150) as g:
        h, t = g.process_batch(seq, off)
        print("DEFAULT", k, bool(np.array_equal(h, want_h) and np.array_equal(t, want_t)), "partitioned=%d" % g.stats()["partitioned"])
try:
    nk.KmerNormaliser(15, 20, 0.9, False, max_batch_reads=4096, max_batch_bases=4096 * 150, flags=B.NKML_F_COUNT_PARTITIONED)
    print("FORCED CREATED")
except Exception as e:
    print("FORCED REFUSED", "lane order" in str(e))
'''
    env = dict(os.environ, NKML_TEST_LDS_DISORDER="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "DEFAULT 15 True partitioned=0" in out.stdout, out.stdout + out.stderr
    assert "DEFAULT 25 True partitioned=0" in out.stdout, out.stdout + out.stderr
    assert "FORCED REFUSED True" in out.stdout, out.stdout + out.stderr
    assert "counting on the atomic path" in out.stderr, out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("k,depth,canonical,log2_bucket", [(15, 20, False, None), (15, 300, True, None), (13, 5, False, None),
                                                          (25, 20, True, None), (25, 4, True, 3), (31, 200, False, (3, 3))])
def test_rank_verification_mode_passes_and_agrees_with_the_oracle(nk, oracle_lib, k, depth, canonical, log2_bucket):
    """NKML_F_VERIFY_RANKS: every batch of the partitioned paths is checked on the device by kernels that take no rank from
    the LDS's lane order (each pass against a ballot-built stable partition, each bucket against a copy of the table from
    before the batch).  Batches with repeats, crossings, saturated k-mers, growth of the hashed table inside a batch and
    buckets of several sub-ranges go through it without a complaint, and the results are the oracle's."""
    from nomalise_kmers_multi_large_amd import binding
    log2_local = None
    if isinstance(log2_bucket, tuple):
        log2_bucket, log2_local = log2_bucket
    rng = np.random.default_rng(9100 + k + depth)
    reads = random_reads(rng, 500, k, 4 * k + 40, b"AC") + random_reads(rng, 800, k, 200, b"ACGT", 0.01) \
        + random_reads(rng, 300, k, 90, b"GT") + [b"A" * 50, b"T" * 77, b"N" * 40] * 3 + [b"G" * 150] * 500
    reads = [reads[i] for i in rng.permutation(len(reads))]
    o = oracle_lib(k, depth, canonical)
    want = [o.read(r) for r in reads]
    saved = {v: os.environ.get(v) for v in ("NKML_HP_LOG2_BUCKET", "NKML_HP_LOG2_LOCAL")}
    if log2_bucket is not None:
        os.environ["NKML_HP_LOG2_BUCKET"] = str(log2_bucket)
    if log2_local is not None:
        os.environ["NKML_HP_LOG2_LOCAL"] = str(log2_local)
    try:
        with nk.KmerNormaliser(k, depth, 0.9, canonical, max_batch_reads=300, max_batch_bases=300 * 200,
                               flags=binding.NKML_F_COUNT_PARTITIONED | binding.NKML_F_VERIFY_RANKS) as g:
            assert g.stats()["partitioned"] == 1
            ok, high, total = g.process_sequence_single(reads)
            np.testing.assert_array_equal(high, [w[1] for w in want])
            np.testing.assert_array_equal(total, [w[2] for w in want])
            assert g.unique_kmers == o.unique
            if log2_bucket is not None:
                assert g.stats()["resizes"] > 0
    finally:
        for var, val in saved.items():
            if val is None:
                os.environ.pop(var, None)
            else:
                os.environ[var] = val


@pytest.mark.gpu
@pytest.mark.parametrize("fault", [1, 2])
def test_rank_verification_mode_catches_a_damaged_batch(fault):
    """The checkers have teeth: with one flag of a bucket turned round (fault 1) or two neighbours of the first pass's
    output swapped (fault 2) - NKML_TEST_VERIFY_FAULT damages every batch this way before the checks run - the batch fails
    with NKML_E_STATE and the context stays unusable until it is reset.  At k = 15 and at k = 25; a process of its own."""
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import nomalise_kmers_multi_large_amd as nk
from nomalise_kmers_multi_large_amd import binding as B, synth
c1, c2 = next(synth.pair_blocks(1000, seed=5, transcripts=4, block=1000))
seq, off = synth.interleave(c1, c2)
for k in (15, 25):
    with nk.KmerNormaliser(k, 20, 0.9, True, max_batch_reads=4096, max_batch_bases=4096 * 150,
                           flags=B.NKML_F_COUNT_PARTITIONED | B.NKML_F_VERIFY_RANKS) as g:
        for attempt in range(2):   # the second call meets the context the first one left unusable
            try:
                g.process_batch(seq, off)
                print("PASSED", k, attempt)
            except Exception as e:
                print("CAUGHT", k, attempt, "rank verification" in str(e) or "unusable" in str(e) or "reset" in str(e), "|", e)
'''
    env = dict(os.environ, NKML_TEST_VERIFY_FAULT=str(fault))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for k in (15, 25):
        for attempt in range(2):
            assert "CAUGHT %d %d True" % (k, attempt) in out.stdout, out.stdout + out.stderr
    assert "PASSED" not in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("k,canonical", [(15, False), (25, True)])
@pytest.mark.parametrize("n_shards", [1, 2, 3, 8, 32])
def test_ranked_routing_sends_every_kmer_to_its_place(nk, k, canonical, n_shards):
    """nkml_route_reads_ranked on its own: with the 16 bits it leaves per base position and the tiles' start offsets, the key
    of every position is found at offsets[shard][tile] + rank in what is sent, shard = nkml_owner_of(key); positions
    without a k-mer say so; every shard's segment is in stream order (places grow with the position); and
    nkml_unroute_ranked brings a flag per sent key back to exactly the position it came from."""
    import torch
    rng = np.random.default_rng(50 + k + n_shards)
    reads = random_reads(rng, 400, k, 200, b"ACGT", 0.02) + random_reads(rng, 200, k, 4 * k, b"AC") + [b"A" * 60, b"N" * 45, b"G" * 150] * 5
    reads = [reads[i] for i in rng.permutation(len(reads))]
    with nk.KmerNormaliser(k, 10, 0.9, canonical, device=0, max_batch_reads=4096, max_batch_bases=1 << 20) as g:
        seq, off = g._concat(reads)
        nb, n16 = len(seq), (len(seq) + 15) // 16 * 16
        d_seq, d_off = torch.from_numpy(seq.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        wide = g.key_bytes == 8
        kt = torch.int64 if wide else torch.int32
        keys = torch.zeros(n16, dtype=kt, device="cuda")
        g.kmerise_device(d_seq.data_ptr(), d_off.data_ptr(), len(reads), nb, keys.data_ptr())
        tiles = (n16 + 1023) // 1024
        send = torch.full((n16,), -1, dtype=kt, device="cuda")
        meta = torch.zeros(tiles * 1024, dtype=torch.int16, device="cuda")
        tile_off = torch.zeros(tiles * n_shards + 1, dtype=torch.int32, device="cuda")
        counts, home = g.route_reads_ranked(d_seq.data_ptr(), d_off.data_ptr(), len(reads), nb, n_shards, send.data_ptr(), meta.data_ptr(), tile_off.data_ptr())
        g.synchronize()
        assert home == 0
        kh = keys.cpu().numpy().astype(np.uint64 if wide else np.uint32).astype(np.uint64)
        m = meta.cpu().numpy().view(np.uint16)[:n16].astype(np.int64)
        to = tile_off.cpu().numpy().astype(np.int64)
        sent = send.cpu().numpy().astype(np.uint64 if wide else np.uint32).astype(np.uint64)
        has = kh != 0
        assert sum(counts) == int(has.sum()) and ((m == 0x8000) == ~has).all()
        pos = np.nonzero(has)[0]
        own, rank = m[pos] >> 10, m[pos] & 1023
        want_own = ((((kh[pos] * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)) * np.uint64(n_shards)) >> np.uint64(32)).astype(np.int64)
        np.testing.assert_array_equal(own, want_own)
        place = to[own * tiles + (pos >> 10)] + rank
        np.testing.assert_array_equal(sent[place], kh[pos])
        assert len(np.unique(place)) == len(place) and place.max() == sum(counts) - 1
        for s in range(n_shards):   # stream order inside every shard's segment, segments back to back in shard order
            ps = place[own == s]
            assert (np.diff(ps) > 0).all() and (len(ps) == 0 or (ps[0] == sum(counts[:s]) and ps[-1] == sum(counts[:s + 1]) - 1))
        # the way home: a flag per sent key (here: its place modulo 251 + 1) arrives at the position the key came from
        back = torch.from_numpy(((np.arange(sum(counts)) % 251) + 1).astype(np.uint8)).cuda()
        flags = torch.full((n16,), 77, dtype=torch.uint8, device="cuda")
        g.unroute_ranked(back.data_ptr(), meta.data_ptr(), tile_off.data_ptr(), n16, flags.data_ptr())
        g.synchronize()
        f = flags.cpu().numpy()
        np.testing.assert_array_equal(f[pos], (place % 251) + 1)
        assert (f[~has] == 0).all()


@pytest.mark.gpu
def test_a_context_without_a_table_refuses_to_count(nk):
    """NKML_F_NO_TABLE: the sender-only contexts of the bucket-ordered exchange hold no table; every call that would count says
    so (NKML_E_STATE) instead of touching memory that is not there, and the flag is refused where it makes no sense (k > 16)."""
    from nomalise_kmers_multi_large_amd import binding as B
    flags = B.NKML_F_COUNT_PARTITIONED | B.NKML_F_NO_TABLE
    with nk.KmerNormaliser(15, 20, 0.9, False, max_batch_reads=64, max_batch_bases=64 * 150, flags=flags) as g:
        assert g.stats()["partitioned"] == 1
        with pytest.raises(B.NkmlError) as e:
            g.process_sequence_single([b"ACGTACGTTGCATGCAAGCTTGCA" * 3])
        assert e.value.code == B.NKML_E_STATE and "without a table" in str(e.value)
        with pytest.raises(B.NkmlError) as e:
            g.dump_table(16)
        assert e.value.code == B.NKML_E_STATE
    with pytest.raises(B.NkmlError):
        nk.KmerNormaliser(25, 20, 0.9, True, max_batch_reads=64, max_batch_bases=64 * 150, flags=flags)


# ---- routing on its own: the whole output against a stable partition computed with numpy ----
@pytest.mark.gpu
@pytest.mark.parametrize("k", [15, 25])
@pytest.mark.parametrize("n_shards", [1, 2, 3, 8, 64])
def test_route_keys_is_the_stable_partition_by_owner(nk, k, n_shards):
    import torch
    rng = np.random.default_rng(40 + k + n_shards)
    with nk.KmerNormaliser(k, 10, 0.9, False, device=0, max_batch_reads=1 << 16, max_batch_bases=1 << 22) as g:
        wide = g.key_bytes == 8
        dt_np, dt_t = (np.int64, torch.int64) if wide else (np.int32, torch.int32)
        for n in (16, 1024 * 37 + 16, 1 << 20):   # less than a tile, ragged last tile, many tiles
            keys = rng.integers(1, (1 << (2 * k)) - 1 if wide else (1 << 30) - 1, n, dtype=np.int64)
            keys[rng.random(n) < 0.15] = 0                      # positions without a k-mer are not sent
            keys[n // 3: n // 3 + min(n // 4, 3000)] = keys[0]  # a long run of one owner inside the tiles it covers
            keys_t = torch.from_numpy(keys.astype(dt_np)).cuda()
            send_keys = torch.full((n,), -1, dtype=dt_t, device="cuda")
            send_src = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            counts = g.route_keys_device(keys_t.data_ptr(), n, n_shards, send_keys.data_ptr(), send_src.data_ptr())
            g.synchronize()
            own = ((((keys.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)) * np.uint64(n_shards)) >> np.uint64(32)).astype(np.int64)
            own[keys == 0] = n_shards
            order = np.argsort(own, kind="stable")[: int((keys != 0).sum())]
            assert counts == [int((own == s).sum()) for s in range(n_shards)]
            np.testing.assert_array_equal(send_src[: len(order)].cpu().numpy(), order.astype(np.int32))
            np.testing.assert_array_equal(send_keys[: len(order)].cpu().numpy(), keys[order].astype(dt_np))
            assert int(send_src[len(order):].max().item()) == -1 if len(order) < n else True   # nothing written past the end
