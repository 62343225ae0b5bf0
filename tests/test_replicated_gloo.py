"""The replicated-reads multi-GPU mode, run for real over torch.distributed (gloo, CPU, 2 and 3 ranks): every rank
gets every read of a step (packed words, one all-gather), counts the k-mers of its own table shard and the per-read
partial counters are summed.  The device work is done by a CPU checker engine built on the oracle, so what is under
test is the step layout (regions, filler reads, offsets), the collectives and the slicing of the result.  It must equal
one oracle reading the global stream (rank 0's reads, then rank 1's, ...) on a single table."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import Oracle
from tests.test_sharded_gloo import concat, free_port, make_reads

CODE = np.zeros(256, dtype=np.uint32)
for _i, _c in enumerate(b"ACGT"):
    CODE[_c] = _i
SHIFTS = np.arange(30, -2, -2, dtype=np.uint32)


class CheckerReplicaEngine:
    """Same face as HipReplicaEngine, on CPU tensors, with the oracle as this rank's table shard."""

    def __init__(self, k, depth, canonical, world, rank):
        self.k, self.canonical, self.world, self.rank = k, canonical, world, rank
        self.o = Oracle(k, depth, canonical)

    def pack(self, seq, n_bases, region, buffer=0):
        codes = np.zeros(region * 16, dtype=np.uint32)
        codes[:n_bases] = CODE[seq.numpy()[:n_bases]]
        words = (codes.reshape(region, 16) << SHIFTS).sum(axis=1, dtype=np.uint64).astype(np.uint32)
        return torch.from_numpy(words.view(np.int32))

    def gather_buffer(self, n_words, buffer=0):
        return torch.empty(n_words, dtype=torch.int32)

    def process_packed(self, words_all, offsets, n_reads, n_bases):
        w = words_all.numpy().view(np.uint32)
        codes = ((w[:, None] >> SHIFTS) & 3).reshape(-1)[:n_bases]
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        off = offsets.numpy().astype(np.uint64)
        assert len(off) == n_reads + 1 and off[0] == 0 and off[-1] == n_bases and (np.diff(off.astype(np.int64)) >= 0).all()
        keys = Oracle.keys(self.k, self.canonical, seq, off)
        keys[keys % np.uint64(self.world) != np.uint64(self.rank)] = 0
        f = self.o.count_keys(keys)
        off = off.astype(np.int64)
        out = np.zeros((2, n_reads), dtype=np.int32)
        for i in range(n_reads):
            out[0, i] = (f[off[i]:off[i + 1]] == 1).sum()
            out[1, i] = (f[off[i]:off[i + 1]] != 0).sum()
        return torch.from_numpy(out)

    def synchronize(self):
        pass

    def stats(self):
        return {"unique_kmers": self.o.unique}


def reads_of(step, rank, k, case):
    if case == "ragged":
        return make_reads(100 * step + rank, 40 + 7 * rank, k, 90)
    if case == "lopsided":   # one rank brings far less than the others: its region is mostly filler
        return make_reads(100 * step + rank, 3 if rank == (step % 2) else 400, k, 120)
    if case == "empty":      # in every step one rank has nothing at all
        return [] if rank == (step % 2) else make_reads(100 * step + rank, 60, k, 120)
    if case == "long":       # a read of 70 000 bases: its counters do not fit 16 bits, so they travel as two words
        return make_reads(100 * step + rank, 5, k, 90) + (make_reads(7 + step, 1, 70000, 70000) if rank == 1 else [])
    if case == "long16":     # 60 000 bases: still one word for both counters, the upper one beyond 2^15
        return make_reads(100 * step + rank, 5, k, 90) + (make_reads(7 + step, 1, 60000, 60000) if rank == 0 else [])
    raise ValueError(case)


def worker(rank, world, port, k, depth, canonical, steps, case, ahead, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nomalise_kmers_multi_large_amd.replicated import ReplicatedReadsNormaliser
        # with look-ahead also: counters reduced to rank 0 only
        rn = ReplicatedReadsNormaliser(CheckerReplicaEngine(k, depth, canonical, world, rank), check_words=5000, root_only=ahead)
        out = []
        batches = []
        for step in range(steps):
            seq, off = concat(reads_of(step, rank, k, case))
            batches.append((torch.from_numpy(seq), torch.from_numpy(off), len(off) - 1, len(seq)))
        for step in range(steps):
            # with look-ahead the next step's reads are exchanged before this step is counted
            then = batches[step + 1] if ahead and step + 1 < steps else None
            high, total, gathered = rn.process_step(*batches[step], then=then)
            assert (gathered is None) == (rank != 0) and (high is None if ahead and rank else high.shape[0] == batches[step][2])
            if rank == 0:
                out.append([g.numpy().copy() for g in gathered])
        uniq = torch.tensor([rn.stats()["unique_kmers"]], dtype=torch.int64)
        dist.all_reduce(uniq)
        if rank == 0:
            q.put((out, int(uniq.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,depth,canonical,case,ahead", [(2, 9, 3, False, "ragged", False), (3, 11, 2, True, "ragged", True),
                                                                (2, 7, 5, False, "lopsided", False), (2, 7, 5, False, "lopsided", True),
                                                                (2, 9, 2, True, "empty", True), (2, 7, 3, False, "long", True), (2, 7, 3, False, "long16", False)])
def test_replicated_reads_equal_single_stream(oracle_lib, world, k, depth, canonical, case, ahead):
    steps = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, k, depth, canonical, steps, case, ahead, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, uniq = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = oracle_lib(k, depth, canonical)
    seen_high = 0
    for step in range(steps):
        for rank in range(world):
            reads = reads_of(step, rank, k, case)
            want = [o.read(r) for r in reads]
            np.testing.assert_array_equal(got[step][rank][0], [w[1] for w in want])
            np.testing.assert_array_equal(got[step][rank][1], [w[2] for w in want])
            seen_high += sum(w[1] for w in want)
    assert seen_high > 0
    assert uniq == o.unique


def test_step_layout_and_offsets():
    from nomalise_kmers_multi_large_amd import replicated as R
    reads, bases = [3, 0, 2], [100, 0, 16 * 5000]
    region, fillers, first, total = R.step_layout(reads, bases)
    assert region == 5000
    assert fillers == [-(-(80000 - 100) // R.GAP_BASES), -(-80000 // R.GAP_BASES), 1]
    assert first == [0, 3 + fillers[0], 3 + fillers[0] + fillers[1]] and total == first[2] + 2 + 1
    local = [torch.tensor([0, 30, 60, 100]), torch.tensor([0]), torch.tensor([0, 40000, 80000])]
    off = R.global_offsets(local, reads, bases, region, fillers).numpy()
    assert len(off) == total + 1 and off[0] == 0 and off[-1] == 3 * region * 16
    assert (np.diff(off) >= 0).all()
    for r in range(3):
        np.testing.assert_array_equal(off[first[r]: first[r] + reads[r] + 1], local[r].numpy() + r * region * 16)
    # a huge gap is cut into GAP_READS fillers at most
    region, fillers, _, _ = R.step_layout([1, 1], [16, 16 * 40000000])
    assert fillers == [R.GAP_READS, 1]


def test_shard_rule_of_the_library_is_a_partition():
    from nomalise_kmers_multi_large_amd import binding
    lib = binding.load_library()
    rng = np.random.default_rng(3)
    keys = rng.integers(1, 1 << 30, 20000, dtype=np.uint64)
    for n in (1, 2, 3, 4, 7):
        s = np.array([lib.nkml_shard_of(int(x), n) for x in keys])
        assert s.min() >= 0 and s.max() < n
        share = np.bincount(s, minlength=n) / len(keys)
        assert np.abs(share - 1.0 / n).max() < 0.02
    assert lib.nkml_shard_of(1 << 33, 2) < 0 and lib.nkml_shard_of(5, 0) < 0


def failing_worker(rank, world, port, q):
    """Rank 1's counting raises (what a byte outside ACGTN among its reads does): both ranks must come back with an exception."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nomalise_kmers_multi_large_amd.replicated import ReplicatedReadsNormaliser

        class Faulty(CheckerReplicaEngine):
            def process_packed(self, words_all, offsets, n_reads, n_bases):
                if rank == 1:
                    raise ValueError("sequence byte outside ACGTN")
                return super().process_packed(words_all, offsets, n_reads, n_bases)

        rn = ReplicatedReadsNormaliser(Faulty(9, 3, False, world, rank), check_words=0, root_only=True)
        seq, off = concat(make_reads(rank, 30, 9, 60))
        try:
            rn.process_step(torch.from_numpy(seq), torch.from_numpy(off), len(off) - 1, len(seq))
            q.put((rank, "returned"))
        except Exception as e:
            q.put((rank, type(e).__name__ + ": " + str(e)))
    finally:
        dist.destroy_process_group()


def test_a_rank_that_fails_stops_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "ValueError" in got[1] and "another rank stopped this step" in got[0]
