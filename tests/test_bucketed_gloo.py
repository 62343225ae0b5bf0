"""The bucket-ordered exchange (bucketed.py), run for real over torch.distributed (gloo, CPU, 2 and 3 ranks).  The device work
is done by a CPU checker engine built on the oracle, so what is under test is the exchange logic itself: what goes to which
owner, the per-bucket counts, the runs' order at the owner, the flags' way back, the saturated k-mers' lists, the gather on
rank 0.  The result must equal one oracle reading the global stream (rank 0's reads, then rank 1's, ...) on a single table."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import Oracle
from tests.test_sharded_gloo import concat, free_port, make_reads

LOW = 6   # bits of a k-mer code below its bucket number (the product has 14; small k-mers need small buckets to have several)


class BucketChecker:
    """Same stage interface as HipBucketEngine, on CPU tensors, with the oracle as the table; buckets are dealt in their
    natural order (the product deals them by a fixed permutation, which the exchange logic never sees)."""

    def __init__(self, k, depth, canonical, world, rank, stay_home=True):
        from nomalise_kmers_multi_large_amd.bucketed import first_place
        self.k, self.canonical, self.depth, self.world, self.rank = k, canonical, depth, world, rank
        self.o = Oracle(k, depth, canonical)
        self.buckets = (1 << (2 * k)) >> LOW
        self.first = [first_place(self.buckets, world, r) for r in range(world + 1)]
        self.mine = self.first[rank + 1] - self.first[rank]
        self.stay_home = stay_home
        self.sat, self.seen, self.home_total = set(), set(), 0
        self.which, self._fronts = 0, [None, None]
        if not stay_home:
            self.sat_collect = None

    def use_buffers(self, which):
        self.which = which

    def xb_front(self, seq, offsets, n_reads, n_bases):
        codes = Oracle.keys(self.k, self.canonical, seq.numpy(), offsets.numpy().astype(np.uint64))
        pos = np.nonzero(codes)[0]
        known = np.isin(codes[pos], np.fromiter(self.sat, dtype=np.uint64, count=len(self.sat))) if self.stay_home else np.zeros(len(pos), dtype=bool)
        home, pos = pos[known], pos[~known]
        self.home_total += len(home)
        bucket = (codes[pos] >> np.uint64(LOW)).astype(np.int64)
        order = np.argsort(bucket, kind="stable")            # bucket after bucket, stream order within a bucket
        self._fronts[self.which] = (pos[order], home, len(codes))
        counts = np.bincount(bucket, minlength=self.buckets).astype(np.int32)
        totals = [int(counts[self.first[r]:self.first[r + 1]].sum()) for r in range(self.world)]
        low = (codes[pos][order] & np.uint64((1 << LOW) - 1)).astype(np.int16)
        return torch.from_numpy(low), torch.from_numpy(counts), totals

    def xb_recv_buffers(self, n):
        return torch.empty(n, dtype=torch.int16), torch.empty(self.world * self.mine, dtype=torch.int32)

    def xb_count(self, recv_keys, recv_counts):
        rc = recv_counts.numpy().reshape(self.world, self.mine).astype(np.int64)
        run_at = np.concatenate([[0], np.cumsum(rc.reshape(-1))])[:-1].reshape(self.world, self.mine)
        low = recv_keys.numpy().astype(np.int64) & ((1 << LOW) - 1)
        # the owner's order: bucket after bucket, within a bucket the ranks' runs in rank order
        src = [np.arange(run_at[s, j], run_at[s, j] + rc[s, j]) for j in range(self.mine) for s in range(self.world)]
        bkt = [np.full(rc[s, j], self.first[self.rank] + j, dtype=np.int64) for j in range(self.mine) for s in range(self.world)]
        src = np.concatenate(src) if src else np.zeros(0, dtype=np.int64)
        bkt = np.concatenate(bkt) if bkt else np.zeros(0, dtype=np.int64)
        codes = ((bkt << LOW) | low[src]).astype(np.uint64)
        self.seen.update(np.unique(codes).tolist())
        flags = np.zeros(len(low), dtype=np.uint8)
        flags[src] = self.o.count_keys(codes)
        return torch.from_numpy(flags)

    def xb_back_buffer(self, n):
        return torch.empty(n, dtype=torch.uint8)

    def xb_finish(self, back_flags, offsets, n_reads, which):
        pos, home, n = self._fronts[which]
        f = np.zeros(n, dtype=np.uint8)
        f[home] = 1
        f[pos] = back_flags.numpy()
        off = offsets.numpy()
        high = np.array([(f[off[i]:off[i + 1]] == 1).sum() for i in range(n_reads)], dtype=np.int32)
        total = np.array([(f[off[i]:off[i + 1]] != 0).sum() for i in range(n_reads)], dtype=np.int32)
        return torch.from_numpy(high), torch.from_numpy(total)

    def sat_collect(self):
        new = [c for c in self.seen if c not in self.sat and self.o.count_of(c) >= self.depth - 1]
        self.sat.update(new)
        self.seen.clear()
        return torch.tensor(new, dtype=torch.int64)

    def sat_apply(self, codes):
        self.sat.update(codes.numpy().view(np.uint64).tolist())

    def synchronize(self):
        pass

    def stats(self):
        return {"unique_kmers": self.o.unique}


def worker(rank, world, port, k, depth, canonical, steps, q, msg, stay_home, empty_rank, ahead=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nomalise_kmers_multi_large_amd.bucketed import BucketedNormaliser
        sn = BucketedNormaliser(BucketChecker(k, depth, canonical, world, rank, stay_home), max_message_bytes=msg)
        out, batches = [], []
        for step in range(steps):
            n = 0 if (rank == empty_rank and step == 1) else 40 + 7 * rank   # ragged; one rank brings nothing to one step
            reads = make_reads(100 * step + rank, n, k, 90)
            seq, off = concat(reads) if reads else (np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
            batches.append((torch.from_numpy(seq), torch.from_numpy(off), len(reads), len(seq)))
        for step in range(steps):
            # with look-ahead the next step is partitioned and its keys are sent before this step's are counted
            high, total, gathered = sn.process_step(*batches[step], then=batches[step + 1] if ahead and step + 1 < steps else None)
            if rank == 0:
                out.append([g.numpy().copy() for g in gathered])
        tallies = torch.tensor([sn.stats()["unique_kmers"], sn.eng.home_total], dtype=torch.int64)
        dist.all_reduce(tallies)
        if rank == 0:
            q.put((out, tallies.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,depth,canonical,msg,stay_home,empty_rank,ahead", [
    (2, 7, 3, False, 256 << 20, True, None, False), (3, 7, 4, True, 256 << 20, True, 1, False), (2, 9, 2, True, 1000, True, None, False),
    (3, 9, 5, False, 256 << 20, False, None, False), (2, 11, 3, True, 1000, False, 0, False),
    (3, 7, 3, True, 256 << 20, True, 1, True), (2, 9, 2, False, 1000, True, 0, True)])
def test_bucketed_exchange_equals_single_stream(oracle_lib, world, k, depth, canonical, msg, stay_home, empty_rank, ahead):
    steps = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, k, depth, canonical, steps, q, msg, stay_home, empty_rank, ahead)) for r in range(world)]
    for p in procs:
        p.start()
    got, (uniq, stayed) = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (stayed > 0) == stay_home
    o = oracle_lib(k, depth, canonical)
    for step in range(steps):
        for rank in range(world):
            n = 0 if (rank == empty_rank and step == 1) else 40 + 7 * rank
            reads = make_reads(100 * step + rank, n, k, 90)
            want = [o.read(r) for r in reads]
            np.testing.assert_array_equal(got[step][rank][0], np.array([w[1] for w in want], dtype=np.int32))
            np.testing.assert_array_equal(got[step][rank][1], np.array([w[2] for w in want], dtype=np.int32))
    assert uniq == o.unique   # the owners' tables are disjoint: their sizes add up to the single table's
