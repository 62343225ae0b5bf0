"""The key exchange of the k-mer-sharded path, run for real over torch.distributed (gloo, CPU, 2 and 3 ranks).
The device work is done by a CPU checker engine built on the oracle, so what is under test is the exchange
logic itself: ownership, stable routing, split sizes, the flags' way back, the gather on rank 0.  The result
must equal one oracle reading the global stream (rank 0's reads, then rank 1's, ...) on a single table."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import Oracle

OWNER_MUL = np.uint64(0x9E3779B97F4A7C15)


def owner_np(keys, n_shards):
    return (((keys * OWNER_MUL) >> np.uint64(32)) * np.uint64(n_shards)) >> np.uint64(32)


class CheckerEngine:
    """Same stage interface as HipShardEngine, on CPU tensors, with the oracle as the table."""

    def __init__(self, k, depth, canonical, stay_home=False):
        self.k, self.canonical, self.depth = k, canonical, depth
        self.o = Oracle(k, depth, canonical)
        # stay_home: the protocol of the saturation bitmap (sat_collect / sat_apply), with Python sets for the bitmap
        self.stay_home = stay_home
        self.sat, self.seen, self.home_positions = set(), set(), 0
        if not stay_home:
            self.sat_collect = None   # ShardedNormaliser asks with getattr(engine, "sat_collect", None)

    def kmerise(self, seq, offsets, n_reads, n_bases):
        keys = Oracle.keys(self.k, self.canonical, seq.numpy(), offsets.numpy().astype(np.uint64))
        return torch.from_numpy(keys.view(np.int64))

    def route(self, keys, n_shards):
        k = keys.numpy().view(np.uint64)
        nz = np.nonzero(k)[0]
        home = np.zeros(0, dtype=np.int64)
        if self.stay_home:   # k-mers known to be saturated are answered here
            known = np.isin(k[nz], np.fromiter(self.sat, dtype=np.uint64, count=len(self.sat)))
            home, nz = nz[known], nz[~known]
            self.home_positions += len(home)
        own = owner_np(k[nz], n_shards).astype(np.int64)
        order = np.argsort(own, kind="stable")
        counts = [int((own == s).sum()) for s in range(n_shards)]
        src = torch.from_numpy(nz[order].astype(np.int32))
        return torch.from_numpy(k[nz][order].view(np.int64).copy()), ((src, torch.from_numpy(home)) if self.stay_home else src), counts

    def recv_buffer(self, n):
        return torch.empty(n, dtype=torch.int64)

    def count(self, recv_keys):
        keys = recv_keys.numpy().view(np.uint64)
        self.seen.update(np.unique(keys).tolist())
        return torch.from_numpy(self.o.count_keys(keys))

    def sat_collect(self):
        """k-mers of this shard whose count reached depth - 1 since the last call"""
        new = [c for c in self.seen if c not in self.sat and self.o.count_of(c) >= self.depth - 1]
        self.sat.update(new)
        self.seen.clear()
        return torch.tensor(new, dtype=torch.int64)

    def sat_apply(self, codes):
        self.sat.update(codes.numpy().view(np.uint64).tolist())

    def back_buffer(self, n):
        return torch.empty(n, dtype=torch.uint8)

    def unroute(self, back_flags, send_src, n_positions):
        flags = np.zeros(n_positions, dtype=np.uint8)
        if isinstance(send_src, tuple):
            send_src, home = send_src
            flags[home.numpy()] = 1   # "high"
        flags[send_src.numpy()] = back_flags.numpy()
        return torch.from_numpy(flags)

    def reduce(self, flags, offsets, n_reads):
        f = flags.numpy()
        off = offsets.numpy()
        high = np.array([(f[off[i]:off[i + 1]] == 1).sum() for i in range(n_reads)], dtype=np.int32)
        total = np.array([(f[off[i]:off[i + 1]] != 0).sum() for i in range(n_reads)], dtype=np.int32)
        return torch.from_numpy(high), torch.from_numpy(total)

    def synchronize(self):
        pass

    def stats(self):
        return {"unique_kmers": self.o.unique}


def make_reads(seed, n, lo, hi, alphabet=b"ACG"):
    rng = np.random.default_rng(seed)
    al = np.frombuffer(alphabet, dtype=np.uint8)
    return [al[rng.integers(0, len(al), rng.integers(lo, hi + 1))].tobytes() for _ in range(n)]


def concat(reads):
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return np.frombuffer(b"".join(reads), dtype=np.uint8).copy(), off


def worker(rank, world, port, k, depth, canonical, steps, q, max_message_bytes=256 << 20, ahead=False, stay_home=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nomalise_kmers_multi_large_amd.sharded import ShardedNormaliser
        # with small messages also run the start-up check of the exchange (3000 elements per peer, many rounds)
        sn = ShardedNormaliser(CheckerEngine(k, depth, canonical, stay_home), max_message_bytes=max_message_bytes, check_elems=3000 if max_message_bytes < 4096 else 0)
        assert sn.max_message_bytes == max_message_bytes
        out, batches = [], []
        for step in range(steps):
            reads = make_reads(100 * step + rank, 40 + 7 * rank, k, 90)   # ragged: ranks bring different amounts
            seq, off = concat(reads)
            batches.append((torch.from_numpy(seq), torch.from_numpy(off), len(reads), len(seq)))
        for step in range(steps):
            # with look-ahead the next step's keys are exchanged before this step's are counted
            then = batches[step + 1] if ahead and step + 1 < steps else None
            high, total, gathered = sn.process_step(*batches[step], then=then)
            if rank == 0:
                out.append([g.numpy().copy() for g in gathered])
        uniq = torch.tensor([sn.stats()["unique_kmers"], sn.eng.home_positions], dtype=torch.int64)
        dist.all_reduce(uniq)
        if rank == 0:
            q.put((out, int(uniq[0].item()), int(uniq[1].item())))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the last case exchanges in rounds of at most 1000 bytes per message (several rounds per step)
# and the last two send the next step's keys ahead of the current step's counting
# ... and the cases with stay_home keep saturated k-mers from travelling (bitmaps fed by the owners' lists after every step)
@pytest.mark.parametrize("world,k,depth,canonical,msg,ahead,stay_home", [
    (2, 9, 3, False, 256 << 20, False, False), (2, 21, 2, True, 256 << 20, False, False),
    (3, 11, 4, True, 256 << 20, False, False), (2, 11, 3, True, 1000, False, False),
    (3, 11, 4, True, 256 << 20, True, False), (2, 11, 3, True, 1000, True, False),
    (2, 7, 3, False, 256 << 20, False, True), (3, 7, 4, True, 256 << 20, True, True), (2, 9, 2, True, 1000, True, True)])
def test_sharded_exchange_equals_single_stream(oracle_lib, world, k, depth, canonical, msg, ahead, stay_home):
    steps = 4 if stay_home else 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, k, depth, canonical, steps, q, msg, ahead, stay_home)) for r in range(world)]
    for p in procs:
        p.start()
    got, uniq, stayed = q.get(timeout=120)
    assert (stayed > 0) == stay_home   # short k-mers over three letters repeat: some are saturated after the first step
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single stream, single table
    o = oracle_lib(k, depth, canonical)
    for step in range(steps):
        for rank in range(world):
            reads = make_reads(100 * step + rank, 40 + 7 * rank, k, 90)
            want = [o.read(r) for r in reads]
            np.testing.assert_array_equal(got[step][rank][0], [w[1] for w in want])
            np.testing.assert_array_equal(got[step][rank][1], [w[2] for w in want])
    assert uniq == o.unique   # shards are disjoint: their sizes add up to the single table's


def test_owner_rule_matches_library():
    from nomalise_kmers_multi_large_amd import binding
    lib = binding.load_library()
    rng = np.random.default_rng(0)
    keys = rng.integers(1, 1 << 62, 2000, dtype=np.uint64)
    for n in (1, 2, 3, 8):
        want = owner_np(keys, n)
        assert [lib.nkml_owner_of(int(x), n) for x in keys] == want.tolist()


def test_startup_check_notices_a_lossy_exchange_and_shrinks_the_rounds():
    """A transport that drops the second half of every message above 4000 bytes (what was seen with RCCL above 1024 MiB):
    the start-up check halves max_message_bytes until messages arrive whole; one that always loses data stops the job."""
    from nomalise_kmers_multi_large_amd.sharded import ShardedNormaliser
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        class Lossy(ShardedNormaliser):
            threshold = 4000

            def _all_to_all_once(self, out, inp, out_splits, in_splits):
                work = super()._all_to_all_once(out, inp, out_splits, in_splits)
                if work is not None:
                    work.wait()
                if inp.numel() * inp.element_size() > self.threshold:
                    out[out.numel() // 2:] = 0
                return None

        sn = Lossy(CheckerEngine(9, 3, False), max_message_bytes=12000, check_elems=2000)   # 16000 bytes per peer
        assert sn.max_message_bytes == 3000
        Lossy.threshold = 0
        with pytest.raises(RuntimeError):
            Lossy(CheckerEngine(9, 3, False), max_message_bytes=12000, check_elems=2000)
    finally:
        dist.destroy_process_group()


def failing_worker(rank, world, port, fault, q):
    """One step in which rank 1 alone fails (fault = "route": its routing raises, as for a read with a byte outside ACGTN;
    "count": its counting raises; "oversize": rank 0 receives more keys than it was sized for).  Every rank must come
    back from process_step with an exception - none may be left waiting in a collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nomalise_kmers_multi_large_amd.sharded import ShardedNormaliser

        class Faulty(CheckerEngine):
            def route(self, keys, n_shards):
                if fault == "route" and rank == 1:
                    raise ValueError("sequence byte outside ACGTN")
                return super().route(keys, n_shards)

            def count(self, recv_keys):
                if fault == "count" and rank == 1:
                    raise MemoryError("table shard cannot grow")
                return super().count(recv_keys)

        eng = Faulty(9, 3, False)
        if fault == "oversize" and rank == 0:
            eng.cap = 10
        sn = ShardedNormaliser(eng, check_elems=0)
        seq, off = concat(make_reads(rank, 30, 9, 60))
        try:
            sn.process_step(torch.from_numpy(seq), torch.from_numpy(off), len(off) - 1, len(seq))
            q.put((rank, "returned"))
        except Exception as e:
            q.put((rank, type(e).__name__ + ": " + str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fault", ["route", "count", "oversize"])
def test_a_rank_that_fails_stops_every_rank(fault):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=failing_worker, args=(r, world, port, fault, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))   # a rank left waiting in a collective would time out here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    culprit = 0 if fault == "oversize" else 1
    assert {"route": "ValueError", "count": "MemoryError", "oversize": "RuntimeError: received"}[fault] in got[culprit]
    assert "another rank stopped this step" in got[1 - culprit]
