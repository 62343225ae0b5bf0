"""Replicated-reads normaliser for two to four GPUs of one node: the table is sharded by k-mer as in
`sharded.py`, but instead of sending every k-mer to its owner, every GPU gets every read of the step (2-bit packed,
one all-gather) and counts only the k-mers of its own shard.  Per-read partial counters are then summed over the
ranks.  Nothing per k-mer crosses a link: 0.25 byte per base goes out once, against 5 bytes per k-mer of the
all-to-all exchange, which matters most on two GPUs, where all of it shares one xGMI link.  The price is that the
first partition pass reads world_size times the positions, so the all-to-all path wins again from about eight GPUs on,
and positions of one step must stay below 2^32 (world_size x bases per rank).

Results are those of one table reading the global stream (per step: rank 0's reads, then rank 1's, ...), like
`ShardedNormaliser`; the reference has no counterpart (its threads keep private tables, normalise_kmers_multi_large.c:2421).

Layout of one step on every rank: rank r's packed words sit in region r of `region` words (the largest word count of the
step, the same on all ranks); what is left of a region after its reads is zero = poly-A, which has no countable k-mer
(code 0 is skipped like in the reference's table, :929-1053 store_kmer callers), and is listed as filler reads so that
the read offsets stay one non-decreasing array.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import binding as B
from .phases import Phases

GAP_READS = 1024     # a region's unused tail is cut into at most this many filler reads ...
GAP_BASES = 4096     # ... of at least this many bases (one lane walks one read in the per-read reduction)
POSITION_LIMIT = (1 << 32) - (4 << 20)


def fits(ksize, depth, world, max_bases):
    """Whether a step of `world` ranks x max_bases fits the replicated-reads mode."""
    region = (max_bases + 15) // 16 * 16
    return 14 <= ksize <= 15 and depth <= 127 and world * region < POSITION_LIMIT


def step_layout(n_reads, n_bases):
    """Host-side plan of one step from every rank's (reads, bases): region size in words, per rank the number of
    filler reads and the index of its first read in the global read list, and the total number of reads."""
    region = max(1, max((int(b) + 15) // 16 for b in n_bases))   # at least one word, so that a step nobody brings reads to stays an ordinary step
    fillers, first, at = [], [], 0
    for r, b in zip(n_reads, n_bases):
        gap = region * 16 - int(b)
        e = min(GAP_READS, max(1, -(-gap // GAP_BASES)))
        fillers.append(e)
        first.append(at)
        at += int(r) + e
    return region, fillers, first, at


def global_offsets(local_offsets, n_reads, n_bases, region, fillers):
    """The step's read offsets (int64 tensor on the device of the inputs) from each rank's own offsets
    (local_offsets[r][: n_reads[r] + 1], any integer dtype): rank r's reads moved to region r, then its filler reads."""
    parts = []
    dev = local_offsets[0].device
    for r, (off, nr, nb, e) in enumerate(zip(local_offsets, n_reads, n_bases, fillers)):
        base = r * region * 16
        parts.append(off[: nr + 1].to(torch.int64) + base)
        if e > 1:
            gap = region * 16 - nb
            j = torch.arange(1, e, dtype=torch.int64, device=dev)
            parts.append(base + nb + (j * gap) // e)
    parts.append(torch.tensor([len(local_offsets) * region * 16], dtype=torch.int64, device=dev))
    return torch.cat(parts)


class HipReplicaEngine:
    """The HIP library on one GPU for the replicated-reads mode: packs this rank's reads, counts a whole step against
    this rank's table shard."""

    def __init__(self, ksize, depth, coverage, canonical, device, max_reads, max_bases, world, rank, profile=False):
        from .normaliser import KmerNormaliser
        if not fits(ksize, depth, world, max_bases):
            raise ValueError("replicated-reads mode needs k = 14 or 15, depth <= 127 and world x bases per step below 2^32")
        self.dev = torch.device("cuda", device)
        self.max_region = (max_bases + 15) // 16
        self.max_reads = max_reads
        all_reads = world * (max_reads + GAP_READS)
        self.nk = KmerNormaliser(ksize, depth, coverage, canonical, device=device, max_batch_reads=all_reads,
                                 max_batch_bases=world * self.max_region * 16, profile=profile, flags=B.NKML_F_COUNT_PARTITIONED)
        self.nk.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.nk.set_shard(rank, world)
        # two sets of exchange buffers: the next step's reads arrive while this step's are counted
        self.words = [torch.zeros(self.max_region, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.words_all = [torch.zeros(world * self.max_region, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.counters = torch.zeros(2 * all_reads, dtype=torch.int32, device=self.dev)

    def pack(self, seq, n_bases, region, buffer=0):
        """This rank's reads as `region` packed words (zero beyond the reads)."""
        nw = (n_bases + 15) // 16
        if region > self.max_region or nw > region:
            raise RuntimeError("step of %d words per rank, sized for %d" % (region, self.max_region))
        words = self.words[buffer]
        if n_bases:   # a rank may have nothing to bring to a step
            self.nk.pack_device(seq.data_ptr(), n_bases, words.data_ptr())
        if region > nw:
            words[nw:region].zero_()
        return words[:region]

    def gather_buffer(self, n_words, buffer=0):
        return self.words_all[buffer][:n_words]

    def process_packed(self, words_all, offsets, n_reads, n_bases):
        """Returns a [2, n_reads] view (high, total): this shard's share of every read of the step."""
        out = self.counters[: 2 * n_reads].view(2, n_reads)
        self.nk.process_packed_device(words_all.data_ptr(), offsets.data_ptr(), n_reads, n_bases, out[0].data_ptr(), out[1].data_ptr())
        return out

    def decide(self, high, total, n_units, paired, keep):
        self.nk.decide_device(high.data_ptr(), total.data_ptr(), n_units, paired, keep.data_ptr())

    def synchronize(self):
        self.nk.synchronize()

    def reset(self):
        self.nk.reset()

    def stats(self):
        return self.nk.stats()


class ReplicatedReadsNormaliser:
    """Same face as ShardedNormaliser: one step = every rank's batch, processed as one global batch in rank-major order."""

    def __init__(self, engine, group=None, stage_through_host=False, check_words=None, root_only=False, time_phases=False):
        self.eng = engine
        self.phases = Phases(time_phases, getattr(engine, "dev", None))
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # rehearsal aid only (several ranks on one GPU under gloo): collectives bounce through host memory
        self.stage_through_host = stage_through_host
        self.root_only = root_only   # only rank 0 needs the summed counters (it takes the keep/skip decisions)
        self._buffer = 0      # which of the engine's two sets of exchange buffers the next step uses
        self._ahead = None    # the exchange started for the step announced by `then`
        if check_words is None:
            check_words = int(getattr(engine, "max_region", 0))
        if check_words:
            self.check_collectives(check_words, getattr(engine, "dev", "cpu"))

    def check_collectives(self, n, device):
        """The two collectives of a step at the largest size this job uses (n words per rank), with known content: a
        stack that delivers something else than what was sent must stop the job, not change its results."""
        W, me = self.world, self.rank
        mine = torch.arange(n, dtype=torch.int32, device=device) + 7919 * me
        out = torch.zeros(W * n, dtype=torch.int32, device=device)
        work = self._gather_equal(out, mine, async_op=True)
        if work is not None:
            work.wait()
        ok = all(bool(torch.equal(out[p * n:(p + 1) * n], mine + 7919 * (p - me))) for p in range(W))
        ones = torch.ones(min(2 * n, 1 << 26), dtype=torch.int32, device=device)
        self._sum(ones)
        ok = ok and bool((ones == W).all())
        cdev = "cpu" if self.stage_through_host or str(device) == "cpu" else device
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if not int(flag.item()):
            raise RuntimeError("all-gather / sum of %d words per rank does not deliver what was sent" % n)

    def _gather_equal(self, out, inp, async_op=False):
        """all-gather of equal pieces; with async_op the returned handle's wait() is owed before `out` is used."""
        if self.stage_through_host and inp.is_cuda:
            self.eng.synchronize()
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, inp.cpu(), group=self.group)
            out.copy_(h)
            return None
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def _agree(self, err, like):
        """Every rank calls this after the stage that can fail on one rank alone (err = its exception or None): all raise,
        or none - a rank left alone with its exception would leave its peers waiting in the next collective."""
        cdev = "cpu" if self.stage_through_host or not like.is_cuda else like.device
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            if err is not None:
                raise err
            raise RuntimeError("another rank stopped this step (a read with a byte outside ACGTN, or a batch beyond the sizes it was set up for)")

    def _sum(self, t, everywhere=True):
        """Sum over the ranks, in place; with everywhere=False only rank 0 is owed the result (half the traffic)."""
        op = (lambda v: dist.all_reduce(v, group=self.group)) if everywhere else (lambda v: dist.reduce(v, dst=0, group=self.group))
        if self.stage_through_host and t.is_cuda:
            self.eng.synchronize()
            h = t.cpu()
            op(h)
            t.copy_(h)
        else:
            op(t)

    def _sum_counters(self, counters, narrow):
        """counters: [2, reads] int32 partial (high, total) of this shard -> summed over the shards.  narrow: no read has
        65536 k-mers or more, so both numbers travel in one 32-bit word (their sums cannot carry into each other)."""
        everywhere = not self.root_only
        if not narrow:
            self._sum(counters, everywhere)
            return
        both = (counters[0] << 16) | counters[1]
        self._sum(both, everywhere)
        counters[0] = (both >> 16) & 0xffff
        counters[1] = both & 0xffff

    def _start_exchange(self, seq, offsets, n_reads, n_bases):
        """Everything of a step that does not need the table: sizes, packing, the two all-gathers (left in flight)."""
        eng, W = self.eng, self.world
        cdev = "cpu" if self.stage_through_host or not seq.is_cuda else seq.device
        longest = int((offsets[1: n_reads + 1] - offsets[:n_reads]).max()) if n_reads else 0
        with self.phases.host("sizes (host round trip)"):
            meta = torch.empty(3 * W, dtype=torch.int64, device=cdev)
            self._gather_equal(meta, torch.tensor([n_reads, n_bases, longest], dtype=torch.int64, device=cdev))
            meta = meta.tolist()
        x = {"key": (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases), "reads": meta[0::3], "bases": meta[1::3],
             "narrow": max(meta[2::3]) < 65536}
        x["region"], x["fillers"], x["first"], x["total_reads"] = step_layout(x["reads"], x["bases"])
        self._buffer ^= 1
        x["words_all"] = eng.gather_buffer(W * x["region"], self._buffer)
        with self.phases.dev("pack"):
            mine_words = eng.pack(seq, n_bases, x["region"], self._buffer)
        self.phases.add_bytes("packed reads received from other ranks", 4 * x["region"] * (W - 1))
        # read offsets travel as 32-bit numbers local to the rank's region
        x["slot"] = max(x["reads"]) + 1
        mine = torch.zeros(x["slot"], dtype=torch.int32, device=seq.device)
        mine[: n_reads + 1] = offsets[: n_reads + 1]
        x["off_all"] = torch.empty(W * x["slot"], dtype=torch.int32, device=seq.device)
        x["works"] = [self._gather_equal(x["words_all"], mine_words, async_op=True), self._gather_equal(x["off_all"], mine, async_op=True)]
        x["alive"] = (mine, mine_words, seq, offsets)   # nothing the collectives read may be freed before they are waited for
        return x

    def process_step(self, seq, offsets, n_reads, n_bases, then=None):
        """seq/offsets: this rank's reads on the device.  Returns (high, total) of this rank's reads and, on rank 0, the
        list of [2, n_reads_r] counters of every rank's reads (views of a buffer the next step overwrites); with
        root_only the other ranks get (None, None, None): the counters are reduced to rank 0 alone.
        then = (seq, offsets, n_reads, n_bases) of the step that follows, if the caller has it already: its reads are
        packed and exchanged while this step is counted.  All ranks must pass `then` in the same steps, and the next
        call must be for exactly that batch."""
        eng, W = self.eng, self.world
        key = (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases)
        x, self._ahead = self._ahead, None
        if x is not None and x["key"] != key:
            raise RuntimeError("process_step was announced another batch by the previous call's `then`")
        if x is None:
            x = self._start_exchange(seq, offsets, n_reads, n_bases)
        ph = self.phases
        with ph.dev("all-gather of packed reads (not hidden)"):
            for w in x["works"]:
                if w is not None:
                    w.wait()
        slot = x["slot"]
        goff = global_offsets([x["off_all"][r * slot:(r + 1) * slot] for r in range(W)], x["reads"], x["bases"], x["region"], x["fillers"])
        if then is not None:
            self._ahead = self._start_exchange(*then)
        err = None
        try:
            with ph.dev("count (own shard of every rank's reads)"):
                counters = eng.process_packed(x["words_all"], goff, x["total_reads"], W * x["region"] * 16)
        except Exception as e:   # e.g. a byte outside ACGTN among this rank's reads: only this rank's context knows
            err = e
        with ph.host("agree (host round trip)"):
            self._agree(err, seq)
        with ph.dev("sum of per-read counters"):
            self._sum_counters(counters, x["narrow"])
        ph.add_bytes("per-read counters sent", (4 if x["narrow"] else 8) * x["total_reads"])
        ph.step_done()
        if self.root_only and self.rank != 0:
            return None, None, None
        per_rank = [counters[:, x["first"][r]: x["first"][r] + x["reads"][r]] for r in range(W)]
        own = per_rank[self.rank]
        return own[0], own[1], (per_rank if self.rank == 0 else None)

    def synchronize(self):
        self.eng.synchronize()

    def reset(self):
        self.eng.reset()
        self.phases.reset()

    def phase_report(self):
        return self.phases.report()

    def stats(self):
        return self.eng.stats()


def make_multi_gpu_normaliser(ksize, depth, coverage, canonical, device, max_reads, max_bases, profile=False,
                              stage_through_host=False, mode=None, group=None, root_only=False, time_phases=False):
    """The multi-GPU normaliser that suits the job.  Three modes:
      "replicated": packed reads all-gathered, every GPU extracts and counts its shard of the k-mers (this file); per-rank
                    compute of a 2 M-pair step at k=15 measured at 10.5 / 13.3 / 15.9 ms on 2 / 3 / 4 GPUs: chosen for two;
      "bucketed":   every GPU partitions its own k-mers by bucket, unsaturated ones travel to the bucket's owner as 16-bit
                    codes (bucketed.py; k <= 15, depth <= 16383): about 6 ms of compute per rank and 2 M-pair step whatever
                    the number of GPUs, 0.3-0.7 GB per rank and step on the links: chosen from three GPUs on (by estimate
                    for three and four: no such node was at hand);
      "exchange":   k-mers routed by a hash of the k-mer and counted at the owner with the full pipeline (sharded.py): every k.
    `mode` or the environment variable NKML_MULTI_GPU overrides the choice.  Returns (normaliser, engine, mode)."""
    import os
    from .sharded import HipShardEngine, ShardedNormaliser
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    can_bucket = ksize <= 15 and 2 <= depth <= 16383 and ((1 << (2 * ksize)) >> 14) >= world
    if not (mode or os.environ.get("NKML_MULTI_GPU")):
        if world <= 2 and fits(ksize, depth, world, max_bases):
            mode = "replicated"
        elif can_bucket:
            mode = "bucketed"
        else:
            mode = "replicated" if world <= 4 and fits(ksize, depth, world, max_bases) else "exchange"
    mode = mode or os.environ.get("NKML_MULTI_GPU")
    if mode == "replicated":
        engine = HipReplicaEngine(ksize, depth, coverage, canonical, device, max_reads, max_bases, world, rank, profile=profile)
        return ReplicatedReadsNormaliser(engine, group=group, stage_through_host=stage_through_host, root_only=root_only,
                                         time_phases=time_phases), engine, mode
    if mode == "bucketed":
        from .bucketed import BucketedNormaliser, HipBucketEngine
        engine = HipBucketEngine(ksize, depth, coverage, canonical, device, max_reads, max_bases, world, rank, profile=profile)
        return BucketedNormaliser(engine, group=group, stage_through_host=stage_through_host, time_phases=time_phases), engine, mode
    if mode != "exchange":
        raise ValueError("NKML_MULTI_GPU must be 'replicated', 'bucketed' or 'exchange', not %r" % mode)
    engine = HipShardEngine(ksize, depth, coverage, canonical, device, max_reads=max_reads, max_bases=max_bases, profile=profile, world=world)
    return ShardedNormaliser(engine, group=group, stage_through_host=stage_through_host, time_phases=time_phases), engine, mode
