"""Bucket-ordered exchange over several GPUs of one node (k <= 15): one process per GPU, torch.distributed (RCCL) for the
exchange, the HIP library for everything else.  Same results as the single-stream run, like sharded.py, with less work
and fewer bytes on the links:

  1. every rank partitions its own reads' k-mers by table bucket (the two partition passes of the single-GPU path)
  2. k-mers known to be saturated (count at the owner >= depth - 1: every further occurrence is "high") are answered on the
     spot; the others travel to the bucket's owner as 16-bit codes within their bucket, bucket after bucket, each bucket in
     stream order, together with one count per bucket
  3. the owner strings the ranks' runs of a bucket together in rank order - the bucket's k-mers of the step's global stream
     (rank 0's reads, then rank 1's, ...) - and runs the bucket kernels on them: no partition pass at the owner
  4. flags travel back; every rank takes them home through its own reverse passes and sums them per read
  5. owners announce the k-mers that became saturated in the step; every rank marks them in its bitmap
  6. per-read counters are gathered on rank 0 (keep / skip decisions, ordered write-back)

The reference has nothing like this (its threads keep private tables and never merge them, normalise_kmers_multi_large.c:2421).
The engine object does the device work (`HipBucketEngine`); tests drive the same logic over gloo with a CPU checker engine.
"""
import torch
import torch.distributed as dist

from .sharded import ShardedNormaliser


def first_place(buckets, world, rank):
    """Rank r owns the buckets at places [first_place(r), first_place(r + 1)) of the deal (nkml_xb_enable)."""
    return buckets * rank // world


def _as_bytes(t):
    """the same memory as uint8 (an empty tensor may come with strides that cannot be viewed: any empty one will do)"""
    return t.view(torch.uint8) if t.numel() else torch.empty(0, dtype=torch.uint8, device=t.device)


class HipBucketEngine:
    """Stage-by-stage face of the HIP library's bucket-ordered exchange on one GPU, holding torch tensors for the buffers.
    Three library contexts: two that play the sender (pack, partition, compact; the way home) for alternate steps, so that a
    step's k-mers can be partitioned and sent while the step before it is still being counted and brought home, and one that
    owns this rank's buckets of the table.  (The senders are created without a table, the owner with partition buffers for
    next to nothing.)"""

    MAX_NEW_SATURATED = 1 << 23

    def __init__(self, ksize, depth, coverage, canonical, device, max_reads, max_bases, world, rank, profile=False, slack=1.5, stay_home=True):
        import os
        from .normaliser import KmerNormaliser
        from . import binding as B
        if ksize > 15:
            raise ValueError("the bucket-ordered exchange needs the direct table (k <= 15)")
        self.dev = torch.device("cuda", device)
        self.world, self.rank = world, rank
        self.cap = int(max_bases * slack) // 16 * 16 + 64   # keys this rank can receive in one step
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.senders = []
        for _ in range(2):
            nk = KmerNormaliser(ksize, depth, coverage, canonical, device=device, max_batch_reads=max_reads, max_batch_bases=max_bases,
                                flags=B.NKML_F_COUNT_PARTITIONED | B.NKML_F_NO_TABLE)
            nk.set_stream(stream)
            nk.xb_enable(world, rank, 16)
            self.senders.append(nk)
        self.nk = KmerNormaliser(ksize, depth, coverage, canonical, device=device, max_batch_reads=16, max_batch_bases=4096,
                                 profile=profile, flags=B.NKML_F_COUNT_PARTITIONED)
        self.nk.set_stream(stream)
        self.nk.xb_enable(world, rank, self.cap)
        self.buckets = max(1, (1 << (2 * ksize)) >> 14)
        env = os.environ.get("NKML_EXCHANGE_STAY_HOME")
        self.stay_home = bool(stay_home) if env not in ("0", "1") else env == "1"
        if self.stay_home:
            for nk in self.senders + [self.nk]:
                nk.sat_enable()
            self.sat_codes = torch.zeros(self.MAX_NEW_SATURATED, dtype=torch.int32, device=self.dev)
        npos = (max_bases + 15) // 16 * 16
        self.max_positions = npos
        self.key_dtype = torch.int16
        self.mine = first_place(self.buckets, world, rank + 1) - first_place(self.buckets, world, rank)
        # two sets of what a step's exchange touches, one per sender
        self.send_keys2 = [torch.zeros(npos, dtype=torch.int16, device=self.dev) for _ in range(2)]
        self.send_counts2 = [torch.zeros(self.buckets, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.recv_keys2 = [torch.zeros(self.cap, dtype=torch.int16, device=self.dev) for _ in range(2)]
        self.recv_counts2 = [torch.zeros(world * self.mine, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.recv_flags = torch.zeros(self.cap, dtype=torch.uint8, device=self.dev)
        self.back_flags = torch.zeros(npos, dtype=torch.uint8, device=self.dev)
        self.high = torch.zeros(max_reads, dtype=torch.int32, device=self.dev)
        self.total = torch.zeros(max_reads, dtype=torch.int32, device=self.dev)
        self.home_positions = self.home_total = 0
        self.which = 0

    def use_buffers(self, which):
        """Which sender (context and exchange buffers) xb_front / xb_recv_buffers work with from now on; xb_finish takes the
        sender of the step it finishes as an argument."""
        self.which = which

    def xb_front(self, seq, offsets, n_reads, n_bases):
        """-> (keys to send, owner after owner; per-bucket counts by place; keys per owner)"""
        w = self.which
        totals, home = self.senders[w].xb_front(seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases, self.send_keys2[w].data_ptr(),
                                                self.send_counts2[w].data_ptr(), self.world)
        self.home_positions = home
        self.home_total += home
        return self.send_keys2[w][: sum(totals)], self.send_counts2[w], totals

    def xb_recv_buffers(self, n):
        if n > self.cap:
            raise RuntimeError("received %d keys, more than the %d this rank was sized for" % (n, self.cap))
        return self.recv_keys2[self.which][:n], self.recv_counts2[self.which]

    def xb_count(self, recv_keys, recv_counts):
        n = recv_keys.numel()
        self.nk.xb_count(recv_keys.data_ptr(), recv_counts.data_ptr(), n, self.recv_flags.data_ptr())
        return self.recv_flags[:n]

    def xb_back_buffer(self, n):
        return self.back_flags[:n]

    def xb_finish(self, back_flags, offsets, n_reads, which):
        self.senders[which].xb_finish(back_flags.data_ptr(), offsets.data_ptr(), n_reads, self.high.data_ptr(), self.total.data_ptr())
        return self.high[:n_reads], self.total[:n_reads]

    def sat_collect(self):
        """k-mers of this rank's buckets that reached depth - 1 since the last call (device tensor of codes; this rank's own
        senders know them from here on); None if no saturation bitmap is kept.  Waits for the device."""
        if not self.stay_home:
            return None
        n = self.nk.sat_collect(self.sat_codes.data_ptr(), self.sat_codes.numel())
        new = self.sat_codes[:n]
        self.sat_apply(new)
        return new

    def sat_apply(self, codes):
        if self.stay_home and codes.numel():
            for nk in self.senders:
                nk.sat_apply(codes.data_ptr(), codes.numel())

    def decide(self, high, total, n_units, paired, keep):
        self.nk.decide_device(high.data_ptr(), total.data_ptr(), n_units, paired, keep.data_ptr())

    def synchronize(self):
        self.nk.synchronize()

    def reset(self):
        for nk in self.senders + [self.nk]:
            nk.reset()
        self.home_positions = self.home_total = 0

    def stats(self):
        """the owner's table statistics; occurrences the senders answered themselves count as counted and saturated"""
        st = self.nk.stats()
        for nk in self.senders:
            own = nk.stats()
            st["kmers_counted"] += own["kmers_counted"]
            st["kmers_saturated"] += own["kmers_saturated"]
        return st

    def close(self):
        for nk in self.senders + [self.nk]:
            nk.close()


class BucketedNormaliser(ShardedNormaliser):
    """One step = every rank's batch of reads, processed as one global batch (rank-major stream order)."""

    def __init__(self, engine, group=None, gather_to_root=True, stage_through_host=False, max_message_bytes=768 << 20, check_elems=0, time_phases=False):
        super().__init__(engine, group=group, gather_to_root=gather_to_root, stage_through_host=stage_through_host,
                         max_message_bytes=max_message_bytes, check_elems=check_elems, time_phases=time_phases)

    def _start_step(self, seq, offsets, n_reads, n_bases):
        """Everything of a step that does not need the table: partition, compaction, sizes, the exchange of counts and keys
        (left in flight)."""
        eng, W, me, ph = self.eng, self.world, self.rank, self.phases
        B = eng.buckets
        self._buffer ^= 1
        if hasattr(eng, "use_buffers"):
            eng.use_buffers(self._buffer)
        x = {"key": (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases), "which": self._buffer}
        err = None
        try:
            with ph.dev("partition + pack"):
                send_keys, send_counts, totals = eng.xb_front(seq, offsets, n_reads, n_bases)
        except Exception as e:   # this rank cannot go on, but its peers have to hear of it before anyone raises
            err, send_keys, send_counts, totals = e, None, None, [0] * W
        with ph.host("sizes (host round trip)"):
            recv_totals, peak, stop, too_many = self._exchange_counts(totals, seq, err is not None, getattr(eng, "cap", None), n_reads)
        x["n_reads_all"] = self._n_reads_all
        if stop:
            if err is not None:
                raise err
            if too_many:
                raise RuntimeError("received %d keys, more than the %d this rank was sized for" % (sum(recv_totals), eng.cap))
            raise RuntimeError("another rank stopped this step (a read with a byte outside ACGTN, or more keys than it was sized for)")
        x["totals"], x["recv_totals"], x["peak"] = totals, recv_totals, peak
        n_recv = sum(recv_totals)
        if W == 1:
            x["recv_keys"], x["recv_counts"], x["owed"] = send_keys, send_counts, []
        else:
            x["recv_keys"], x["recv_counts"] = eng.xb_recv_buffers(n_recv)
            mine = [first_place(B, W, r + 1) - first_place(B, W, r) for r in range(W)]
            # per-bucket counts: to owner r its buckets' part (by place); from every rank one count per bucket of mine
            x["owed"] = self._all_to_all_start(x["recv_counts"], send_counts, [mine[me]] * W, mine, max(mine))
            # keys as bytes (two per key): every backend moves uint8
            x["owed"] += self._all_to_all_start(_as_bytes(x["recv_keys"]), _as_bytes(send_keys), [2 * t for t in recv_totals], [2 * t for t in totals], 2 * peak)
            ph.add_bytes("keys sent to other ranks", 2 * (sum(totals) - totals[me]) + 4 * (B - mine[me]))
            ph.add_bytes("keys received from other ranks", 2 * (n_recv - recv_totals[me]) + 4 * mine[me] * (W - 1))
        x["alive"] = (send_keys, send_counts, seq, offsets)
        return x

    def process_step(self, seq, offsets, n_reads, n_bases, then=None):
        """seq/offsets: this rank's reads.  Returns (high, total) of this rank's reads, and on rank 0 additionally the
        gathered (high_all, total_all) per rank.  then = (seq, offsets, n_reads, n_bases) of the step that follows, if the
        caller has it already: its k-mers are partitioned and sent while this step's are counted (engines with two senders).
        All ranks must pass `then` in the same steps, and the next call must be for exactly that batch."""
        eng, W, me, ph = self.eng, self.world, self.rank, self.phases
        key = (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases)
        x, self._ahead = self._ahead, None
        if x is not None and x["key"] != key:
            raise RuntimeError("process_step was announced another batch by the previous call's `then`")
        if x is None:
            x = self._start_step(seq, offsets, n_reads, n_bases)
        with ph.dev("key exchange (not hidden)"):
            self._all_to_all_finish(x["owed"])
        if then is not None and hasattr(eng, "use_buffers"):
            self._ahead = self._start_step(*then)
        totals, recv_totals = x["totals"], x["recv_totals"]
        n_recv = sum(recv_totals)
        err = None
        try:
            with ph.dev("count"):
                recv_flags = eng.xb_count(x["recv_keys"], x["recv_counts"])
        except Exception as e:
            err = e
        with ph.host("agree (host round trip)"):
            self._agree_and_share_saturated(err, seq)
        if W == 1:
            back = recv_flags
        else:
            back = eng.xb_back_buffer(sum(totals))
            with ph.dev("flag exchange"):
                self._all_to_all(back, recv_flags, totals, recv_totals, x["peak"])
            ph.add_bytes("flags sent to other ranks", n_recv - recv_totals[me])
        with ph.dev("flags home + per-read sums"):
            high, total = eng.xb_finish(back, offsets, n_reads, x["which"])
        gathered = None
        if self.gather_to_root:
            with ph.dev("gather to rank 0"):
                gathered = self._p2p_gather(torch.stack([high, total]), x["n_reads_all"])
        ph.step_done()
        return high, total, gathered
