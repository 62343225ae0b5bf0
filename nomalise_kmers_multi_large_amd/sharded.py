"""K-mer-sharded normaliser over several GPUs of one node: one process per GPU, torch.distributed (RCCL) for
the exchange, the HIP library for everything else.

The reference has nothing like this (its threads keep private tables and never merge them,
normalise_kmers_multi_large.c:2421); read-sharding with private tables would change the results, so the table
is sharded by k-mer instead and results stay identical to the single-stream run:

  global stream order of one step = rank 0's reads, then rank 1's, ... (each rank brings a contiguous slice)
  1. every rank k-merises its reads (keys in stream order)
  2. keys are stably partitioned by owning rank and exchanged with one all-to-all; what an owner receives is
     ordered by source rank, and within a source by stream position, i.e. it is the global stream order
     restricted to the keys this rank owns - exactly what the counting pass needs for exact ranks
  3. the owner counts them against its shard of the table (same kernels as on one GPU) and gets one flag per key
  4. the flags travel back with the reverse all-to-all and are put back at their base positions
  5. per-read counters are reduced where the reads live and gathered on rank 0, which takes the keep/skip
     decisions for the whole step (ordered write-back happens there)

Saturated k-mers stay home (engines that offer sat_collect / sat_apply; the HIP engine does at k <= 15): once a k-mer's
count at its owner has reached depth - 1, every later occurrence is "high" whatever the owner would add, so it need not
travel.  Every rank keeps a bitmap of k-mers known to be that far; routing answers their occurrences on the spot.  After
counting a step, every owner lists the k-mers that got there in it and the lists go round in one all-gather.  A bitmap
may lag (with look-ahead the next step is routed before this one is counted): a k-mer without its bit travels and is
counted as before, so results do not depend on when a bit arrives.

The engine object does the device work; `HipShardEngine` is the product one.  Tests drive the same exchange
logic over gloo with a CPU checker engine.
"""
import numpy as np
import torch
import torch.distributed as dist

from .phases import Phases


class HipShardEngine:
    """Stage-by-stage face of the HIP library on one GPU, holding torch tensors for the buffers."""

    MAX_NEW_SATURATED = 1 << 23   # k-mers one step can announce as newly saturated (the rest is announced never: see sat_collect)

    def __init__(self, ksize, depth, coverage, canonical, device, max_reads, max_bases, profile=False, slack=1.5, world=1,
                 ranked=None, stay_home=None):
        """ranked: route with 16 bits per base position instead of 4 bytes per sent key (None = wherever the world has at most
        32 ranks); stay_home: keep the saturation bitmap (k <= 15 on the partitioned direct table; None = no: here the
        routing looks the bits up in stream order, one random access per k-mer - measured at 9.6 ms per 544 M k-mers, more
        than the occurrences it spares cost; the bucket-ordered exchange of bucketed.py looks them up bucket by bucket);
        the environment variables NKML_EXCHANGE_RANKED / NKML_EXCHANGE_STAY_HOME = 0 / 1 override both."""
        import os
        from .normaliser import KmerNormaliser
        self.dev = torch.device("cuda", device)
        self.cap = int(max_bases * slack) // 16 * 16 + 64
        self.nk = KmerNormaliser(ksize, depth, coverage, canonical, device=device, max_batch_reads=max_reads,
                                 max_batch_bases=self.cap, profile=profile)
        self.nk.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.key_dtype = torch.int32 if self.nk.key_bytes == 4 else torch.int64
        npos = (max_bases + 15) // 16 * 16
        self.max_positions = npos
        env = os.environ.get("NKML_EXCHANGE_RANKED")
        self.ranked = (world <= 32) if ranked is None else bool(ranked)
        if env in ("0", "1"):
            self.ranked = env == "1"
        env = os.environ.get("NKML_EXCHANGE_STAY_HOME")
        want_home = False if stay_home is None else bool(stay_home)
        if env in ("0", "1"):
            want_home = env == "1"
        self.stay_home = False
        if want_home and self.ranked and ksize <= 15 and self.nk.stats()["partitioned"]:
            self.nk.sat_enable()
            self.stay_home = True
            self.sat_codes = torch.zeros(self.MAX_NEW_SATURATED, dtype=torch.int32, device=self.dev)
        self.home_positions = self.home_total = 0   # base positions the bitmap answered: in the last step routed / since the start
        self._keys = None   # key array of the stage-by-stage form (kmerise + route), made when first asked for
        # two sets of what the key exchange touches: the next step's keys travel while this step's are counted
        self.send_keys2 = [torch.zeros(npos, dtype=self.key_dtype, device=self.dev) for _ in range(2)]
        if self.ranked:   # per base position: shard and rank in the routing tile; per routing tile and shard: where its run starts
            tiles = (npos + 1023) // 1024
            self.send_src2 = [(torch.zeros(tiles * 1024, dtype=torch.int16, device=self.dev), torch.zeros(tiles * world + 1, dtype=torch.int32, device=self.dev))
                              for _ in range(2)]
        else:
            self.send_src2 = [torch.zeros(npos, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self._plain_src = None   # source positions of the stage-by-stage route() when the exchange itself runs ranked
        self.recv_keys2 = [torch.zeros(self.cap, dtype=self.key_dtype, device=self.dev) for _ in range(2)]
        self.send_keys, self.send_src, self.recv_keys = self.send_keys2[0], self.send_src2[0], self.recv_keys2[0]
        self.world = world
        self.flags = torch.zeros(npos, dtype=torch.uint8, device=self.dev)
        self.recv_flags = torch.zeros(self.cap, dtype=torch.uint8, device=self.dev)
        self.back_flags = torch.zeros(npos, dtype=torch.uint8, device=self.dev)
        self.high = torch.zeros(max_reads, dtype=torch.int32, device=self.dev)
        self.total = torch.zeros(max_reads, dtype=torch.int32, device=self.dev)

    @property
    def keys(self):
        if self._keys is None:
            self._keys = torch.zeros(self.max_positions, dtype=self.key_dtype, device=self.dev)
        return self._keys

    def kmerise(self, seq, offsets, n_reads, n_bases):
        self.nk.kmerise_device(seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases, self.keys.data_ptr())
        return self.keys[: (n_bases + 15) // 16 * 16]

    def route(self, keys, n_shards):
        src = self.send_src
        if self.ranked:   # the stage-by-stage form always works with source positions
            if self._plain_src is None:
                self._plain_src = torch.zeros(self.max_positions, dtype=torch.int32, device=self.dev)
            src = self._plain_src
        counts = self.nk.route_keys_device(keys.data_ptr(), keys.numel(), n_shards, self.send_keys.data_ptr(), src.data_ptr())
        n = sum(counts)
        return self.send_keys[:n], src[:n], counts

    def use_buffers(self, which):
        """Which of the two sets of exchange buffers kmerise_route / route / recv_buffer hand out from now on."""
        self.send_keys, self.send_src, self.recv_keys = self.send_keys2[which], self.send_src2[which], self.recv_keys2[which]

    def kmerise_route(self, seq, offsets, n_reads, n_bases, n_shards):
        """kmerise + route in one go: the k-mers are recomputed from the packed reads by the routing passes
        themselves, no key array is written.  Returns (send_keys, send_src, counts, n_positions)."""
        if self.ranked:
            if n_shards != self.world:
                raise RuntimeError("the engine was sized for %d ranks, not %d" % (self.world, n_shards))
            if n_reads == 0:
                return self.send_keys[:0], self.send_src, [0] * n_shards, 0
            meta, tile_off = self.send_src
            counts, home = self.nk.route_reads_ranked(seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases, n_shards,
                                                      self.send_keys.data_ptr(), meta.data_ptr(), tile_off.data_ptr())
            self.home_positions = home          # of this step
            self.home_total += home
            return self.send_keys[:sum(counts)], self.send_src, counts, (n_bases + 15) // 16 * 16
        if n_reads == 0:   # a rank may have nothing to bring to a step
            return self.send_keys[:0], self.send_src[:0], [0] * n_shards, 0
        counts = self.nk.route_reads_device(seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases, n_shards,
                                            self.send_keys.data_ptr(), self.send_src.data_ptr())
        n = sum(counts)
        return self.send_keys[:n], self.send_src[:n], counts, (n_bases + 15) // 16 * 16

    def recv_buffer(self, n):
        if n > self.cap:
            raise RuntimeError("received %d keys, more than the %d this rank was sized for" % (n, self.cap))
        return self.recv_keys[:n]

    def count(self, recv_keys):
        """recv_keys: what recv_buffer returned, filled (the buffer behind it is padded to a multiple of 16 keys here)."""
        n = recv_keys.numel()
        n16 = (n + 15) // 16 * 16
        # a world of one counts its keys where routing left them: the send buffers are candidates too
        whole = next(b for b in self.recv_keys2 + self.send_keys2 if b.data_ptr() == recv_keys.data_ptr()) if n else self.recv_keys
        if n16 > n:
            whole[n:n16].zero_()
        self.nk.count_keys_device(whole.data_ptr(), n16, self.recv_flags.data_ptr())
        return self.recv_flags[:n]

    def back_buffer(self, n):
        return self.back_flags[:n]

    def unroute(self, back_flags, send_src, n_positions):
        if isinstance(send_src, tuple):
            meta, tile_off = send_src
            self.nk.unroute_ranked(back_flags.data_ptr(), meta.data_ptr(), tile_off.data_ptr(), n_positions, self.flags.data_ptr())
            return self.flags[:n_positions]
        self.nk.unroute_flags_device(back_flags.data_ptr(), send_src.data_ptr(), back_flags.numel(), self.flags.data_ptr(), n_positions)
        return self.flags[:n_positions]

    def sat_collect(self):
        """k-mers of this rank's table that reached depth - 1 since the last call (device tensor of codes); None if this
        engine keeps no saturation bitmap.  Waits for the device."""
        if not self.stay_home:
            return None
        n = self.nk.sat_collect(self.sat_codes.data_ptr(), self.sat_codes.numel())
        return self.sat_codes[:n]

    def sat_apply(self, codes):
        if self.stay_home and codes.numel():
            self.nk.sat_apply(codes.data_ptr(), codes.numel())

    def reduce(self, flags, offsets, n_reads):
        if n_reads == 0:
            return self.high[:0], self.total[:0]
        self.nk.reduce_reads_device(flags.data_ptr(), flags.data_ptr(), offsets.data_ptr(), n_reads, self.high.data_ptr(), self.total.data_ptr())
        return self.high[:n_reads], self.total[:n_reads]

    def decide(self, high, total, n_units, paired, keep):
        self.nk.decide_device(high.data_ptr(), total.data_ptr(), n_units, paired, keep.data_ptr())

    def synchronize(self):
        self.nk.synchronize()

    def reset(self):
        self.nk.reset()
        self.home_positions = self.home_total = 0

    def stats(self):
        return self.nk.stats()


class ShardedNormaliser:
    """One step = every rank's batch of reads, processed as one global batch (rank-major stream order)."""

    def __init__(self, engine, group=None, gather_to_root=True, stage_through_host=False, max_message_bytes=768 << 20, check_elems=None,
                 time_phases=False):
        self.eng = engine
        self.phases = Phases(time_phases, getattr(engine, "dev", None))
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gather_to_root = gather_to_root
        # rehearsal aid only: lets several ranks share one GPU under the gloo backend (RCCL refuses two ranks on
        # one device) by bouncing the exchanged buffers through host memory
        self.stage_through_host = stage_through_host
        self.max_message_bytes = max_message_bytes
        self._buffer = 0      # which of the engine's two sets of exchange buffers the next step uses
        self._ahead = None    # the exchange started for the step announced by `then`
        # Start-up check of the exchange with messages as large as this job can send (see check_exchange): elements per
        # peer, None = from the engine's sizes, 0 = no check
        if check_elems is None:
            check_elems = -(-int(getattr(engine, "max_positions", 0)) * 5 // (4 * self.world))
        if check_elems:
            self.check_exchange(check_elems, getattr(engine, "key_dtype", torch.int64), getattr(engine, "dev", "cpu"))

    def check_exchange(self, n, dtype, device):
        """One all-to-all with known content and n elements for every peer, through the same rounds as the real exchange.
        A stack that loses part of a large message (seen with one message of more than 1024 MiB) must not go unnoticed:
        if what arrives is not what was sent, max_message_bytes is halved (more rounds) and the check repeated; after
        three failures the job stops."""
        W, me = self.world, self.rank
        n = int(n)
        for attempt in range(4):
            src = torch.arange(W * n, dtype=torch.int64, device=device)
            src = (src + 7919 * me).to(dtype)
            out = torch.zeros_like(src)
            self._all_to_all(out, src, [n] * W, [n] * W, n)
            ok = True
            for p in range(W):   # rank p sent its segment number `me`
                want = (torch.arange(me * n, (me + 1) * n, dtype=torch.int64, device=device) + 7919 * p).to(dtype)
                ok = ok and bool(torch.equal(out[p * n:(p + 1) * n], want))
            del src, out
            cdev = "cpu" if self.stage_through_host or str(device) == "cpu" else device
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if int(flag.item()):
                return
            self.max_message_bytes //= 2
        raise RuntimeError("all-to-all of %d elements per peer does not deliver what was sent, even in rounds of %d bytes"
                           % (n, self.max_message_bytes * 2))

    def _all_to_all_start(self, out, inp, out_splits, in_splits, peak):
        """out/inp: per-peer segments back to back.  `peak` is the largest segment anywhere in the job (the same
        number on every rank); segments are exchanged in rounds of at most max_message_bytes each, because one
        message of more than 1024 MiB arrives cut in half through all_to_all_single on this stack (RCCL 2.26.6 of torch 2.10+rocm7.0; tools/rccl_msg_check.py: 1024 MiB whole, 1025 MiB half lost).
        The collectives are left in flight: returns a list of (handle to wait for or None, what to do after it or None)."""
        limit = max(1, self.max_message_bytes // inp.element_size())
        rounds = max(1, -(-peak // limit))
        if rounds == 1:
            return [(self._all_to_all_once(out, inp, out_splits, in_splits), None, (out, inp))]
        in_at = np.concatenate([[0], np.cumsum(in_splits)])
        out_at = np.concatenate([[0], np.cumsum(out_splits)])
        owed = []
        for r in range(rounds):
            lo = r * limit
            isz = [int(min(max(n - lo, 0), limit)) for n in in_splits]
            osz = [int(min(max(n - lo, 0), limit)) for n in out_splits]
            piece_in = torch.cat([inp[int(in_at[p]) + lo: int(in_at[p]) + lo + isz[p]] for p in range(self.world)])
            piece_out = torch.empty(sum(osz), dtype=out.dtype, device=out.device)
            work = self._all_to_all_once(piece_out, piece_in, osz, isz)

            def place(piece_out=piece_out, osz=osz, lo=lo):
                at = 0
                for p in range(self.world):
                    out[int(out_at[p]) + lo: int(out_at[p]) + lo + osz[p]].copy_(piece_out[at: at + osz[p]])
                    at += osz[p]
            owed.append((work, place, (piece_in, piece_out)))
        return owed

    @staticmethod
    def _all_to_all_finish(owed):
        for work, after, _alive in owed:
            if work is not None:
                work.wait()
            if after is not None:
                after()

    def _all_to_all(self, out, inp, out_splits, in_splits, peak):
        self._all_to_all_finish(self._all_to_all_start(out, inp, out_splits, in_splits, peak))

    def _all_to_all_once(self, out, inp, out_splits, in_splits):
        if not self.stage_through_host or not inp.is_cuda:
            return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)
        self.eng.synchronize()
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits, group=self.group)
        out.copy_(h_out)
        return None

    def _p2p_gather(self, packed, sizes):
        host = self.stage_through_host and packed.is_cuda
        src = packed.cpu() if host else packed.contiguous()
        if self.rank != 0:
            dist.send(src, dst=0, group=self.group)
            return None
        bufs = [torch.empty((2, s), dtype=src.dtype, device=src.device) for s in sizes]
        bufs[0].copy_(src)
        for q in [dist.irecv(bufs[r], src=r, group=self.group) for r in range(1, self.world)]:
            q.wait()
        return [b.to(packed.device) for b in bufs] if host else bufs

    def _exchange_counts(self, send_counts, like, failed=False, cap=None, n_reads=0):
        """What every peer sends here, the largest segment of the whole exchange (for the round count), every rank's number
        of reads (for the gather at the end of the step), and whether the step has to stop: a rank whose routing failed
        (`failed`: e.g. a read with a byte outside ACGTN) or that would receive more keys than it was sized for (`cap`) must
        not be left alone with its exception while its peers walk into the next collective.  All of it travels in one small
        all-gather (the whole table of counts: world x world numbers), one host round trip per step; whether a rank would
        receive too much can be worked out by everybody from that table."""
        W = self.world
        cdev = "cpu" if self.stage_through_host or not like.is_cuda else like.device
        mine = torch.tensor(list(send_counts) + [1 if failed else 0, int(n_reads), -1 if cap is None else int(cap)], dtype=torch.int64, device=cdev)
        table = torch.empty(W * (W + 3), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(table, mine, group=self.group)
        rows = table.view(W, W + 3).tolist()
        recv = [int(rows[p][self.rank]) for p in range(W)]
        peak = max(max(r[:W]) for r in rows)
        over = [rows[q][W + 2] >= 0 and sum(rows[p][q] for p in range(W)) > rows[q][W + 2] for q in range(W)]
        stop = any(r[W] for r in rows) or any(over)
        self._n_reads_all = [int(r[W + 1]) for r in rows]
        return recv, int(peak), bool(stop), bool(over[self.rank])

    def _agree(self, err, like):
        """Every rank calls this after a stage that can fail on one rank alone (err = its exception or None): all raise, or none."""
        cdev = "cpu" if self.stage_through_host or not like.is_cuda else like.device
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            if err is not None:
                raise err
            raise RuntimeError("another rank stopped this step (a read with a byte outside ACGTN, or a batch beyond the sizes it was set up for)")

    def _agree_and_share_saturated(self, err, like):
        """_agree after the counting stage, and with an engine that keeps a saturation bitmap: every rank's newly saturated
        k-mers reach every other rank (the sizes ride in the same small collective that carries the error flags; the lists
        follow in one all-gather, padded to the longest)."""
        eng, W = self.eng, self.world
        collect = getattr(eng, "sat_collect", None)
        new = None
        if collect is not None and err is None:
            try:
                new = collect()
            except Exception as e:
                err = e
        if collect is None or (new is None and err is None):
            return self._agree(err, like)
        cdev = "cpu" if self.stage_through_host or not like.is_cuda else like.device
        mine = torch.tensor([0 if err is None else 1, 0 if new is None else int(new.numel())], dtype=torch.int64, device=cdev)
        table = torch.empty(2 * W, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(table, mine, group=self.group)
        rows = table.view(W, 2).tolist()
        if any(r[0] for r in rows):
            if err is not None:
                raise err
            raise RuntimeError("another rank stopped this step (a read with a byte outside ACGTN, or a batch beyond the sizes it was set up for)")
        longest = max(r[1] for r in rows)
        self.phases.add_bytes("newly saturated k-mers announced", int(new.numel()) * new.element_size() * (W - 1))
        if W == 1 or longest == 0:
            return
        host = self.stage_through_host and new.is_cuda
        pad = torch.zeros(longest, dtype=new.dtype, device="cpu" if host else new.device)
        pad[: new.numel()].copy_(new)
        everyone = torch.empty(W * longest, dtype=new.dtype, device=pad.device)
        dist.all_gather_into_tensor(everyone, pad, group=self.group)
        if host:
            everyone = everyone.to(new.device)
        for p in range(W):
            if p != self.rank and rows[p][1]:
                eng.sat_apply(everyone[p * longest: p * longest + int(rows[p][1])])

    def _start_exchange(self, seq, offsets, n_reads, n_bases):
        """Everything of a step that does not need the table: k-mers, routing, the all-to-all of keys (left in flight)."""
        eng, W = self.eng, self.world
        self._buffer ^= 1
        if hasattr(eng, "use_buffers"):
            eng.use_buffers(self._buffer)
        x = {"key": (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases)}
        err = None
        try:
            with self.phases.dev("route"):
                if hasattr(eng, "kmerise_route"):
                    send_keys, x["send_src"], x["send_counts"], x["n_positions"] = eng.kmerise_route(seq, offsets, n_reads, n_bases, W)
                else:
                    keys = eng.kmerise(seq, offsets, n_reads, n_bases)
                    send_keys, x["send_src"], x["send_counts"] = eng.route(keys, W)
                    x["n_positions"] = keys.numel()
        except Exception as e:   # this rank cannot go on, but its peers have to hear of it before anyone raises
            err, send_keys, x["send_counts"] = e, seq[:0], [0] * W
        with self.phases.host("sizes (host round trip)"):
            x["recv_counts"], x["peak"], stop, too_many = self._exchange_counts(x["send_counts"], seq, err is not None, getattr(eng, "cap", None), n_reads)
        x["n_reads_all"] = self._n_reads_all
        if stop:
            if err is not None:
                raise err
            if too_many:
                raise RuntimeError("received %d keys, more than the %d this rank was sized for" % (sum(x["recv_counts"]), eng.cap))
            raise RuntimeError("another rank stopped this step (a read with a byte outside ACGTN, or more keys than it was sized for)")
        if W == 1 and hasattr(eng, "kmerise_route"):
            # nothing has to travel: the only segment is this rank's own, and it is counted where routing left it
            x["recv_keys"], x["owed"] = send_keys, []
        else:
            x["recv_keys"] = eng.recv_buffer(sum(x["recv_counts"]))
            x["owed"] = self._all_to_all_start(x["recv_keys"], send_keys, x["recv_counts"], x["send_counts"], x["peak"])
            away = sum(x["send_counts"]) - x["send_counts"][self.rank]
            self.phases.add_bytes("keys sent to other ranks", away * send_keys.element_size())
            self.phases.add_bytes("keys received from other ranks", (sum(x["recv_counts"]) - x["recv_counts"][self.rank]) * send_keys.element_size())
        x["alive"] = (send_keys, seq, offsets)
        return x

    def process_step(self, seq, offsets, n_reads, n_bases, then=None):
        """seq/offsets: this rank's reads (device tensors for the HIP engine).  Returns (high, total) of this
        rank's reads, and on rank 0 additionally the gathered (high_all, total_all) lists per rank.
        then = (seq, offsets, n_reads, n_bases) of the step that follows, if the caller has it already: its k-mers are
        routed and sent while this step's are counted.  All ranks must pass `then` in the same steps, and the next
        call must be for exactly that batch."""
        eng, W = self.eng, self.world
        key = (seq.data_ptr(), offsets.data_ptr(), n_reads, n_bases)
        x, self._ahead = self._ahead, None
        if x is not None and x["key"] != key:
            raise RuntimeError("process_step was announced another batch by the previous call's `then`")
        if x is None:
            x = self._start_exchange(seq, offsets, n_reads, n_bases)
        ph = self.phases
        with ph.dev("key exchange (not hidden)"):
            self._all_to_all_finish(x["owed"])
        if then is not None:
            self._ahead = self._start_exchange(*then)
        err = None
        try:
            with ph.dev("count"):
                recv_flags = eng.count(x["recv_keys"])
        except Exception as e:   # e.g. the table shard cannot grow any further
            err = e
        with ph.host("agree (host round trip)"):
            self._agree_and_share_saturated(err, seq)
        if W == 1 and hasattr(eng, "kmerise_route"):
            back = recv_flags
        else:
            back = eng.back_buffer(sum(x["send_counts"]))
            with ph.dev("flag exchange"):
                self._all_to_all(back, recv_flags, x["send_counts"], x["recv_counts"], x["peak"])
            ph.add_bytes("flags sent to other ranks", sum(x["recv_counts"]) - x["recv_counts"][self.rank])
        with ph.dev("flags home + per-read sums"):
            flags = eng.unroute(back, x["send_src"], x["n_positions"])
            high, total = eng.reduce(flags, offsets, n_reads)
        gathered = None
        if self.gather_to_root:
            with ph.dev("gather to rank 0"):
                gathered = self._p2p_gather(torch.stack([high, total]), x["n_reads_all"])
        ph.step_done()
        return high, total, gathered

    def phase_report(self):
        return self.phases.report()

    def synchronize(self):
        self.eng.synchronize()

    def reset(self):
        self.eng.reset()
        self.phases.reset()

    def stats(self):
        return self.eng.stats()
