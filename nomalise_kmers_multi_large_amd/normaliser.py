"""Host-side mirror of the reference's per-read interface, in batch form, over the C ABI.

The reference calls process_sequence_pair(fwd, rev, table, &hiF, &totF, &hiR, &totR, ...)
(normalise_kmers_multi_large.c:1551-1566) once per record and then applies the float32 coverage test
(:1641-1646).  KmerNormaliser keeps the same names, argument meaning and error behaviour for a batch of
records: too-short mates make the record "not processed" (the reference's `false`, :1630-1631), a byte
outside ACGTN raises (the reference exits, :1416-1420), counts persist across calls like the per-run table
(:2279)."""
import ctypes

import numpy as np

from . import binding as B


class KmerNormaliser:
    def __init__(self, ksize=15, depth=100, coverage=0.9, canonical=False, memory_gb=0, device=0,
                 max_batch_reads=1 << 20, max_batch_bases=1 << 28, chunk_bases=0, profile=False, flags=0):
        self.lib = B.load_library()
        self.ksize, self.depth, self.canonical = int(ksize), int(depth), bool(canonical)
        self.coverage = np.float32(coverage)
        cfg = B.Config(self.ksize, self.depth, int(self.canonical), int(device), int(memory_gb), int(max_batch_reads),
                       int(max_batch_bases), int(chunk_bases), (B.NKML_F_PROFILE if profile else 0) | int(flags))
        self.max_batch_reads, self.max_batch_bases = int(max_batch_reads), int(max_batch_bases)
        h = ctypes.c_void_p()
        rc = self.lib.nkml_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h if h.value else None
        if rc != B.NKML_OK:
            msg = self._err(rc)
            self.close()
            raise B.NkmlError(rc, msg)

    # -- plumbing ---------------------------------------------------------------------------------------
    def _err(self, rc):
        msg = self.lib.nkml_strerror(rc).decode()
        if self.h:
            detail = self.lib.nkml_last_error(self.h).decode()
            if detail:
                msg += ": " + detail
        return msg

    def _chk(self, rc):
        if rc != B.NKML_OK:
            raise B.NkmlError(rc, self._err(rc))

    def close(self):
        if getattr(self, "h", None):
            self.lib.nkml_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        self._chk(self.lib.nkml_reset(self.h))

    def synchronize(self):
        self._chk(self.lib.nkml_synchronize(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.nkml_set_stream(self.h, ctypes.c_void_p(stream_ptr)))

    def stats(self):
        st = B.Stats()
        self._chk(self.lib.nkml_get_stats(self.h, ctypes.byref(st)))
        return st.as_dict()

    @property
    def unique_kmers(self):
        return self.stats()["unique_kmers"]

    @property
    def key_bytes(self):
        return self.lib.nkml_key_bytes(self.h)

    # -- host-buffer path -------------------------------------------------------------------------------
    @staticmethod
    def _concat(reads):
        lens = np.fromiter((len(r) for r in reads), dtype=np.uint64, count=len(reads))
        offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        seq = np.frombuffer(b"".join(reads), dtype=np.uint8)
        return seq, offsets

    def process_batch(self, seq, offsets):
        """seq: uint8[n_bases] (ASCII, reads concatenated); offsets: uint64[n_reads+1].
        Returns (high int32[n_reads], total int32[n_reads]); every read must be >= ksize long."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        high = np.zeros(n, dtype=np.int32)
        total = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.nkml_submit(self.h, seq.ctypes.data, offsets.ctypes.data, n))
        self._chk(self.lib.nkml_collect(self.h, high.ctypes.data, total.ctypes.data))
        return high, total

    def seed_batch(self, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self.lib.nkml_seed(self.h, seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1))

    def process_sequence_single(self, seqs):
        """Batch form of process_sequence_single (normalise_kmers_multi_large.c:1539-1550).
        seqs: list of bytes.  Returns (ok bool[n], high int32[n], total int32[n]); ok[i] False = shorter than k
        (not counted, like the reference's early `return false`)."""
        ok = np.fromiter((len(s) >= self.ksize for s in seqs), dtype=bool, count=len(seqs))
        high = np.zeros(len(seqs), dtype=np.int32)
        total = np.zeros(len(seqs), dtype=np.int32)
        good = [s for s, o in zip(seqs, ok) if o]
        if good:
            h, t = self._process_list(good)
            high[ok], total[ok] = h, t
        return ok, high, total

    def process_sequence_pair(self, fwd_seqs, rev_seqs):
        """Batch form of process_sequence_pair (normalise_kmers_multi_large.c:1551-1566): record i is
        (fwd_seqs[i], rev_seqs[i]); both mates are validated before anything is counted, forward is counted
        before reverse.  Returns (ok, fwd_high, fwd_total, rev_high, rev_total)."""
        assert len(fwd_seqs) == len(rev_seqs)
        n = len(fwd_seqs)
        ok = np.fromiter((len(f) >= self.ksize and len(r) >= self.ksize for f, r in zip(fwd_seqs, rev_seqs)),
                         dtype=bool, count=n)
        out = [np.zeros(n, dtype=np.int32) for _ in range(4)]
        inter = []
        for f, r, o in zip(fwd_seqs, rev_seqs, ok):
            if o:
                inter.append(f)
                inter.append(r)
        if inter:
            h, t = self._process_list(inter)
            out[0][ok], out[1][ok], out[2][ok], out[3][ok] = h[0::2], t[0::2], h[1::2], t[1::2]
        return (ok,) + tuple(out)

    def _process_list(self, reads):
        """Counts a list of reads in order, cutting it into batches that fit the context."""
        highs, totals = [], []
        i = 0
        while i < len(reads):
            j, bases = i, 0
            while j < len(reads) and j - i < self.max_batch_reads and bases + len(reads[j]) <= self.max_batch_bases:
                bases += len(reads[j])
                j += 1
            if j == i:
                raise B.NkmlError(B.NKML_E_ARG, "read longer than max_batch_bases")
            seq, offsets = self._concat(reads[i:j])
            h, t = self.process_batch(seq, offsets)
            highs.append(h)
            totals.append(t)
            i = j
        return np.concatenate(highs), np.concatenate(totals)

    # -- decisions --------------------------------------------------------------------------------------
    def keep_pair(self, fwd_high, fwd_total, rev_high, rev_total):
        return bool(self.lib.nkml_keep_pair(int(fwd_high), int(fwd_total), int(rev_high), int(rev_total), self.coverage))

    def keep_single(self, high, total):
        return bool(self.lib.nkml_keep_single(int(high), int(total), self.coverage))

    # -- device-buffer path (pointers are plain integers, e.g. torch.Tensor.data_ptr()) -----------------
    def process_device(self, d_seq, d_offsets, n_reads, n_bases, d_high, d_total):
        self._chk(self.lib.nkml_process_device(self.h, d_seq, d_offsets, n_reads, n_bases, d_high, d_total))

    def set_shard(self, shard, n_shards):
        """Count only the k-mers of table shard `shard` of `n_shards` from now on (replicated-reads multi-GPU mode)."""
        self._chk(self.lib.nkml_set_shard(self.h, shard, n_shards))

    def pack_device(self, d_seq, n_bases, d_words):
        self._chk(self.lib.nkml_pack_device(self.h, d_seq, n_bases, d_words))

    def process_packed_device(self, d_words, d_offsets, n_reads, n_bases, d_high, d_total):
        self._chk(self.lib.nkml_process_packed_device(self.h, d_words, d_offsets, n_reads, n_bases, d_high, d_total))

    def decide_device(self, d_high, d_total, n_units, paired, d_keep):
        self._chk(self.lib.nkml_decide_device(self.h, d_high, d_total, n_units, int(paired), self.coverage, d_keep))

    def kmerise_device(self, d_seq, d_offsets, n_reads, n_bases, d_keys):
        self._chk(self.lib.nkml_kmerise_device(self.h, d_seq, d_offsets, n_reads, n_bases, d_keys))

    def count_keys_device(self, d_keys, n_keys, d_flags):
        self._chk(self.lib.nkml_count_keys_device(self.h, d_keys, n_keys, d_flags))

    def reduce_reads_device(self, d_keys, d_flags, d_offsets, n_reads, d_high, d_total):
        self._chk(self.lib.nkml_reduce_reads_device(self.h, d_keys, d_flags, d_offsets, n_reads, d_high, d_total))

    def route_keys_device(self, d_keys, n_keys, n_shards, d_send_keys, d_send_src):
        counts = (ctypes.c_uint64 * n_shards)()
        self._chk(self.lib.nkml_route_keys_device(self.h, d_keys, n_keys, n_shards, d_send_keys, d_send_src, counts))
        return [int(x) for x in counts]

    def route_keys_static(self, d_keys, n_keys, n_shards, d_send_keys, d_send_src):
        """nkml_route_keys_device without any allocation (fixed scratch in device memory; see include/nkml_hip.h)"""
        counts = (ctypes.c_uint64 * n_shards)()
        self._chk(self.lib.nkml_route_keys_static(self.h, d_keys, n_keys, n_shards, d_send_keys, d_send_src, counts))
        return [int(x) for x in counts]

    def route_reads_device(self, d_seq, d_offsets, n_reads, n_bases, n_shards, d_send_keys, d_send_src):
        counts = (ctypes.c_uint64 * n_shards)()
        self._chk(self.lib.nkml_route_reads_device(self.h, d_seq, d_offsets, n_reads, n_bases, n_shards, d_send_keys, d_send_src, counts))
        return [int(x) for x in counts]

    def unroute_flags_device(self, d_recv_flags, d_send_src, n_sent, d_flags, n_keys):
        self._chk(self.lib.nkml_unroute_flags_device(self.h, d_recv_flags, d_send_src, n_sent, d_flags, n_keys))

    def route_reads_ranked(self, d_seq, d_offsets, n_reads, n_bases, n_shards, d_send_keys, d_meta, d_tile_off):
        """nkml_route_reads_ranked: (keys sent to every shard, positions the saturation bitmap answered here)"""
        counts = (ctypes.c_uint64 * n_shards)()
        home = ctypes.c_uint64(0)
        self._chk(self.lib.nkml_route_reads_ranked(self.h, d_seq, d_offsets, n_reads, n_bases, n_shards, d_send_keys, d_meta, d_tile_off, counts,
                                                   ctypes.byref(home)))
        return [int(x) for x in counts], int(home.value)

    def unroute_ranked(self, d_back_flags, d_meta, d_tile_off, n_positions, d_flags):
        self._chk(self.lib.nkml_unroute_ranked(self.h, d_back_flags, d_meta, d_tile_off, n_positions, d_flags))

    def sat_enable(self):
        self._chk(self.lib.nkml_sat_enable(self.h))

    def sat_collect(self, d_codes, max_codes):
        n = ctypes.c_uint64(0)
        self._chk(self.lib.nkml_sat_collect(self.h, d_codes, max_codes, ctypes.byref(n)))
        return int(n.value)

    def sat_apply(self, d_codes, n_codes):
        self._chk(self.lib.nkml_sat_apply(self.h, d_codes, n_codes))

    def xb_enable(self, world, rank, recv_capacity):
        self._chk(self.lib.nkml_xb_enable(self.h, world, rank, recv_capacity))

    def xb_front(self, d_seq, d_offsets, n_reads, n_bases, d_send_keys, d_send_counts, world):
        """nkml_xb_front: (keys for every owner, positions answered here)"""
        totals = (ctypes.c_uint64 * world)()
        home = ctypes.c_uint64(0)
        self._chk(self.lib.nkml_xb_front(self.h, d_seq, d_offsets, n_reads, n_bases, d_send_keys, d_send_counts, totals, ctypes.byref(home)))
        return [int(x) for x in totals], int(home.value)

    def xb_count(self, d_recv_keys, d_recv_counts, n_recv, d_recv_flags):
        self._chk(self.lib.nkml_xb_count(self.h, d_recv_keys, d_recv_counts, n_recv, d_recv_flags))

    def xb_finish(self, d_back_flags, d_offsets, n_reads, d_high, d_total):
        self._chk(self.lib.nkml_xb_finish(self.h, d_back_flags, d_offsets, n_reads, d_high, d_total))

    def dump_table(self, max_entries=1 << 24):
        codes = np.zeros(max_entries, dtype=np.uint64)
        counts = np.zeros(max_entries, dtype=np.uint32)
        n = ctypes.c_uint64(0)
        self._chk(self.lib.nkml_dump_table(self.h, codes.ctypes.data, counts.ctypes.data, max_entries, ctypes.byref(n)))
        m = min(n.value, max_entries)
        return codes[:m], counts[:m], n.value


class MultiDeviceNormaliser:
    """ctypes face of the nkml_multi_* functions: one k-mer table sharded over several GPUs, driven from this process
    (what the command-line host uses with NKML_DEVICES).  `devices` lists one HIP ordinal per table shard; an ordinal may
    repeat (shards then share that GPU).  Host buffers in, per-read counters out, as KmerNormaliser.process_batch."""

    def __init__(self, devices, ksize=15, depth=100, coverage=0.9, canonical=False, memory_gb=0,
                 max_batch_reads=1 << 20, max_batch_bases=1 << 28, chunk_bases=0, flags=0):
        self.lib = B.load_library()
        self.ksize, self.depth, self.canonical = int(ksize), int(depth), bool(canonical)
        self.coverage = np.float32(coverage)
        cfg = B.Config(self.ksize, self.depth, int(self.canonical), 0, int(memory_gb), int(max_batch_reads),
                       int(max_batch_bases), int(chunk_bases), int(flags))
        devs = (ctypes.c_int32 * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.lib.nkml_multi_create(ctypes.byref(cfg), devs, len(devices), ctypes.byref(h))
        self.h = h if h.value else None
        if rc != B.NKML_OK:
            msg = self._err(rc)
            self.close()
            raise B.NkmlError(rc, msg)

    def _err(self, rc):
        msg = self.lib.nkml_strerror(rc).decode()
        if self.h:
            msg += ": " + self.lib.nkml_multi_last_error(self.h).decode()
        return msg

    def _chk(self, rc):
        if rc != B.NKML_OK:
            raise B.NkmlError(rc, self._err(rc))

    def close(self):
        if getattr(self, "h", None):
            self.lib.nkml_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        self._chk(self.lib.nkml_multi_reset(self.h))

    def stats(self):
        st = B.Stats()
        self._chk(self.lib.nkml_multi_get_stats(self.h, ctypes.byref(st)))
        return st.as_dict()

    @property
    def unique_kmers(self):
        return self.stats()["unique_kmers"]

    def process_batch(self, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        high = np.zeros(n, dtype=np.int32)
        total = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.nkml_multi_submit(self.h, seq.ctypes.data, offsets.ctypes.data, n))
        self._chk(self.lib.nkml_multi_collect(self.h, high.ctypes.data, total.ctypes.data))
        return high, total

    def seed_batch(self, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self.lib.nkml_multi_seed(self.h, seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1))

    def dump_table(self, max_entries=1 << 24):
        codes = np.zeros(max_entries, dtype=np.uint64)
        counts = np.zeros(max_entries, dtype=np.uint32)
        n = ctypes.c_uint64(0)
        self._chk(self.lib.nkml_multi_dump_table(self.h, codes.ctypes.data, counts.ctypes.data, max_entries, ctypes.byref(n)))
        m = min(n.value, max_entries)
        return codes[:m], counts[:m], n.value
