/*
 * nkml_hip.h - C ABI of the MI355X (gfx950) k-mer coverage-normalisation hot path.
 *
 * The reference program (normalise_kmers_multi_large.c, "ref" below) has no FFI or plugin
 * interface: its hot path is the file-local call
 *     bool process_sequence_pair(fwd_seq, rev_seq, hash_table, &fwd_high, &fwd_total,
 *                                &rev_high, &rev_total, thread_id, K, depth_per_cpu)   ref:279-280, 1551-1566
 *     bool process_sequence_single(fwd_seq, hash_table, &high, &total, ...)              ref:1539-1550
 * made once per record from process_thread_chunk_paired (ref:1629) / _single (ref:1976), on a table that
 * lives for the whole run (ref:2279).  A per-record call is useless for a GPU, so this ABI is the same
 * contract in batch form: a context owns the k-mer table for the run; each call takes a batch of reads
 * in stream order and returns, per read, the two integers the reference's callers consume
 * (high-count k-mers, total k-mers).  The float32 keep/skip test (ref:1641-1646, 1988-1992) is exported
 * separately so the host record loop can stay where it is.
 *
 * Semantics (equal to the reference whenever its table is collision-free, i.e. capacity == 4^k, ref:676-684):
 *   count[x] starts at 0; k-mer occurrences are taken in stream order (read order, window left to right);
 *   each occurrence does count[x] += 1 and is "high" iff count[x] >= depth afterwards (ref:1486-1497);
 *   windows whose 2-bit code is 0 are neither counted nor totalled (ref:1483-1484);
 *   'N' is read as 'A' (ref:1406); reads shorter than k must not be submitted (the caller drops the
 *   record like ref:1630-1631); any byte outside ACGTN is an error (the reference exits, ref:1416-1420).
 *   Results do not depend on how the stream is cut into batches.
 *
 * Plain C types only.  All functions return NKML_OK (0) or a negative NKML_E_* code; none calls exit().
 * A context may be used from one host thread at a time.
 */
#ifndef NKML_HIP_H
#define NKML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NKML_ABI_VERSION 5

enum {
    NKML_OK = 0,
    NKML_E_ARG = -1,        /* bad argument (k outside 5..31 ref:724, depth < 2 ref:734, null pointer, batch too large) */
    NKML_E_HIP = -2,        /* a HIP runtime call failed; see nkml_last_error() */
    NKML_E_NOMEM = -3,      /* device or host allocation failed (reference: exit on calloc failure, ref:1071-1075) */
    NKML_E_BADBASE = -4,    /* a sequence byte outside ACGTN (reference: FATAL + exit, ref:1416-1420, 1445-1454) */
    NKML_E_STATE = -5,      /* call order violated (collect without submit, ...) */
    NKML_E_NODEVICE = -6    /* no usable gfx950 device */
};

typedef struct nkml_ctx nkml_ctx;

typedef struct nkml_config {
    int32_t ksize;            /* --ksize, 5..31 (ref:535, 724) */
    int32_t depth;            /* --depth; the threshold compared is depth / cpus with cpus == 1 (ref:674) */
    int32_t canonical;        /* --canonical (ref:1472) */
    int32_t device;           /* HIP device ordinal */
    int32_t memory_gb;        /* --memory_start: initial capacity of the hashed table used for k > 16, in GiB of
                                 16-byte slots like ref:416-422; 0 = the reference default of 1 GiB (ref:137).
                                 k <= 16 uses a direct-addressed counter array of 4^k entries (what ref:678-684
                                 clamps to) and ignores this field. */
    uint32_t max_batch_reads; /* largest n_reads of one batch */
    uint64_t max_batch_bases; /* largest total sequence bytes of one batch (< 2^32) */
    uint32_t chunk_bases;     /* window positions counted per internal pass; 0 = default (32 Mi) */
    uint32_t flags;           /* NKML_F_* */
} nkml_config;

#define NKML_F_PROFILE 1u      /* time the counting kernel with HIP events on the context's stream */
#define NKML_F_EXACT_COUNTS 2u /* keep counting k-mers after they reach depth (only nkml_dump_table can tell: --print,
                                  ref:354-385); costs one more atomic per saturated occurrence and, for k <= 16,
                                  a second 4^k-entry counter array */
#define NKML_F_COUNT_ATOMIC 4u /* count with atomics on the table even where the partitioned path would be chosen */
#define NKML_F_COUNT_PARTITIONED 8u /* count by radix-partitioning each batch's k-mers into buckets that are counted in LDS,
                                  wherever that path can run (depth <= 16383, no NKML_F_EXACT_COUNTS).  k <= 15: buckets of
                                  2^14 codes against a table of one byte per k-mer (two for depth > 127).  k > 16: buckets by
                                  the top 16 bits of a hash of the code against a table of 8-byte slots that is swept once
                                  per batch and doubles when a bucket fills up.  With neither flag the library chooses:
                                  partitioned at k = 15 and, for batches up to about 2 M pairs of 150 bases, at k = 14;
                                  partitioned at k > 16 while the table has at most 4 slots per position of a full batch,
                                  atomics from then on (the stored k-mers move over, between two batches, for good: the
                                  sweep of a large table costs more than the random accesses it replaces - nkml_stats
                                  .partitioned tells which path a context is on); atomics otherwise.  The environment
                                  variable NKML_PARTITIONED=0/1 overrides the choice like the flags.  Per-read results are
                                  identical either way; dumped counts of k-mers past the depth differ (see nkml_dump_table). */
#define NKML_F_VERIFY_RANKS 16u /* check every batch of the partitioned paths with kernels that do not rely on the order in
                                  which the LDS serves the returning adds of one instruction (the partitioned paths take their
                                  stream-order ranks from it; nkml_create measures it, the hardware documents no guarantee):
                                  every partition pass against a stable partition worked out from wavefront ballots, every
                                  bucket against a copy of the table taken before the batch (per k-mer: count moved by its
                                  occurrences, as many low flags as the depth leaves room for, none after a high one).  A
                                  batch that fails returns NKML_E_STATE and leaves the context unusable until nkml_reset.
                                  Several times slower, one table copy more in memory: for validation runs on new hardware,
                                  drivers or compilers.  The environment variable NKML_VERIFY_RANKS=1 sets it as well.
                                  No effect on the atomic path (its ranks come from a sort). */
#define NKML_F_NO_TABLE 32u     /* no table: the context only plays the sender of the bucket-ordered exchange (nkml_xb_front /
                                  nkml_xb_finish, nkml_sat_apply) - k <= 15 with NKML_F_COUNT_PARTITIONED; every call that
                                  would count returns NKML_E_STATE.  Saves the 4^k-entry table and its clearing. */

typedef struct nkml_stats {
    uint64_t unique_kmers;     /* hash_table->used (ref:166, 965): distinct non-zero codes stored, seeded ones included */
    uint64_t capacity;         /* table slots (4^k for the direct table) */
    uint64_t table_bytes;
    uint64_t kmers_counted;    /* occurrences that went through the table since create/reset (sum of all 'total') */
    uint64_t kmers_saturated;  /* of those, occurrences answered from the saturation bitmap without an atomic */
    uint64_t crossing_kmers;   /* k-mers whose count reached depth so far */
    uint64_t crossing_occurrences; /* occurrences that needed exact stream-order ranking */
    uint64_t resizes;          /* hashed table growth events (expand_local_hash_table, ref:1055-1108) */
    double count_kernel_ms;    /* NKML_F_PROFILE: summed duration of the counting kernel's launches (atomic path: the
                                  kernel that counts a chunk of positions; partitioned path: the bucket-counting kernel) */
    uint64_t count_kernel_launches;
    uint64_t count_kernel_keys;    /* window positions those launches covered */
    double count_stage_ms;     /* NKML_F_PROFILE, partitioned path: summed duration of the whole counting stage of each
                                  batch (histograms, partition passes, bucket counting, the flags' way back); else 0 */
    uint64_t partitioned;      /* 1 if this context counts by partitioning (NKML_F_COUNT_PARTITIONED), 0 if with atomics */
} nkml_stats;

/* Library/ABI probes (no GPU needed). */
int nkml_abi_version(void);
const char *nkml_strerror(int code);
/* Message of the last failure on this context (static buffer owned by the context). */
const char *nkml_last_error(const nkml_ctx *ctx);

/* Life cycle.  Replaces init_hash_table / destroy_hash_table (ref:881-906) and the per-run table of ref:2279.
 * Fails with NKML_E_NODEVICE when no gfx950 device is present: there is no CPU fallback.  The partitioned counting paths
 * take stream-order ranks from returning LDS adds, which the device has to serve in lane order within one instruction
 * (measured behaviour of gfx950, not a documented guarantee); nkml_create checks that once per device (under a
 * millisecond, every CU as busy as the counting kernels keep it).  On a device that fails the check a context counts on
 * the atomic path instead (one line on stderr; nkml_stats.partitioned says 0), and only a context that asks for the
 * partitioned path with NKML_F_COUNT_PARTITIONED is refused (NKML_E_NODEVICE, message in nkml_last_error()).
 * A context whose batch failed half-way (out of memory while the hashed table grew inside a batch, a HIP error between
 * two kernels of a batch) holds a partly counted batch: every later submit / process / seed call on it returns
 * NKML_E_STATE until nkml_reset. */
int nkml_create(const nkml_config *cfg, nkml_ctx **out);
void nkml_destroy(nkml_ctx *ctx);
/* Forget every count (a fresh run on the same context). */
int nkml_reset(nkml_ctx *ctx);
/* Use a caller-owned HIP stream (hipStream_t passed as void*) instead of the context's own. */
int nkml_set_stream(nkml_ctx *ctx, void *hip_stream);
int nkml_synchronize(nkml_ctx *ctx);
int nkml_get_stats(nkml_ctx *ctx, nkml_stats *out);

/* Host-buffer path (what the command-line host uses): reads concatenated in `seq`, read i occupies
 * [offsets[i], offsets[i+1]); mates of a pair are consecutive reads (forward first, ref:1559-1563).
 * submit stages the batch with hipMemcpyAsync and enqueues the kernels.  On the partitioned path it returns without
 * waiting for the device, so the caller can work while the batch is counted, and collect reports a bad base
 * (NKML_E_BADBASE) if the batch held one.  On the atomic path submit waits once inside (for the number of occurrences to
 * rank) and returns NKML_E_BADBASE itself; nothing is pending then and collect must not be called for that batch.
 * collect waits and returns the per-read counters in submission order.
 * Memory of the hashed table (k > 16): it starts at memory_gb GiB and doubles whenever a counting pass (chunk_bases
 * positions) could take it beyond 80 % full; during a doubling the old and the new table coexist, and the next table is
 * allocated ahead of time once the current one is 40 % full, so the path holds up to 3x the current table's bytes.  Batch form of process_sequence_pair/_single (ref:1551, 1539). */
int nkml_submit(nkml_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads);
int nkml_collect(nkml_ctx *ctx, int32_t *high, int32_t *total);
/* The context's own pinned staging area (max_batch_bases bytes, max_batch_reads + 1 offsets): a caller that
 * builds its batch there and passes these pointers to nkml_submit / nkml_seed saves one copy of the batch.
 * No counterpart in the reference (its records are counted where read_line left them, ref:1605-1629). */
int nkml_host_staging(nkml_ctx *ctx, uint8_t **seq, uint64_t **offsets);
/* Waits until the last submitted / seeded batch has left the staging area for the device (not for its kernels):
 * from then on the area may be filled with the next batch while the GPU is still counting.  No counterpart in the
 * reference. */
int nkml_wait_staged(nkml_ctx *ctx);

/* Device-buffer path: the same batch already resident in HBM; d_high/d_total are device arrays of n_reads.
 * Runs on the context's stream; the call waits for the device once per batch (for the number of occurrences to
 * rank on the atomic path, for the bad-base tally on the partitioned path) and returns NKML_E_BADBASE if the batch
 * held a byte outside ACGTN. */
int nkml_process_device(nkml_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads,
                        uint64_t n_bases, int32_t *d_high, int32_t *d_total);

/* Seeding insert, batch form of sequence_to_hash_zero (ref:1501-1537): keys enter the table with count 0.
 * Only nkml_stats.unique_kmers can tell the difference.  Host buffers, synchronous. */
int nkml_seed(nkml_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads);

/* Keep/skip test, float32 with strict '<' (ref:1641-1646 paired; ref:1988-1992 single). Host side. */
int nkml_keep_pair(int32_t fwd_high, int32_t fwd_total, int32_t rev_high, int32_t rev_total, float coverage);
int nkml_keep_single(int32_t high, int32_t total, float coverage);
/* The same test for a whole batch on the device: reads 2i and 2i+1 are the mates of pair i (paired != 0),
 * or every read on its own (paired == 0).  d_keep gets one byte per pair / read. */
int nkml_decide_device(nkml_ctx *ctx, const int32_t *d_high, const int32_t *d_total, uint32_t n_units,
                       int paired, float coverage, uint8_t *d_keep);

/* Stages of nkml_process_device, exported for the k-mer-sharded multi-GPU path, where the keys are exchanged
 * between devices (all-to-all) after the first stage and the flags travel back before the last one.
 *   kmerise : sequence bytes -> one key per base position (0 where no countable window starts); key width is
 *             4 bytes for k <= 16, else 8 (nkml_key_bytes).  Restates ref:1464-1484 + encode/canonical.
 *   count   : keys in stream order -> one flag byte per key (NKML_FLAG_*), updating the table.  Restates
 *             store_kmer + the high test (ref:929-1053, 1486-1497) with exact stream-order ranks.
 *   reduce  : per read, high = number of NKML_FLAG_HIGH flags, total = number of non-zero keys. */
/* d_keys / d_flags of the three stages hold one entry per base position rounded up to a multiple of 16. */
int nkml_key_bytes(const nkml_ctx *ctx);
int nkml_kmerise_device(nkml_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads,
                        uint64_t n_bases, void *d_keys);
int nkml_count_keys_device(nkml_ctx *ctx, const void *d_keys, uint64_t n_keys, uint8_t *d_flags);
int nkml_reduce_reads_device(nkml_ctx *ctx, const void *d_keys, const uint8_t *d_flags, const uint64_t *d_offsets,
                             uint32_t n_reads, int32_t *d_high, int32_t *d_total);
/*   seed_keys : the seeding insert of nkml_seed (sequence_to_hash_zero, ref:1501-1537) for keys that are already made: every
 *               non-zero key enters the table with count 0.  n_keys must be a multiple of 16.  Synchronous. */
int nkml_seed_keys_device(nkml_ctx *ctx, const void *d_keys, uint64_t n_keys);

/* Sharding of the table over several devices (one context per device, each holding the counts of the keys it
 * owns).  The reference never merges its per-thread tables (ref:2421); this exchange step is new.
 *   owner_of      : the shard that owns a key (host function; the device kernels use the same rule).
 *   route_keys    : stable partition of a key array by owner: d_send_keys/d_send_src get the non-zero keys grouped
 *                   by destination shard, each group in stream order, with the base position each key came from;
 *                   send_counts[s] (host) = keys for shard s.  Synchronous (the counts drive the all-to-all).
 *   unroute_flags : flags returned by the owners, in send order, back to their base positions (others get 0). */
int nkml_owner_of(uint64_t key, uint32_t n_shards);
int nkml_route_keys_device(nkml_ctx *ctx, const void *d_keys, uint64_t n_keys, uint32_t n_shards, void *d_send_keys,
                           uint32_t *d_send_src, uint64_t *send_counts);
/*   route_keys_static : nkml_route_keys_device for callers that cannot allocate (tight serving loops, harnesses that
 *                   forbid hipMalloc).  It allocates nothing, but like nkml_route_keys_device it waits for the stream once,
 *                   to hand the counts back in host memory - so it cannot run inside a HIP graph capture.  No counterpart
 *                   in the reference.  Same inputs,
 *                   outputs and results as nkml_route_keys_device, but the per-tile shard counts, their scan and the scan's
 *                   block totals live in a fixed array in device memory (static, about 4 MiB per GPU, held until the
 *                   process ends) instead of buffers allocated on demand, and the call uses no allocating routine.
 *                   Limit: ceil(n_keys / 1024) * n_shards <= NKML_ROUTE_STATIC_MAX_ENTRIES, else NKML_E_ARG.  All calls on
 *                   one GPU share that array: calls that overlap in time on one GPU (two contexts, two streams) race. */
#define NKML_ROUTE_STATIC_MAX_ENTRIES (1u << 19)
int nkml_route_keys_static(nkml_ctx *ctx, const void *d_keys, uint64_t n_keys, uint32_t n_shards, void *d_send_keys,
                           uint32_t *d_send_src, uint64_t *send_counts);
/*   route_reads   : the same partition taken straight from a batch of reads (k-mers recomputed from the packed
 *                   sequence, no key array in between); positions in d_send_src are base positions of the batch. */
int nkml_route_reads_device(nkml_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases,
                            uint32_t n_shards, void *d_send_keys, uint32_t *d_send_src, uint64_t *send_counts);
int nkml_unroute_flags_device(nkml_ctx *ctx, const uint8_t *d_recv_flags, const uint32_t *d_send_src, uint64_t n_sent,
                              uint8_t *d_flags, uint64_t n_keys);
/*   Ranked routing (no counterpart in the reference; exchange mode's cheaper form of route_reads / unroute_flags, at most
 *   32 shards): instead of every sent key's base position (4 bytes, in send order) the routing leaves 16 bits per base
 *   position in d_meta (room for n_bases rounded up to a multiple of 1024) - the shard and the key's rank among the keys
 *   of its routing tile for that shard, or "no k-mer", or "answered here" - and the tiles' start offsets in every shard's
 *   segment in d_tile_off (ceil(n / 1024) * n_shards + 1 entries).  nkml_unroute_ranked turns the flags that came back
 *   (d_back_flags: in send order, like d_recv_flags above) into the flag of every base position with them.  The caller
 *   keeps d_meta and d_tile_off from the routing of a batch to its way home (batches may overlap: buffers per batch).
 *   n_home: positions answered by the saturation bitmap (0 without one). */
int nkml_route_reads_ranked(nkml_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases,
                            uint32_t n_shards, void *d_send_keys, uint16_t *d_meta, uint32_t *d_tile_off, uint64_t *send_counts,
                            uint64_t *n_home);
int nkml_unroute_ranked(nkml_ctx *ctx, const uint8_t *d_back_flags, const uint16_t *d_meta, const uint32_t *d_tile_off,
                        uint64_t n_positions, uint8_t *d_flags);
/*   Saturation bitmap (no counterpart in the reference; k <= 15 on the partitioned direct table): a k-mer whose count has
 *   reached depth - 1 at its owner makes every later occurrence "high" whatever is added to it, so such occurrences need
 *   not travel.  nkml_sat_enable gives the context one bit per k-mer code (4^k / 8 bytes; cleared by nkml_reset), from
 *   then on nkml_route_reads_ranked answers k-mers whose bit is set where they are read (flag "high", not sent, not
 *   counted anywhere: nkml_stats adds them to kmers_counted and kmers_saturated of the context that read them; dumped
 *   counts of such k-mers stop growing).  nkml_sat_collect scans this context's table for k-mers that have reached
 *   depth - 1 and have no bit yet, sets their bits and writes their codes to d_codes (device memory of the caller; what
 *   exceeds max_codes is dropped from the list, n_codes = what was written; waits for the stream); nkml_sat_apply sets
 *   the bits of a list that another rank collected.  A bitmap may lag behind the tables by any number of batches: a k-mer without its bit just
 *   travels and is counted by its owner as before, results are the same. */
int nkml_sat_enable(nkml_ctx *ctx);
int nkml_sat_collect(nkml_ctx *ctx, uint32_t *d_codes, uint64_t max_codes, uint64_t *n_codes);
int nkml_sat_apply(nkml_ctx *ctx, const uint32_t *d_codes, uint64_t n_codes);
/*   Bucket-ordered exchange (no counterpart in the reference; k <= 15 on the partitioned direct table; what bucketed.py drives
 *   with one process per GPU).  Every rank partitions its own batch by table bucket (2^14 k-mer codes each) with the passes of
 *   the single-GPU path; k-mers travel to their bucket's owner as 16-bit codes within the bucket, bucket after bucket, each
 *   bucket in stream order; the owner strings the ranks' runs of a bucket together in rank order - the bucket's k-mers of
 *   the step's global stream (rank 0's reads, then rank 1's, ...) - and counts them with the bucket kernels alone.
 *   xb_enable  : once per context.  world ranks, this one is `rank`; recv_capacity = most keys that may arrive in one step.
 *                Buckets are dealt by a fixed pseudo-random permutation (the same on every rank): with B = max(4^k / 2^14, 1)
 *                buckets, rank r owns those at places [B * r / world, B * (r + 1) / world) of the deal.
 *   xb_front   : a batch of this rank's reads -> d_send_keys (room for n_bases + 16 keys): the travelling k-mers, owner
 *                after owner and within an owner bucket after bucket (by place); d_send_counts (B entries): travelling
 *                k-mers per bucket, by place - the owner of places [a, b) gets entries [a, b); send_totals[r] = keys for
 *                rank r.  With a saturation bitmap (nkml_sat_enable) k-mers whose bit is set are answered here (n_home of
 *                them) and do not travel.  Waits for the stream.  Nothing else may use the context's partition buffers
 *                until xb_finish (one batch under way per context).
 *   xb_count   : the owner's side.  d_recv_keys: what the ranks sent, rank after rank (n_recv keys in all);
 *                d_recv_counts: [world][buckets of this rank], the counts that came with them; d_recv_flags (n_recv
 *                bytes, same order as d_recv_keys): NKML_FLAG_HIGH or 0 per key, to be sent back the way the keys came.
 *   xb_finish  : d_back_flags: the owners' answers in the order of d_send_keys -> per-read counters of the batch that
 *                xb_front took (d_offsets, n_reads: the same as there). */
int nkml_xb_enable(nkml_ctx *ctx, uint32_t world, uint32_t rank, uint64_t recv_capacity);
int nkml_xb_front(nkml_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, uint16_t *d_send_keys,
                  uint32_t *d_send_counts, uint64_t *send_totals, uint64_t *n_home);
int nkml_xb_count(nkml_ctx *ctx, const uint16_t *d_recv_keys, const uint32_t *d_recv_counts, uint64_t n_recv, uint8_t *d_recv_flags);
int nkml_xb_finish(nkml_ctx *ctx, const uint8_t *d_back_flags, const uint64_t *d_offsets, uint32_t n_reads, int32_t *d_high, int32_t *d_total);


/* The other way to spread one table over a few devices: every device sees every read of a step and counts only
 * the k-mers of its own table shard, so nothing per k-mer crosses a link - only the 2-bit packed reads (one all-gather)
 * and the per-read partial counters (one sum).  Worth it while the step's reads, times the number of devices, still
 * fit one context (positions stay below 2^32) - two to four devices.  New like the exchange above (ref:2421).
 *   set_shard      : from now on the context counts only k-mers with shard_of(key, n_shards) == shard (partitioned
 *                    counting path only, else NKML_E_STATE); (0, 1) switches the filter off.  The per-read high and
 *                    total then are this shard's share; summed over all shards they are the single-table numbers.
 *   shard_of       : the shard of a k-mer code below 2^32 (host function, same rule as the kernels).
 *   pack           : sequence bytes -> 2-bit packed words, 16 bases per word, first base in the top bits (the order
 *                    of encode_kmer_plain, ref:1118-1126), the tail of the last word filled with A.  Asynchronous; a
 *                    byte outside ACGTN is reported by the context's next call that waits.
 *   process_packed : nkml_process_device for a batch that is already packed.  d_offsets may leave gaps between
 *                    reads' words as long as it is non-decreasing: list a gap as a read of its own (if it holds only
 *                    A it has no countable k-mer).  Waits like nkml_process_device. */
int nkml_set_shard(nkml_ctx *ctx, uint32_t shard, uint32_t n_shards);
int nkml_shard_of(uint64_t key, uint32_t n_shards);
int nkml_pack_device(nkml_ctx *ctx, const uint8_t *d_seq, uint64_t n_bases, uint32_t *d_words);
int nkml_process_packed_device(nkml_ctx *ctx, const uint32_t *d_words, const uint64_t *d_offsets, uint32_t n_reads,
                               uint64_t n_bases, int32_t *d_high, int32_t *d_total);

/* One table over several GPUs, driven from one process (what the command-line host uses when it is given several
 * devices).  The reference has no counterpart: its threads keep private tables and never merge them (ref:2421).  The
 * table is sharded by k-mer (nkml_owner_of); every batch is cut into one contiguous slice of reads per device, the
 * k-mers travel to their owners (RCCL send/receive over xGMI when every shard has a GPU of its own, device-to-device
 * copies when shards share a GPU - NKML_TRANSPORT=copy|rccl overrides), are counted there, and the flags travel back;
 * the per-read counters are those of one table reading the batch in stream order, whatever the number of devices.
 *   create   : devices[i] = HIP ordinal of shard i (an ordinal may repeat: shards then share that GPU, which is how the
 *              path is exercised on a one-GPU machine); cfg->device is ignored, cfg->max_batch_* are for the whole batch.
 *              On failure *out may still be set, for nkml_multi_last_error; destroy it.
 *   the rest : as the single-device functions of the same name (host buffers, submit / collect in pairs). */
typedef struct nkml_multi nkml_multi;
int nkml_multi_create(const nkml_config *cfg, const int32_t *devices, uint32_t n_devices, nkml_multi **out);
void nkml_multi_destroy(nkml_multi *m);
const char *nkml_multi_last_error(const nkml_multi *m);
uint32_t nkml_multi_device_count(const nkml_multi *m);
int nkml_multi_reset(nkml_multi *m);
int nkml_multi_host_staging(nkml_multi *m, uint8_t **seq, uint64_t **offsets);
int nkml_multi_wait_staged(nkml_multi *m);
int nkml_multi_submit(nkml_multi *m, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads);
int nkml_multi_collect(nkml_multi *m, int32_t *high, int32_t *total);
int nkml_multi_seed(nkml_multi *m, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads);
int nkml_multi_get_stats(nkml_multi *m, nkml_stats *out);
int nkml_multi_dump_table(nkml_multi *m, uint64_t *codes, uint32_t *counts, uint64_t max_entries, uint64_t *n_entries);

#define NKML_FLAG_NONE 0u /* no k-mer at this position */
#define NKML_FLAG_HIGH 1u /* occurrence whose stream-order count is >= depth */
#define NKML_FLAG_LOW 2u  /* counted, below depth */

/* Table dump for --print (print_kmer_table, ref:354-385): up to max_entries (code, count) pairs of every
 * stored k-mer, unsorted like the reference; *n_entries gets the number available. Host arrays, synchronous.
 * Without NKML_F_EXACT_COUNTS a k-mer's count stops growing once it has reached depth-1 at the start of a
 * counting pass (from then on every occurrence is high whatever the count); on the partitioned paths it keeps
 * growing up to 127 (one-byte entries), 32767 (two-byte entries, hashed slots) and stops there. */
int nkml_dump_table(nkml_ctx *ctx, uint64_t *codes, uint32_t *counts, uint64_t max_entries, uint64_t *n_entries);

#ifdef __cplusplus
}
#endif
#endif /* NKML_HIP_H */
