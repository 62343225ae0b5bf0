// One k-mer table spread over several GPUs of one node, driven from one process: the multi-device face of the C ABI
// (nkml_multi_* in nkml_hip.h), what the command-line host uses when it is given more than one device.
//
// The reference has nothing like it: its threads keep private tables and never merge them
// (normalise_kmers_multi_large.c:2421), which changes results with the thread count.  Here the table is sharded by k-mer
// (owner = nkml_owner_of(key, n): a multiplicative hash of the whole key, so that low-complexity k-mers do not pile up on
// one device) and results are those of one table reading the batch in stream order:
//
//   1. the batch's reads are cut into n contiguous slices, slice i goes to device i (stream order = slice 0, slice 1, ...);
//   2. every device makes the k-mers of its slice and partitions them stably by owner (nkml_route_reads_device);
//   3. exchange: owner j receives, from every source i in order, the keys i holds for it - the global stream order
//      restricted to j's keys, which is what exact stream-order ranks need;
//   4. owner j counts them against its shard (nkml_count_keys_device: the single-device kernels) -> one flag per key;
//   5. the flags travel back to their sources, go back to base positions (nkml_unroute_flags_device) and are reduced
//      per read (nkml_reduce_reads_device); each device's (high, total) go straight to the caller's host arrays.
//
// Transport of steps 3 and 5: RCCL send/receive pairs inside one group call per direction (the all-to-all(v) over xGMI)
// when every shard has a GPU of its own; device-to-device copies otherwise - several shards on one GPU is how the
// multi-device path is exercised on a one-GPU machine (NKML_TRANSPORT=copy|rccl overrides).  RCCL is opened at run
// time (dlopen), so the library has no link-time dependency on it.
//
// One host thread per device runs the per-device stages (some of them wait for their device); the calling thread
// joins them between stages.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "nkml_hip.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;

    bool open()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
#define NKML_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym))
        NKML_SYM(CommInitAll, "ncclCommInitAll");
        NKML_SYM(CommDestroy, "ncclCommDestroy");
        NKML_SYM(GroupStart, "ncclGroupStart");
        NKML_SYM(GroupEnd, "ncclGroupEnd");
        NKML_SYM(Send, "ncclSend");
        NKML_SYM(Recv, "ncclRecv");
        NKML_SYM(GetErrorString, "ncclGetErrorString");
#undef NKML_SYM
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};

struct Shard {
    int device = 0;
    nkml_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_staged = nullptr, ev_routed = nullptr, ev_counted = nullptr;
    // this device's slice of the batch
    uint8_t *d_seq = nullptr;
    uint64_t *d_offsets = nullptr;
    uint32_t n_reads = 0, first_read = 0;
    uint64_t n_bases = 0, first_base = 0, n_positions = 0;
    // exchange buffers (key_bytes per key)
    void *d_send_keys = nullptr, *d_recv_keys = nullptr;
    uint32_t *d_send_src = nullptr;
    uint8_t *d_recv_flags = nullptr, *d_back_flags = nullptr, *d_flags = nullptr;
    int32_t *d_high = nullptr, *d_total = nullptr;
    std::vector<uint64_t> send_counts, recv_counts; // keys for / from every shard
    uint64_t n_sent = 0, n_recv = 0;
    int rc = NKML_OK;
};

} // namespace

struct nkml_multi {
    nkml_config cfg;
    uint32_t n = 0;
    std::vector<Shard> sh;
    uint64_t cap_bases = 0;   // per shard: bases of a slice, and keys received
    uint32_t cap_reads = 0;
    int key_bytes = 4;
    bool use_rccl = false;
    uint64_t rccl_piece = 512ull << 20; // most bytes of one send / receive (NKML_RCCL_PIECE_BYTES: test aid)
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    // pinned staging of the whole batch
    uint8_t *h_seq = nullptr;
    uint64_t *h_offsets = nullptr, *h_slice_offsets = nullptr;
    int32_t *h_high = nullptr, *h_total = nullptr;
    uint32_t pending_reads = 0;
    bool pending = false, staged_pending = false;
    bool broken = false; // a batch failed after some shard had counted its part: unusable until nkml_multi_reset
    char err[640];
};

namespace {

int mfail(nkml_multi *m, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(m->err, sizeof m->err, fmt, ap);
    va_end(ap);
    return code;
}

#define MHIP(m, call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return mfail(m, e_ == hipErrorOutOfMemory ? NKML_E_NOMEM : NKML_E_HIP, "%s failed: %s (%s:%d)", #call, \
                         hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

// runs fn(i) for every shard on a host thread of its own and waits for all of them; the first failure is returned
template <typename F>
int for_each_shard(nkml_multi *m, F fn)
{
    if (m->n == 1) {
        m->sh[0].rc = fn(0u);
    } else {
        std::vector<std::thread> th;
        th.reserve(m->n);
        for (uint32_t i = 0; i < m->n; i++) th.emplace_back([m, i, &fn] { m->sh[i].rc = fn(i); });
        for (auto &t : th) t.join();
    }
    for (uint32_t i = 0; i < m->n; i++) {
        if (m->sh[i].rc != NKML_OK) {
            const int rc = m->sh[i].rc;
            snprintf(m->err, sizeof m->err, "shard %u (device %d): %s: %s", i, m->sh[i].device, nkml_strerror(rc),
                     m->sh[i].ctx ? nkml_last_error(m->sh[i].ctx) : "");
            return rc;
        }
    }
    return NKML_OK;
}

// Cuts the batch into n contiguous slices of about equal base counts (whole reads; pairs are kept together when
// `pair_aligned`, so that a slice's first read is a forward mate - not needed for the counts, only tidy).
void cut_slices(nkml_multi *m, const uint64_t *offsets, uint32_t n_reads)
{
    const uint64_t total = offsets[n_reads];
    uint32_t at = 0;
    for (uint32_t i = 0; i < m->n; i++) {
        Shard &s = m->sh[i];
        s.first_read = at;
        s.first_base = offsets[at];
        uint32_t end = n_reads;
        if (i + 1 < m->n) {
            const uint64_t want = total * (i + 1) / m->n;
            // first read boundary at or beyond `want` (binary search over the offsets)
            uint32_t lo = at, hi = n_reads;
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (offsets[mid] < want) lo = mid + 1; else hi = mid;
            }
            end = lo;
        }
        s.n_reads = end - at;
        s.n_bases = offsets[end] - offsets[at];
        s.n_positions = (s.n_bases + 15) / 16 * 16;
        at = end;
    }
}

// stage 1 on shard i: its slice to the device, k-mers made and routed by owner (waits for the device: the per-owner
// counts steer the exchange)
int stage_route(nkml_multi *m, uint32_t i)
{
    Shard &s = m->sh[i];
    s.send_counts.assign(m->n, 0);
    s.n_sent = 0;
    if (hipSetDevice(s.device) != hipSuccess) return NKML_E_HIP;
    if (s.n_reads == 0) return NKML_OK;
    if (s.n_bases > m->cap_bases || s.n_reads > m->cap_reads) return NKML_E_ARG;
    uint64_t *ho = m->h_slice_offsets + s.first_read + i; // n_reads + 1 entries per slice: slices do not overlap
    for (uint32_t r = 0; r <= s.n_reads; r++) ho[r] = m->h_offsets[s.first_read + r] - s.first_base;
    if (hipMemcpyAsync(s.d_seq, m->h_seq + s.first_base, s.n_bases, hipMemcpyHostToDevice, s.stream) != hipSuccess) return NKML_E_HIP;
    if (hipMemcpyAsync(s.d_offsets, ho, ((uint64_t)s.n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream) != hipSuccess) return NKML_E_HIP;
    if (hipEventRecord(s.ev_staged, s.stream) != hipSuccess) return NKML_E_HIP;
    const int rc = nkml_route_reads_device(s.ctx, s.d_seq, s.d_offsets, s.n_reads, s.n_bases, m->n, s.d_send_keys, s.d_send_src, s.send_counts.data());
    if (rc != NKML_OK) return rc;
    for (uint32_t j = 0; j < m->n; j++) s.n_sent += s.send_counts[j];
    return NKML_OK;
}

// Exchange of per-peer segments: shard i's segment for j (send[i] at its running offset, send_counts[i][j] elements of
// `elem` bytes) lands in shard j's buffer after the segments of the sources before i.  forward: keys to their owners;
// !forward: the flags on their way back (send/recv roles swapped, counts likewise).
int exchange(nkml_multi *m, bool forward, size_t elem)
{
    const uint32_t n = m->n;
    auto seg = [&](uint32_t from, uint32_t to) { return forward ? m->sh[from].send_counts[to] : m->sh[from].recv_counts[to]; };
    auto src_buf = [&](uint32_t i) { return static_cast<char *>(forward ? m->sh[i].d_send_keys : (void *)m->sh[i].d_recv_flags); };
    auto dst_buf = [&](uint32_t i) { return static_cast<char *>(forward ? m->sh[i].d_recv_keys : (void *)m->sh[i].d_back_flags); };
    if (m->use_rccl) {
        // the sending side's data must be complete before its stream sends: keys are (stage_route waited), flags are
        // ordered by the stream itself (the counting kernels run on the same stream)
        // A segment travels in pieces of at most rccl_piece bytes, the same cuts on both sides: on this stack (RCCL 2.26.6)
        // a grouped send/receive of more than 1 GiB delivers its first half only (seen through torch's all_to_all_single
        // with a world of one: 1024 MiB whole, 1025 MiB and up cut at exactly half their bytes, the rest zero; all-gather,
        // all-reduce and broadcast of the same sizes arrive whole).
        ncclResult_t r = m->rccl.GroupStart();
        for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
            uint64_t so = 0, ro = 0;
            for (uint32_t j = 0; j < n && r == ncclSuccess; j++) {
                const uint64_t ns = seg(i, j) * elem, nr = seg(j, i) * elem;
                for (uint64_t at = 0; at < ns && r == ncclSuccess; at += m->rccl_piece)
                    r = m->rccl.Send(src_buf(i) + so + at, ns - at < m->rccl_piece ? ns - at : m->rccl_piece, ncclUint8, (int)j, m->comms[i], m->sh[i].stream);
                for (uint64_t at = 0; at < nr && r == ncclSuccess; at += m->rccl_piece)
                    r = m->rccl.Recv(dst_buf(i) + ro + at, nr - at < m->rccl_piece ? nr - at : m->rccl_piece, ncclUint8, (int)j, m->comms[i], m->sh[i].stream);
                so += ns;
                ro += nr;
            }
        }
        const ncclResult_t e = m->rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return mfail(m, NKML_E_HIP, "RCCL exchange failed: %s", m->rccl.GetErrorString(r));
        return NKML_OK;
    }
    // copies on the receiving shard's stream, after the sending shard's data is complete
    for (uint32_t j = 0; j < n; j++) {
        MHIP(m, hipSetDevice(m->sh[j].device));
        uint64_t ro = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t cnt = seg(i, j);
            if (cnt) {
                uint64_t so = 0;
                for (uint32_t t = 0; t < j; t++) so += seg(i, t);
                if (i != j) MHIP(m, hipStreamWaitEvent(m->sh[j].stream, forward ? m->sh[i].ev_routed : m->sh[i].ev_counted, 0));
                MHIP(m, hipMemcpyAsync(dst_buf(j) + ro * elem, src_buf(i) + so * elem, cnt * elem, hipMemcpyDeviceToDevice, m->sh[j].stream));
            }
            ro += cnt;
        }
    }
    return NKML_OK;
}

// between the stages: who receives how much (host side; the counts came back from stage_route)
int plan_receives(nkml_multi *m)
{
    for (uint32_t j = 0; j < m->n; j++) {
        Shard &d = m->sh[j];
        d.recv_counts.assign(m->n, 0);
        d.n_recv = 0;
        for (uint32_t i = 0; i < m->n; i++) {
            d.recv_counts[i] = m->sh[i].send_counts[j];
            d.n_recv += d.recv_counts[i];
        }
        if (d.n_recv + 16 > m->cap_bases)
            return mfail(m, NKML_E_ARG, "shard %u would receive %llu k-mers of this batch, it is sized for %llu (skewed input: use smaller batches)",
                         j, (unsigned long long)d.n_recv, (unsigned long long)m->cap_bases);
    }
    return NKML_OK;
}

int record_all(nkml_multi *m, hipEvent_t Shard::*ev)
{
    for (uint32_t i = 0; i < m->n; i++) {
        MHIP(m, hipSetDevice(m->sh[i].device));
        MHIP(m, hipEventRecord(m->sh[i].*ev, m->sh[i].stream));
    }
    return NKML_OK;
}

// stage 2 on shard j: count what it received (or insert it with count 0), flags in receive order
int stage_count(nkml_multi *m, uint32_t j, bool seeding)
{
    Shard &s = m->sh[j];
    if (hipSetDevice(s.device) != hipSuccess) return NKML_E_HIP;
    if (s.n_recv == 0) return NKML_OK;
    const uint64_t n16 = (s.n_recv + 15) / 16 * 16;
    if (n16 > s.n_recv &&
        hipMemsetAsync(static_cast<char *>(s.d_recv_keys) + s.n_recv * m->key_bytes, 0, (n16 - s.n_recv) * m->key_bytes, s.stream) != hipSuccess)
        return NKML_E_HIP;
    return seeding ? nkml_seed_keys_device(s.ctx, s.d_recv_keys, n16) : nkml_count_keys_device(s.ctx, s.d_recv_keys, n16, s.d_recv_flags);
}

// stage 3 on shard i: flags back at their base positions, per-read counters, and home to the host
int stage_reduce(nkml_multi *m, uint32_t i)
{
    Shard &s = m->sh[i];
    if (hipSetDevice(s.device) != hipSuccess) return NKML_E_HIP;
    if (s.n_reads == 0) return NKML_OK;
    int rc = nkml_unroute_flags_device(s.ctx, s.d_back_flags, s.d_send_src, s.n_sent, s.d_flags, s.n_positions);
    if (rc != NKML_OK) return rc;
    rc = nkml_reduce_reads_device(s.ctx, s.d_flags, s.d_flags, s.d_offsets, s.n_reads, s.d_high, s.d_total);
    if (rc != NKML_OK) return rc;
    if (hipMemcpyAsync(m->h_high + s.first_read, s.d_high, (uint64_t)s.n_reads * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream) != hipSuccess) return NKML_E_HIP;
    if (hipMemcpyAsync(m->h_total + s.first_read, s.d_total, (uint64_t)s.n_reads * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream) != hipSuccess) return NKML_E_HIP;
    return NKML_OK;
}

int stage_host_batch(nkml_multi *m, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads)
{
    if (!seq || !offsets) return NKML_E_ARG;
    if (offsets[0] != 0) return mfail(m, NKML_E_ARG, "offsets[0] must be 0");
    const uint64_t n_bases = offsets[n_reads];
    if (n_reads > m->cfg.max_batch_reads) return mfail(m, NKML_E_ARG, "batch of %u reads exceeds max_batch_reads %u", n_reads, m->cfg.max_batch_reads);
    if (n_bases > m->cfg.max_batch_bases) return mfail(m, NKML_E_ARG, "batch of %llu bases exceeds max_batch_bases %llu", (unsigned long long)n_bases, (unsigned long long)m->cfg.max_batch_bases);
    for (uint32_t i = 0; i < n_reads; i++)
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] < (uint64_t)m->cfg.ksize)
            return mfail(m, NKML_E_ARG, "read %u is shorter than k=%d (the caller drops such records)", i, m->cfg.ksize);
    if (seq != m->h_seq) memcpy(m->h_seq, seq, n_bases);
    if (offsets != m->h_offsets) memcpy(m->h_offsets, offsets, ((uint64_t)n_reads + 1) * sizeof(uint64_t));
    cut_slices(m, m->h_offsets, n_reads);
    return NKML_OK;
}

// the whole batch up to the point where every device has its work enqueued
int run_batch(nkml_multi *m, bool seeding)
{
    int rc;
    if ((rc = for_each_shard(m, [m](uint32_t i) { return stage_route(m, i); })) != NKML_OK) return rc;
    m->staged_pending = true;
    if ((rc = plan_receives(m)) != NKML_OK) return rc;
    if ((rc = record_all(m, &Shard::ev_routed)) != NKML_OK) return rc;
    if ((rc = exchange(m, true, (size_t)m->key_bytes)) != NKML_OK) return rc;
    if ((rc = for_each_shard(m, [m, seeding](uint32_t j) { return stage_count(m, j, seeding); })) != NKML_OK) return rc;
    if (seeding) return NKML_OK;
    if ((rc = record_all(m, &Shard::ev_counted)) != NKML_OK) return rc;
    if ((rc = exchange(m, false, 1)) != NKML_OK) return rc;
    return for_each_shard(m, [m](uint32_t i) { return stage_reduce(m, i); });
}

int sync_all(nkml_multi *m)
{
    for (uint32_t i = 0; i < m->n; i++) {
        MHIP(m, hipSetDevice(m->sh[i].device));
        MHIP(m, hipStreamSynchronize(m->sh[i].stream));
    }
    return NKML_OK;
}

} // namespace

extern "C" {

const char *nkml_multi_last_error(const nkml_multi *m) { return m ? m->err : "null context"; }

void nkml_multi_destroy(nkml_multi *m)
{
    if (!m) return;
    for (auto &s : m->sh) {
        hipSetDevice(s.device);
        if (s.stream) hipStreamSynchronize(s.stream);
    }
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i]) m->rccl.CommDestroy(m->comms[i]);
    for (auto &s : m->sh) {
        hipSetDevice(s.device);
        if (s.ctx) nkml_destroy(s.ctx);
        hipFree(s.d_seq); hipFree(s.d_offsets); hipFree(s.d_send_keys); hipFree(s.d_recv_keys); hipFree(s.d_send_src);
        hipFree(s.d_recv_flags); hipFree(s.d_back_flags); hipFree(s.d_flags); hipFree(s.d_high); hipFree(s.d_total);
        if (s.ev_staged) hipEventDestroy(s.ev_staged);
        if (s.ev_routed) hipEventDestroy(s.ev_routed);
        if (s.ev_counted) hipEventDestroy(s.ev_counted);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (m->h_seq) hipHostFree(m->h_seq);
    if (m->h_offsets) hipHostFree(m->h_offsets);
    if (m->h_slice_offsets) hipHostFree(m->h_slice_offsets);
    if (m->h_high) hipHostFree(m->h_high);
    if (m->h_total) hipHostFree(m->h_total);
    delete m;
}

int nkml_multi_create(const nkml_config *cfg, const int32_t *devices, uint32_t n_devices, nkml_multi **out)
{
    if (!cfg || !devices || !out || n_devices == 0 || n_devices > 64) return NKML_E_ARG;
    *out = nullptr;
    nkml_multi *m = new (std::nothrow) nkml_multi();
    if (!m) return NKML_E_NOMEM;
    m->err[0] = 0;
    m->cfg = *cfg;
    m->n = n_devices;
    m->sh.resize(n_devices);
    *out = m; // handed out even on failure, so that the caller can read the message (and must destroy it)
    // a slice is 1/n of the batch cut at a read boundary; what a shard receives is 1/n of the k-mers on average:
    // sized with half as much again for uneven owners
    const uint64_t per = (cfg->max_batch_bases + n_devices - 1) / n_devices;
    m->cap_bases = (per + per / 2 + 8192 + 15) / 16 * 16;
    if (n_devices == 1) m->cap_bases = (cfg->max_batch_bases + 64 + 15) / 16 * 16;
    m->cap_reads = cfg->max_batch_reads;
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; i++)
        for (uint32_t j = 0; j < i; j++)
            if (devices[i] == devices[j]) distinct = false;
    const char *tr = getenv("NKML_TRANSPORT");
    m->use_rccl = tr ? strcmp(tr, "rccl") == 0 : (distinct && n_devices > 1);
    if (const char *e = getenv("NKML_RCCL_PIECE_BYTES")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 16 && v <= (1ull << 30)) m->rccl_piece = v & ~15ull;
    }
    if (m->use_rccl && !distinct) return mfail(m, NKML_E_ARG, "RCCL needs a GPU of its own for every shard (NKML_TRANSPORT=copy for shards that share one)");
    for (uint32_t i = 0; i < n_devices; i++) {
        Shard &s = m->sh[i];
        s.device = devices[i];
        nkml_config c = *cfg;
        c.device = devices[i];
        c.max_batch_bases = m->cap_bases;
        c.max_batch_reads = m->cap_reads;
        c.flags &= ~NKML_F_PROFILE;
        const int rc = nkml_create(&c, &s.ctx);
        if (rc != NKML_OK) return mfail(m, rc, "shard %u (device %d): %s: %s", i, devices[i], nkml_strerror(rc), s.ctx ? nkml_last_error(s.ctx) : "");
        MHIP(m, hipSetDevice(s.device));
        MHIP(m, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        if (nkml_set_stream(s.ctx, s.stream) != NKML_OK) return mfail(m, NKML_E_HIP, "shard %u: %s", i, nkml_last_error(s.ctx));
        MHIP(m, hipEventCreateWithFlags(&s.ev_staged, hipEventDisableTiming));
        MHIP(m, hipEventCreateWithFlags(&s.ev_routed, hipEventDisableTiming));
        MHIP(m, hipEventCreateWithFlags(&s.ev_counted, hipEventDisableTiming));
        m->key_bytes = nkml_key_bytes(s.ctx);
        const uint64_t cap = m->cap_bases;
        MHIP(m, hipMalloc(&s.d_seq, cap + 64));
        MHIP(m, hipMalloc(&s.d_offsets, ((uint64_t)m->cap_reads + 1) * sizeof(uint64_t)));
        MHIP(m, hipMalloc(&s.d_send_keys, cap * m->key_bytes));
        MHIP(m, hipMalloc(&s.d_recv_keys, cap * m->key_bytes));
        MHIP(m, hipMalloc(&s.d_send_src, cap * sizeof(uint32_t)));
        MHIP(m, hipMalloc(&s.d_recv_flags, cap));
        MHIP(m, hipMalloc(&s.d_back_flags, cap));
        MHIP(m, hipMalloc(&s.d_flags, cap));
        MHIP(m, hipMalloc(&s.d_high, (uint64_t)m->cap_reads * sizeof(int32_t)));
        MHIP(m, hipMalloc(&s.d_total, (uint64_t)m->cap_reads * sizeof(int32_t)));
    }
    if (distinct && n_devices > 1) { // copies between devices want peer access; without it the runtime stages them
        for (uint32_t i = 0; i < n_devices; i++) {
            MHIP(m, hipSetDevice(devices[i]));
            for (uint32_t j = 0; j < n_devices; j++) {
                int can = 0;
                if (i != j && hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) MHIP(m, e);
                    (void)hipGetLastError();
                }
            }
        }
    }
    if (m->use_rccl) {
        if (!m->rccl.open()) return mfail(m, NKML_E_HIP, "cannot open librccl.so (NKML_TRANSPORT=copy uses device-to-device copies instead)");
        m->comms.assign(n_devices, nullptr);
        std::vector<int> devs(devices, devices + n_devices);
        const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), (int)n_devices, devs.data());
        if (r != ncclSuccess) return mfail(m, NKML_E_HIP, "ncclCommInitAll: %s", m->rccl.GetErrorString(r));
    }
    MHIP(m, hipHostMalloc(&m->h_seq, cfg->max_batch_bases + 64, hipHostMallocPortable));
    MHIP(m, hipHostMalloc(&m->h_offsets, ((uint64_t)cfg->max_batch_reads + 1) * sizeof(uint64_t), hipHostMallocPortable));
    MHIP(m, hipHostMalloc(&m->h_slice_offsets, ((uint64_t)cfg->max_batch_reads + 1 + n_devices) * sizeof(uint64_t), hipHostMallocPortable));
    MHIP(m, hipHostMalloc(&m->h_high, (uint64_t)cfg->max_batch_reads * sizeof(int32_t), hipHostMallocPortable));
    MHIP(m, hipHostMalloc(&m->h_total, (uint64_t)cfg->max_batch_reads * sizeof(int32_t), hipHostMallocPortable));
    return NKML_OK;
}

uint32_t nkml_multi_device_count(const nkml_multi *m) { return m ? m->n : 0; }

int nkml_multi_reset(nkml_multi *m)
{
    if (!m) return NKML_E_ARG;
    m->pending = m->staged_pending = false;
    const int rc = for_each_shard(m, [m](uint32_t i) { return nkml_reset(m->sh[i].ctx); });
    if (rc == NKML_OK) m->broken = false;
    return rc;
}

int nkml_multi_host_staging(nkml_multi *m, uint8_t **seq, uint64_t **offsets)
{
    if (!m || !seq || !offsets) return NKML_E_ARG;
    *seq = m->h_seq;
    *offsets = m->h_offsets;
    return NKML_OK;
}

int nkml_multi_wait_staged(nkml_multi *m)
{
    if (!m) return NKML_E_ARG;
    if (m->staged_pending) {
        for (uint32_t i = 0; i < m->n; i++)
            if (m->sh[i].n_reads) {
                MHIP(m, hipSetDevice(m->sh[i].device));
                MHIP(m, hipEventSynchronize(m->sh[i].ev_staged));
            }
        m->staged_pending = false;
    }
    return NKML_OK;
}

int nkml_multi_submit(nkml_multi *m, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads)
{
    if (!m) return NKML_E_ARG;
    if (m->broken) return mfail(m, NKML_E_STATE, "an earlier batch failed half-way and left some shards' tables updated: nkml_multi_reset before counting again");
    if (m->pending) return mfail(m, NKML_E_STATE, "previous batch not collected");
    m->pending_reads = n_reads;
    if (n_reads == 0) { m->pending = true; return NKML_OK; }
    int rc = stage_host_batch(m, seq, offsets, n_reads);
    if (rc != NKML_OK) return rc;
    // once the shards have started counting, a failure on one of them leaves the others' tables ahead of it
    if ((rc = run_batch(m, false)) != NKML_OK) { m->broken = true; return rc; }
    m->pending = true;
    return NKML_OK;
}

int nkml_multi_collect(nkml_multi *m, int32_t *high, int32_t *total)
{
    if (!m || !high || !total) return NKML_E_ARG;
    if (!m->pending) return mfail(m, NKML_E_STATE, "nothing submitted");
    m->pending = false;
    if (m->pending_reads == 0) return NKML_OK;
    int rc = sync_all(m);
    if (rc != NKML_OK) { m->broken = true; return rc; }
    // the contexts' own end-of-batch checks (e.g. the bad-base tally of a device whose route stage did not wait for it)
    if ((rc = for_each_shard(m, [m](uint32_t i) { return nkml_synchronize(m->sh[i].ctx); })) != NKML_OK) { m->broken = true; return rc; }
    memcpy(high, m->h_high, (uint64_t)m->pending_reads * sizeof(int32_t));
    memcpy(total, m->h_total, (uint64_t)m->pending_reads * sizeof(int32_t));
    return NKML_OK;
}

int nkml_multi_seed(nkml_multi *m, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads)
{
    if (!m) return NKML_E_ARG;
    if (m->broken) return mfail(m, NKML_E_STATE, "an earlier batch failed half-way: nkml_multi_reset before seeding again");
    if (m->pending) return mfail(m, NKML_E_STATE, "previous batch not collected");
    if (n_reads == 0) return NKML_OK;
    int rc = stage_host_batch(m, seq, offsets, n_reads);
    if (rc != NKML_OK) return rc;
    if ((rc = run_batch(m, true)) != NKML_OK) { m->broken = true; return rc; }
    return sync_all(m);
}

int nkml_multi_get_stats(nkml_multi *m, nkml_stats *out)
{
    if (!m || !out) return NKML_E_ARG;
    memset(out, 0, sizeof *out);
    for (uint32_t i = 0; i < m->n; i++) {
        nkml_stats st;
        const int rc = nkml_get_stats(m->sh[i].ctx, &st);
        if (rc != NKML_OK) return mfail(m, rc, "shard %u: %s", i, nkml_last_error(m->sh[i].ctx));
        out->unique_kmers += st.unique_kmers; // the shards hold disjoint sets of k-mers
        out->capacity += st.capacity;
        out->table_bytes += st.table_bytes;
        out->kmers_counted += st.kmers_counted;
        out->kmers_saturated += st.kmers_saturated;
        out->crossing_kmers += st.crossing_kmers;
        out->crossing_occurrences += st.crossing_occurrences;
        out->resizes += st.resizes;
        out->partitioned = st.partitioned;
    }
    return NKML_OK;
}

int nkml_multi_dump_table(nkml_multi *m, uint64_t *codes, uint32_t *counts, uint64_t max_entries, uint64_t *n_entries)
{
    if (!m || !n_entries || (max_entries && (!codes || !counts))) return NKML_E_ARG;
    uint64_t have = 0, total = 0;
    for (uint32_t i = 0; i < m->n; i++) {
        uint64_t n = 0;
        const uint64_t room = max_entries > have ? max_entries - have : 0;
        const int rc = nkml_dump_table(m->sh[i].ctx, room ? codes + have : nullptr, room ? counts + have : nullptr, room, &n);
        if (rc != NKML_OK) return mfail(m, rc, "shard %u: %s", i, nkml_last_error(m->sh[i].ctx));
        total += n;
        have += n < room ? n : room;
    }
    *n_entries = total;
    return NKML_OK;
}

} // extern "C"
