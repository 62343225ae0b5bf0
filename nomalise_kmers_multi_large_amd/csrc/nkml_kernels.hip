// HIP kernels (gfx950 / MI355X) of the k-mer coverage-normalisation hot path and the C ABI around them.
//
// Data layout in HBM for one batch of n_bases sequence bytes (reads concatenated, stream order):
//   seq       u8  [n_bases]            ASCII as read from FASTQ/FASTA
//   packed    u32 [ceil(n_bases/16)+4] 2-bit codes, 16 bases per word, first base in the top bits
//   startbits u32 [same]               bit j of word w set  <=>  base 16w+j is the first base of a read
//                                      (plus one sentinel bit at n_bases)
//   keys      u32|u64 [16*words]       per base position: code of the k-mer starting there, 0 = none
//   flags     u8  [16*words]           per base position: NKML_FLAG_*
// The table lives for the whole run:
//   k <= 16 : counts u32[4^k] (direct-addressed, what the reference degenerates to when capacity == 4^k,
//             normalise_kmers_multi_large.c:676-684), saturation bitmap of 4^k bits
//   k >  16 : 16-byte slots {key u64, count u32, state u32}, open addressing with linear probing
//             (same slot size as kmer_t, normalise_kmers_multi_large.c:156-161)
//
// Order: the reference is sequential, and an occurrence is "high" iff its 1-based stream-order rank among
// the occurrences of its k-mer is >= depth (normalise_kmers_multi_large.c:1486-1497).  Atomic adds give
// arbitrary ranks inside one parallel pass, so the pass is split:
//   (1) k-mers whose count was already >= depth-1 when the pass began ("saturated", one bit per k-mer, frozen
//       during the pass) are high whatever the order, and need no update at all;
//   (2) every other occurrence does one atomic add; a k-mer can only be mis-ranked if its count reaches depth
//       inside this pass ("crossing"); the occurrence that sees the count become depth lists the k-mer and sets
//       its bit in a small hashed filter (2 MiB, stays in L2; false positives only cost a little extra work);
//   (3) occurrences of crossing k-mers (a small set: each k-mer crosses once per run) are collected, sorted by
//       (k-mer, position) and given their true ranks: with m occurrences in the pass and final count F, the
//       i-th in stream order (0-based) has rank F - m + i + 1.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <algorithm>
#include <new>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "nkml_hip.h"

namespace {

constexpr uint32_t FLAG_NONE = NKML_FLAG_NONE, FLAG_HIGH = NKML_FLAG_HIGH, FLAG_LOW = NKML_FLAG_LOW;
constexpr uint32_t SEEDED_BIT = 0x80000000u; // direct table: k-mer present through seeding (count part may be 0)
constexpr uint32_t COUNT_MASK = 0x7fffffffu;
constexpr uint32_t ST_SAT = 0x80000000u; // hashed table slot state: top bit = saturated, low 31 bits = occurrences
                                          // counted after saturation (NKML_F_EXACT_COUNTS only)
constexpr int FILTER_BITS = 24;  // crossing filter: 2^24 bits = 2 MiB
constexpr uint32_t FILTER_WORDS = 1u << (FILTER_BITS - 5);

__device__ __forceinline__ uint32_t filter_index(uint32_t id) { return (id * 0x9E3779B1u) >> (32 - FILTER_BITS); }
__device__ __forceinline__ uint32_t filter_index(unsigned long long id)
{
    return (uint32_t)((id * 0x9E3779B97F4A7C15ull) >> (64 - FILTER_BITS));
}
__device__ __forceinline__ void filter_set(uint32_t *__restrict__ filter, uint32_t h) { atomicOr(&filter[h >> 5], 1u << (h & 31)); }
__device__ __forceinline__ bool filter_test(const uint32_t *__restrict__ filter, uint32_t h) { return (filter[h >> 5] >> (h & 31)) & 1u; }

struct alignas(256) Counters {
    unsigned long long unique;    // distinct k-mers stored (hash_table->used)
    unsigned long long counted;   // occurrences with a non-zero key
    unsigned long long saturated; // of those, answered by the saturation bit
    unsigned int n_sat_new;       // S list length (count reached depth-1 in this pass)
    unsigned int n_cross;         // X list length (count reached depth in this pass)
    unsigned int n_occ;           // E list length (occurrences of crossing k-mers, whole batch)
    unsigned long long total_cross; // crossing k-mers since reset
    unsigned int bad_bases;       // bytes outside ACGTN seen by the packer
    unsigned int probe_overflow;  // hashed table: a probe sequence wrapped (table full) - must stay 0
    unsigned int hp_left;         // hashed table, partitioned: buckets of the batch that the fast bucket kernel left to the general one
    unsigned int route_home;      // ranked routing: positions of the last batch whose k-mer the saturation bitmap answered (not sent)
    unsigned int sat_new;         // nkml_sat_collect: k-mers found saturated by the last scan
    unsigned int verify_bad;      // NKML_F_VERIFY_RANKS: elements whose place or flag differs from what the checking kernels work out
};

struct Slot {
    unsigned long long key; // 0 = empty
    unsigned int count;
    unsigned int state;
};
static_assert(sizeof(Slot) == 16, "slot layout");

// Hand-over of LDS data between the lanes of ONE wavefront.  A wavefront's LDS instructions execute in program order, so
// no barrier instruction is needed; what has to be pinned is the compiler's order: the stores before this point stay
// before it, the loads after it stay after it, whatever their types (wave_barrier alone is a scheduling marker, not a
// memory fence, and type-based alias analysis may otherwise move a differently typed load above the stores).
__device__ __forceinline__ void wave_lds_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------
// K0: read starts -> startbits.  One bit per base position; a window is valid iff no read starts inside it
// after its first base.  The sentinel at n_bases ends the last read.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mark_starts(const uint64_t *__restrict__ offsets, uint32_t n_reads,
                                                     uint32_t *__restrict__ startbits)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > n_reads) return;
    const uint64_t p = offsets[i];
    // A start whose neighbours are 16 or more positions away has its word of 16 positions to itself (the words were
    // cleared before): a plain store.  Reads shorter than a word share words and take turns with an atomic.
    const bool alone = (i == 0 || p - offsets[i - 1] >= 16) && (i == n_reads || offsets[i + 1] - p >= 16);
    if (alone) startbits[p >> 4] = 1u << (p & 15);
    else atomicOr(&startbits[p >> 4], 1u << (p & 15));
}

// ------------------------------------------------------------------------------------------------------
// K1: ASCII -> 2-bit, 16 bases per 32-bit word, first base in the top bits (the order of encode_kmer_plain,
// normalise_kmers_multi_large.c:1118-1126; codes of base_map :150-153; N reads as A like replacestr :1406).
// Each lane turns one 16-byte load into one 4-byte store; both sides are fully coalesced, so nothing is
// staged through LDS.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4(uint32_t v)
{
    // A=0x41 C=0x43 G=0x47 T=0x54 N=0x4e : ((c>>1)^(c>>2))&3 -> 0,1,2,3,0
    uint32_t x = ((v >> 1) ^ (v >> 2)) & 0x03030303u;
    return (x * 0x40100401u) >> 24; // byte 0 (first base) lands in the top two bits of the result byte
}

// Alphabet check of four bytes at once (the kernel is bound by its instruction count, not by memory): all of ACGTN
// sit in 0x40-0x5f, where the low five bits tell them apart - bit (c & 31) of a 32-bit set, and a shift takes its
// amount modulo 32 by itself.
__device__ __forceinline__ bool all_acgtn(uint32_t v)
{
    constexpr uint32_t SET = (1u << ('A' & 31)) | (1u << ('C' & 31)) | (1u << ('G' & 31)) | (1u << ('T' & 31)) | (1u << ('N' & 31));
    const uint32_t in_set = (SET >> (v & 31u)) & (SET >> ((v >> 8) & 31u)) & (SET >> ((v >> 16) & 31u)) & (SET >> ((v >> 24) & 31u));
    return (v & 0xe0e0e0e0u) == 0x40404040u && (in_set & 1u);
}

__device__ __forceinline__ uint32_t count_bad4(uint32_t v)
{
    uint32_t bad = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        uint32_t c = (v >> (8 * b)) & 0xffu;
        bad += !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
    }
    return bad;
}

// With startbits != nullptr the pass also clears the batch's start bits and the four halo words behind both arrays
// (the grid then covers n_words + 4 words): a separate fill of the start bits took as long as the packing itself.
__global__ void __launch_bounds__(256) k_pack(const uint8_t *__restrict__ seq, uint64_t n_bases,
                                              uint32_t *__restrict__ packed, uint64_t n_words, Counters *ctr,
                                              uint32_t *__restrict__ startbits)
{
    uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t bad = 0;
    if (startbits && w < n_words + 4) {
        startbits[w] = 0;
        if (w >= n_words) packed[w] = 0;
    }
    if (w < n_words) {
        uint64_t base = w * 16;
        uint32_t v[4];
        if (base + 16 <= n_bases && ((reinterpret_cast<uintptr_t>(seq) & 15) == 0)) {
            uint4 q = *reinterpret_cast<const uint4 *>(seq + base);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint64_t p = base + 4 * i + b;
                    uint32_t c = (p < n_bases) ? seq[p] : (uint32_t)'A';
                    x |= c << (8 * b);
                }
                v[i] = x;
            }
        }
        uint32_t word = (pack4(v[0]) << 24) | (pack4(v[1]) << 16) | (pack4(v[2]) << 8) | pack4(v[3]);
        packed[w] = word;
        if (!(all_acgtn(v[0]) && all_acgtn(v[1]) && all_acgtn(v[2]) && all_acgtn(v[3]))) // rare: count them one by one
            bad = count_bad4(v[0]) + count_bad4(v[1]) + count_bad4(v[2]) + count_bad4(v[3]);
    }
    if (bad) atomicAdd(&ctr->bad_bases, bad);
}

// A batch that arrives packed: its words into the context's array, its start bits cleared, four zero words of halo
// behind both - one pass instead of a fill and a copy.
__global__ void __launch_bounds__(256) k_take_words(const uint32_t *__restrict__ src, uint64_t n_words, uint32_t *__restrict__ packed,
                                                    uint32_t *__restrict__ startbits)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w < n_words + 4) {
        packed[w] = w < n_words ? src[w] : 0u;
        startbits[w] = 0;
    }
}

// ------------------------------------------------------------------------------------------------------
// K2: one lane per packed word = 16 window starts.  The lane needs the next one or two words as well; they
// come from the neighbouring lanes with wavefront shuffles (the last two lanes of a wave load theirs).
// Forward code and reverse-complement code roll from window to window; canonical = the smaller
// (get_canonical_kmer, normalise_kmers_multi_large.c:1175-1180).  Writes 0 where no window of a single read
// starts, or where the code is 0 (skipped by the reference, :1483-1484).
// ------------------------------------------------------------------------------------------------------
// The 16 keys of packed word w.  Every lane of the wavefront must call it (shuffles); lanes with live == false
// get zeros.  Consecutive lanes must hold consecutive words.
// reverses the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t reverse_pairs(uint32_t x)
{
    const uint32_t y = __brev(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

// k <= 16: the 16 windows of a word lie within the word and its successor, so every code is one 32-bit funnel shift of
// the pair (the reverse-complement codes likewise, of the pair complemented and reversed once), and the validity of all
// 16 windows comes out of one bit-parallel pass over the start bits - about a third of the instructions of the rolling
// 64-bit form below, which matters because the kernels that recompute keys from packed words are bound by them.
template <bool CANON>
__device__ __forceinline__ void word_keys32(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits, uint64_t w,
                                            bool live, uint64_t n_bases, int k, uint32_t (&out)[16])
{
    const int lane = threadIdx.x & 63;
    const uint32_t w0 = live ? packed[w] : 0u;   // packed/startbits carry 4 zero words of halo
    const uint32_t s0 = live ? startbits[w] : 0u;
    uint32_t w1 = __shfl_down(w0, 1), s1 = __shfl_down(s0, 1);
    if (live && lane == 63) { w1 = packed[w + 1]; s1 = startbits[w + 1]; }
    if (!live) {
#pragma unroll
        for (int j = 0; j < 16; j++) out[j] = 0u;
        return;
    }
    // windows without a read start after their first base: start bit j+i clear for i = 1..k-1
    const uint32_t m = (uint32_t)k - 1u;
    uint32_t inside = 0;
    if (m) {
        inside = ((s0 & 0xffffu) | (s1 << 16)) >> 1;
        uint32_t s = 1;
        while (2 * s <= m) { inside |= inside >> s; s <<= 1; }
        inside |= inside >> (m - s);
    }
    // ... and that end within the batch: j + k <= rem
    const uint64_t rem = n_bases - w * 16;
    const uint32_t fits = rem >= (uint64_t)(15 + k) ? 0xffffu : (rem < (uint64_t)k ? 0u : (2u << (uint32_t)(rem - (uint64_t)k)) - 1u);
    const uint32_t ok = ~inside & fits;
    const int down = 32 - 2 * k;
    const uint32_t mask = k == 16 ? 0xffffffffu : (1u << (2 * k)) - 1u;
    uint32_t r0 = 0, r1 = 0;
    if (CANON) { r0 = reverse_pairs(~w0); r1 = reverse_pairs(~w1); } // complement of base p at bits 2p, 2p+1 of r1:r0
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t code = (j == 0 ? w0 : __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * j)) >> down;
        if (CANON) {
            const uint32_t rc = (j == 0 ? r0 : __builtin_amdgcn_alignbit(r1, r0, 2 * j)) & mask;
            code = rc < code ? rc : code;
        }
        out[j] = ((ok >> j) & 1u) ? code : 0u;
    }
}

// Half a packed word per lane (half = 0: windows 0-7, 1: windows 8-15), for kernels whose every lane should take part;
// the lane loads the word and its successor itself (no shuffles, so any lane may be skipped).
template <bool CANON>
__device__ __forceinline__ void word_keys32_half(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits, uint64_t w,
                                                 uint32_t half, bool live, uint64_t n_bases, int k, uint32_t (&out)[8])
{
    if (!live) {
#pragma unroll
        for (int j = 0; j < 8; j++) out[j] = 0u;
        return;
    }
    const uint32_t w0 = packed[w], w1 = packed[w + 1], s0 = startbits[w], s1 = startbits[w + 1]; // 4 zero words of halo
    const uint32_t m = (uint32_t)k - 1u;
    uint32_t inside = 0;
    if (m) {
        inside = ((s0 & 0xffffu) | (s1 << 16)) >> 1;
        uint32_t s = 1;
        while (2 * s <= m) { inside |= inside >> s; s <<= 1; }
        inside |= inside >> (m - s);
    }
    const uint64_t rem = n_bases - w * 16;
    const uint32_t fits = rem >= (uint64_t)(15 + k) ? 0xffffu : (rem < (uint64_t)k ? 0u : (2u << (uint32_t)(rem - (uint64_t)k)) - 1u);
    const uint32_t ok = (~inside & fits) >> (8u * half);
    const int down = 32 - 2 * k;
    const uint32_t mask = k == 16 ? 0xffffffffu : (1u << (2 * k)) - 1u;
    const unsigned long long fw = ((unsigned long long)w0 << 32) | w1;
    unsigned long long rv = 0;
    if (CANON) rv = ((unsigned long long)reverse_pairs(~w1) << 32) | reverse_pairs(~w0);
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        const uint32_t j = 8u * half + (uint32_t)jj;
        uint32_t code = (uint32_t)(fw >> (32u - 2u * j)) >> down;
        if (CANON) {
            const uint32_t rc = (uint32_t)(rv >> (2u * j)) & mask;
            code = rc < code ? rc : code;
        }
        out[jj] = ((ok >> jj) & 1u) ? code : 0u;
    }
}

template <typename KEY, bool CANON>
__device__ __forceinline__ void word_keys(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits, uint64_t w,
                                          bool live, uint64_t n_bases, int k, KEY (&out)[16])
{
    if constexpr (sizeof(KEY) == 4) { // 32-bit keys are only used for k <= 16
        word_keys32<CANON>(packed, startbits, w, live, n_bases, k, out);
        return;
    }
    int lane = threadIdx.x & 63;
    uint32_t w0 = live ? packed[w] : 0u;         // packed/startbits carry 4 zero words of halo
    uint32_t s0 = live ? startbits[w] : 0u;
    uint32_t w1 = __shfl_down(w0, 1), w2 = __shfl_down(w0, 2);
    uint32_t s1 = __shfl_down(s0, 1), s2 = __shfl_down(s0, 2);
    if (live && lane >= 62) {
        if (lane == 63) { w1 = packed[w + 1]; s1 = startbits[w + 1]; }
        w2 = packed[w + 2]; s2 = startbits[w + 2];
    }
    if (!live) {
#pragma unroll
        for (int j = 0; j < 16; j++) out[j] = (KEY)0;
        return;
    }

    const uint64_t S = (uint64_t)s0 | ((uint64_t)s1 << 16) | ((uint64_t)s2 << 32);
    const uint64_t span = (1ull << (k - 1)) - 1;             // positions 1..k-1 after the window start
    const uint64_t hi = ((uint64_t)w0 << 32) | w1;           // bases 0..31 of this lane's 48
    const uint64_t mask = (1ull << (2 * k)) - 1;             // k <= 31
    // first window: bases 0..k-1 (k <= 31 < 32, so they all sit in hi)
    uint64_t fwd = hi >> (64 - 2 * k), rc = 0;
    if (CANON) {
        // reverse complement of fwd: reverse the 2-bit groups, complement
        uint64_t x = ~fwd & mask;
        uint64_t r = 0;
        for (int i = 0; i < k; i++) { r = (r << 2) | (x & 3ull); x >>= 2; }
        rc = r;
    }
    const uint64_t rem = n_bases - w * 16; // bases from this word's first position to the end of the batch
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j > 0) {
            int p = j + k - 1; // incoming base position, 1..46
            uint64_t b = (p < 32) ? ((hi >> (62 - 2 * p)) & 3ull) : (((uint64_t)w2 >> (94 - 2 * p)) & 3ull);
            fwd = ((fwd << 2) | b) & mask;
            if (CANON) rc = (rc >> 2) | ((3ull - b) << (2 * (k - 1)));
        }
        bool valid = ((S >> (j + 1)) & span) == 0 && (uint64_t)(j + k) <= rem;
        uint64_t code = fwd;
        if (CANON) code = rc < fwd ? rc : fwd;
        out[j] = valid ? (KEY)code : (KEY)0;
    }
}

template <typename KEY, bool CANON>
__global__ void __launch_bounds__(256) k_kmerise(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits,
                                                 uint64_t n_bases, uint64_t n_words, int k, KEY *__restrict__ keys)
{
    uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool live = w < n_words;
    KEY out[16];
    word_keys<KEY, CANON>(packed, startbits, w, live, n_bases, k, out);
    if (!live) return;
    KEY *dst = keys + w * 16;
    if (sizeof(KEY) == 4) {
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
        for (int q = 0; q < 4; q++) d4[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    } else {
        ulonglong2 *d2 = reinterpret_cast<ulonglong2 *>(dst);
#pragma unroll
        for (int q = 0; q < 8; q++) d2[q] = make_ulonglong2(out[2 * q], out[2 * q + 1]);
    }
}

// ------------------------------------------------------------------------------------------------------
// Block-level sum of three per-thread tallies, then one atomic each (a per-occurrence atomic on one word
// would serialise the whole chip).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

__device__ __forceinline__ void flush_tallies(uint32_t uniq, uint32_t counted, uint32_t sat, Counters *ctr)
{
    __shared__ uint32_t red[3][4];
    uniq = wave_sum(uniq); counted = wave_sum(counted); sat = wave_sum(sat);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = uniq; red[1][wv] = counted; red[2][wv] = sat; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        uint32_t b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        uint32_t c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (a) atomicAdd(&ctr->unique, (unsigned long long)a);
        if (b) atomicAdd(&ctr->counted, (unsigned long long)b);
        if (c) atomicAdd(&ctr->saturated, (unsigned long long)c);
    }
}

// ------------------------------------------------------------------------------------------------------
// Appends to the device-wide lists go through a per-workgroup list in LDS that is emptied with one global
// atomic per flush: a single counter word takes under 100 M atomics a second chip-wide, far less than the
// rate at which saturation / crossing events and crossing occurrences turn up.
// ------------------------------------------------------------------------------------------------------
template <typename T, int CAP>
struct BlockList {
    T items[CAP];
    uint32_t n;
    uint32_t base;
};

template <typename T, int CAP>
__device__ __forceinline__ void blocklist_push(BlockList<T, CAP> &bl, T v, T *__restrict__ global_list, unsigned int *global_n)
{
    uint32_t at = atomicAdd(&bl.n, 1u);
    if (at < CAP) bl.items[at] = v;
    else global_list[atomicAdd(global_n, 1u)] = v; // LDS list full: rare, go straight to the global list
}

// all threads of the workgroup must call this (it synchronises)
template <typename T, int CAP>
__device__ __forceinline__ void blocklist_flush(BlockList<T, CAP> &bl, T *__restrict__ global_list, unsigned int *global_n)
{
    __syncthreads();
    uint32_t n = bl.n < CAP ? bl.n : CAP;
    if (threadIdx.x == 0 && n) bl.base = atomicAdd(global_n, n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) global_list[bl.base + i] = bl.items[i];
    __syncthreads();
    if (threadIdx.x == 0) bl.n = 0;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------
// K3 (k <= 16): count one chunk of keys against the direct-addressed table.  Restates store_kmer's insert /
// increment (normalise_kmers_multi_large.c:948-1008) and the high test (:1494-1497).  Four keys per lane: the
// four bitmap gathers, then the four atomic adds, are independent and in flight together.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_direct(const uint32_t *__restrict__ keys, uint64_t n4, uint8_t *__restrict__ flags,
                                                      uint32_t *__restrict__ counts, const uint32_t *__restrict__ sat,
                                                      uint32_t *__restrict__ filter, uint32_t depth, uint32_t *__restrict__ extra,
                                                      Counters *ctr, uint32_t *__restrict__ list_sat,
                                                      uint32_t *__restrict__ list_cross)
{
    __shared__ BlockList<uint32_t, 1024> bl_sat, bl_cross;
    if (threadIdx.x == 0) { bl_sat.n = 0; bl_cross.n = 0; }
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t uniq = 0, counted = 0, nsat = 0;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
        uint4 kv = reinterpret_cast<const uint4 *>(keys)[q];
        uint32_t key[4] = {kv.x, kv.y, kv.z, kv.w};
        uint32_t sw[4], old[4], fl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) sw[u] = key[u] ? sat[key[u] >> 5] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool is_sat = (sw[u] >> (key[u] & 31)) & 1u;
            old[u] = 0;
            fl[u] = FLAG_NONE;
            if (key[u]) {
                counted++;
                if (is_sat) {
                    fl[u] = FLAG_HIGH;
                    nsat++;
                    if (extra) atomicAdd(&extra[key[u]], 1u); // counts[] stays frozen: the ranking pass relies on it
                } else {
                    fl[u] = FLAG_LOW;
                    old[u] = atomicAdd(&counts[key[u]], 1u);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (fl[u] == FLAG_LOW) {
                uint32_t c1 = (old[u] & COUNT_MASK) + 1;
                if (old[u] == 0) uniq++;
                if (c1 == depth - 1) blocklist_push(bl_sat, key[u], list_sat, &ctr->n_sat_new);
                if (c1 == depth) {
                    blocklist_push(bl_cross, key[u], list_cross, &ctr->n_cross);
                    filter_set(filter, filter_index(key[u]));
                }
            }
        }
        reinterpret_cast<uint32_t *>(flags)[q] = fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24);
    }
    blocklist_flush(bl_sat, list_sat, &ctr->n_sat_new);
    blocklist_flush(bl_cross, list_cross, &ctr->n_cross);
    flush_tallies(uniq, counted, nsat, ctr);
}


// K4a: right after a chunk's counting pass, collect the occurrences of the k-mers that crossed in it:
// (id << posbits) | position-in-batch.  The filter screens, the count (>= depth, and the occurrence was not
// saturated when the pass began) confirms, so the list is exact.  Nothing runs if no k-mer crossed.
// Hits are staged in LDS and flushed whenever another full iteration might not fit.
constexpr int OCC_CAP = 6144; // 48 KB of LDS: room for one more full iteration (256 lanes x 16 windows) past the flush threshold
using OccList = BlockList<unsigned long long, OCC_CAP>;

__global__ void __launch_bounds__(256) k_collect_direct(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ flags,
                                                        uint64_t n4, uint64_t pos_base, const uint32_t *__restrict__ filter,
                                                        const uint32_t *__restrict__ counts, uint32_t depth, int posbits,
                                                        unsigned long long *__restrict__ occ, Counters *ctr)
{
    __shared__ OccList bl;
    if (ctr->n_cross == 0) return; // uniform: written by the previous kernel
    if (threadIdx.x == 0) bl.n = 0;
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t n4r = (n4 + stride - 1) / stride * stride; // same trip count for every lane: the flush synchronises
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n4r; q += stride) {
        if (q < n4) {
            uint32_t f4 = reinterpret_cast<const uint32_t *>(flags)[q];
            if (f4 & 0x02020202u) {
                uint4 kv = reinterpret_cast<const uint4 *>(keys)[q];
                uint32_t key[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    bool low = ((f4 >> (8 * u)) & 0xffu) == FLAG_LOW;
                    if (low && filter_test(filter, filter_index(key[u])) && (counts[key[u]] & COUNT_MASK) >= depth)
                        blocklist_push(bl, ((unsigned long long)key[u] << posbits) | (pos_base + q * 4 + u), occ, &ctr->n_occ);
                }
            }
        }
        __syncthreads();
        uint32_t fill = bl.n;
        __syncthreads(); // every lane has read the same fill level before anyone pushes again
        if (fill > OCC_CAP - 1024) blocklist_flush(bl, occ, &ctr->n_occ);
    }
    blocklist_flush(bl, occ, &ctr->n_occ);
}

// K4b: exact stream-order ranks for the sorted occurrences.  occ is sorted by (id, position); an id's group is
// found by two binary searches; the i-th member (0-based) of a group of m whose k-mer ended the pass at count F
// has rank F - m + i + 1.
template <typename TABLE>
__global__ void __launch_bounds__(256) k_rank(const unsigned long long *__restrict__ occ, uint32_t n_occ, int posbits,
                                              TABLE table, uint32_t depth, uint8_t *__restrict__ flags)
{
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_occ) return;
    unsigned long long e = occ[s];
    unsigned long long id = e >> posbits;
    unsigned long long lo_key = id << posbits, hi_key = (id + 1) << posbits;
    uint32_t lo = 0, hi = s; // first index with occ >= lo_key, in [0, s]
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (occ[mid] < lo_key) lo = mid + 1; else hi = mid; }
    uint32_t start = lo;
    lo = s + 1; hi = n_occ;  // first index with occ >= hi_key, in (s, n_occ]
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (occ[mid] < hi_key) lo = mid + 1; else hi = mid; }
    uint32_t m = lo - start, i = s - start;
    uint32_t F = table.count_of(id);
    if (F - m + i + 1 >= depth) flags[e & ((1ull << posbits) - 1)] = (uint8_t)FLAG_HIGH;
}

struct DirectTable {
    const uint32_t *counts;
    __device__ uint32_t count_of(unsigned long long id) const { return counts[id] & COUNT_MASK; }
};
struct HashedTable {
    const Slot *slots;
    __device__ uint32_t count_of(unsigned long long id) const { return slots[id].count; }
};

// K4c: between passes, k-mers whose count reached depth-1 or depth become saturated.  List lengths are read
// on the device, so the host never waits for them.
__global__ void __launch_bounds__(256) k_apply_direct(const uint32_t *__restrict__ list_sat, const uint32_t *__restrict__ list_cross,
                                                      uint32_t *__restrict__ sat, const Counters *ctr)
{
    uint32_t n_sat = ctr->n_sat_new, n_cross = ctr->n_cross;
    uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_sat; i += stride) { uint32_t k = list_sat[i]; atomicOr(&sat[k >> 5], 1u << (k & 31)); }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_cross; i += stride) { uint32_t k = list_cross[i]; atomicOr(&sat[k >> 5], 1u << (k & 31)); }
}

// end of a chunk: the two list lengths restart at zero, the running totals keep them for the statistics
__global__ void k_end_chunk(Counters *ctr)
{
    ctr->total_cross += ctr->n_cross;
    ctr->n_sat_new = 0;
    ctr->n_cross = 0;
}

// ------------------------------------------------------------------------------------------------------
// K3 (k > 16): hashed table.  One lane per key, linear probing from a mixed hash; the key is claimed with a
// 64-bit compare-and-swap (store_kmer's insert, normalise_kmers_multi_large.c:948-971), the count bumped with
// an atomic add (:1003).  A plain load may show a stale "empty" - the CAS then reports the true owner - but a
// non-zero key never changes, so a matching or foreign key read by a plain load can be trusted.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// returns the slot index of key (inserting it if absent), or ~0 if the probe wrapped; *inserted tells a new key
__device__ __forceinline__ uint64_t find_or_insert(Slot *__restrict__ slots, uint64_t cap_mask, uint64_t key, bool &inserted,
                                                   uint32_t &state)
{
    uint64_t h = mix64(key) & cap_mask;
    inserted = false;
    for (uint64_t probes = 0; probes <= cap_mask; probes++) {
        uint4 raw = *reinterpret_cast<const uint4 *>(&slots[h]);
        unsigned long long cur = ((unsigned long long)raw.y << 32) | raw.x;
        state = raw.w;
        if (cur == 0) {
            cur = atomicCAS(&slots[h].key, 0ull, (unsigned long long)key);
            if (cur == 0) { inserted = true; state = 0; return h; }
            if (cur == key) { state = 0; return h; } // inserted by another lane during this pass: not saturated yet
        }
        if (cur == key) return h;
        h = (h + 1) & cap_mask;
    }
    return ~0ull;
}

__global__ void __launch_bounds__(256) k_count_hashed(const unsigned long long *__restrict__ keys, uint64_t n, uint8_t *__restrict__ flags,
                                                      Slot *__restrict__ slots,
                                                      uint64_t cap_mask, uint32_t *__restrict__ filter, uint32_t depth,
                                                      int count_saturated, Counters *ctr,
                                                      unsigned long long *__restrict__ list_sat,
                                                      unsigned long long *__restrict__ list_cross)
{
    __shared__ BlockList<unsigned long long, 512> bl_sat, bl_cross;
    if (threadIdx.x == 0) { bl_sat.n = 0; bl_cross.n = 0; }
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t uniq = 0, counted = 0, nsat = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long key = keys[i];
        uint32_t fl = FLAG_NONE;
        if (key) {
            counted++;
            bool inserted; uint32_t state;
            uint64_t h = find_or_insert(slots, cap_mask, key, inserted, state);
            if (h == ~0ull) { atomicAdd(&ctr->probe_overflow, 1u); }
            else {
                if (inserted) uniq++;
                if (state & ST_SAT) {
                    fl = FLAG_HIGH;
                    nsat++;
                    if (count_saturated) atomicAdd(&slots[h].state, 1u); // slot.count stays frozen for the ranking pass
                } else {
                    fl = FLAG_LOW;
                    uint32_t c1 = atomicAdd(&slots[h].count, 1u) + 1;
                    if (c1 == depth - 1) blocklist_push(bl_sat, (unsigned long long)h, list_sat, &ctr->n_sat_new);
                    if (c1 == depth) {
                        blocklist_push(bl_cross, (unsigned long long)h, list_cross, &ctr->n_cross);
                        filter_set(filter, filter_index(key));
                    }
                }
            }
        }
        flags[i] = (uint8_t)fl;
    }
    blocklist_flush(bl_sat, list_sat, &ctr->n_sat_new);
    blocklist_flush(bl_cross, list_cross, &ctr->n_cross);
    flush_tallies(uniq, counted, nsat, ctr);
}


// read-only lookup: slot of a key that is known to be stored, ~0 if it is not
__device__ __forceinline__ uint64_t find_slot(const Slot *__restrict__ slots, uint64_t cap_mask, unsigned long long key)
{
    uint64_t h = mix64(key) & cap_mask;
    for (uint64_t probes = 0; probes <= cap_mask; probes++) {
        unsigned long long cur = slots[h].key;
        if (cur == key) return h;
        if (cur == 0) return ~0ull;
        h = (h + 1) & cap_mask;
    }
    return ~0ull;
}

__global__ void __launch_bounds__(256) k_collect_hashed(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ flags,
                                                        uint64_t n, uint64_t pos_base, const uint32_t *__restrict__ filter,
                                                        const Slot *__restrict__ slots, uint64_t cap_mask, uint32_t depth, int posbits,
                                                        unsigned long long *__restrict__ occ, Counters *ctr)
{
    __shared__ OccList bl;
    if (ctr->n_cross == 0) return;
    if (threadIdx.x == 0) bl.n = 0;
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t nr = (n + stride - 1) / stride * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nr; i += stride) {
        if (i < n && flags[i] == FLAG_LOW) {
            unsigned long long key = keys[i];
            if (filter_test(filter, filter_index(key))) {
                uint64_t h = find_slot(slots, cap_mask, key);
                if (h != ~0ull && slots[h].count >= depth) blocklist_push(bl, (unsigned long long)(h << posbits) | (pos_base + i), occ, &ctr->n_occ);
            }
        }
        __syncthreads();
        uint32_t fill = bl.n;
        __syncthreads();
        if (fill > OCC_CAP - 256) blocklist_flush(bl, occ, &ctr->n_occ);
    }
    blocklist_flush(bl, occ, &ctr->n_occ);
}

__global__ void __launch_bounds__(256) k_apply_hashed(const unsigned long long *__restrict__ list_sat,
                                                      const unsigned long long *__restrict__ list_cross, Slot *__restrict__ slots,
                                                      const Counters *ctr)
{
    uint32_t n_sat = ctr->n_sat_new, n_cross = ctr->n_cross;
    uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_sat; i += stride) atomicOr(&slots[list_sat[i]].state, ST_SAT);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_cross; i += stride) atomicOr(&slots[list_cross[i]].state, ST_SAT);
}

// ------------------------------------------------------------------------------------------------------
// Fused single-GPU form of K2+K3 (+K4a): the keys of a packed word are made in registers and counted at once,
// so the per-position key and flag arrays never exist.  Per packed word one 32-bit result: bit j = window j is
// HIGH, bit 16+j = window j was counted below depth in its pass (LOW).  One lane = one word = up to 16
// occurrences, whose bitmap gathers and atomic adds are all independent.
// ------------------------------------------------------------------------------------------------------
template <bool CANON>
__global__ void __launch_bounds__(256) k_count_words_direct(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits,
                                                            uint64_t n_bases, uint64_t w_begin, uint64_t w_end, int k,
                                                            uint32_t *__restrict__ wflags, uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ sat, uint32_t *__restrict__ filter, uint32_t depth,
                                                            uint32_t *__restrict__ extra, Counters *ctr, uint32_t *__restrict__ list_sat,
                                                            uint32_t *__restrict__ list_cross)
{
    __shared__ BlockList<uint32_t, 1024> bl_sat, bl_cross;
    if (threadIdx.x == 0) { bl_sat.n = 0; bl_cross.n = 0; }
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256, n = w_end - w_begin, n_words_total = (n_bases + 15) / 16;
    const uint64_t nr = (n + stride - 1) / stride * stride; // whole wavefronts iterate together (shuffles inside)
    uint32_t uniq = 0, counted = 0, nsat = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nr; i += stride) {
        const bool live = i < n;
        const uint64_t w = w_begin + i;
        uint32_t key[16], sw[16], old[16];
        word_keys<uint32_t, CANON>(packed, startbits, w, w < n_words_total, n_bases, k, key);
        if (!live) {
#pragma unroll
            for (int j = 0; j < 16; j++) key[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) sw[j] = key[j] ? sat[key[j] >> 5] : 0u;
        uint32_t hi = 0, lo = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            old[j] = 0;
            if (key[j]) {
                counted++;
                if ((sw[j] >> (key[j] & 31)) & 1u) {
                    hi |= 1u << j;
                    nsat++;
                    if (extra) atomicAdd(&extra[key[j]], 1u);
                } else {
                    lo |= 1u << j;
                    old[j] = atomicAdd(&counts[key[j]], 1u);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if ((lo >> j) & 1u) {
                uint32_t c1 = (old[j] & COUNT_MASK) + 1;
                if (old[j] == 0) uniq++;
                if (c1 == depth - 1) blocklist_push(bl_sat, key[j], list_sat, &ctr->n_sat_new);
                if (c1 == depth) {
                    blocklist_push(bl_cross, key[j], list_cross, &ctr->n_cross);
                    filter_set(filter, filter_index(key[j]));
                }
            }
        }
        if (live) wflags[w] = hi | (lo << 16);
    }
    blocklist_flush(bl_sat, list_sat, &ctr->n_sat_new);
    blocklist_flush(bl_cross, list_cross, &ctr->n_cross);
    flush_tallies(uniq, counted, nsat, ctr);
}

template <bool CANON>
__global__ void __launch_bounds__(256) k_count_words_hashed(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits,
                                                            uint64_t n_bases, uint64_t w_begin, uint64_t w_end, int k,
                                                            uint32_t *__restrict__ wflags, Slot *__restrict__ slots, uint64_t cap_mask,
                                                            uint32_t *__restrict__ filter, uint32_t depth, int count_saturated, Counters *ctr,
                                                            unsigned long long *__restrict__ list_sat,
                                                            unsigned long long *__restrict__ list_cross)
{
    __shared__ BlockList<unsigned long long, 512> bl_sat, bl_cross;
    if (threadIdx.x == 0) { bl_sat.n = 0; bl_cross.n = 0; }
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256, n = w_end - w_begin, n_words_total = (n_bases + 15) / 16;
    const uint64_t nr = (n + stride - 1) / stride * stride;
    uint32_t uniq = 0, counted = 0, nsat = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nr; i += stride) {
        const bool live = i < n;
        const uint64_t w = w_begin + i;
        unsigned long long key[16];
        word_keys<unsigned long long, CANON>(packed, startbits, w, w < n_words_total, n_bases, k, key);
        if (!live) {
#pragma unroll
            for (int j = 0; j < 16; j++) key[j] = 0;
        }
        uint32_t hi = 0, lo = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (!key[j]) continue;
            counted++;
            bool inserted; uint32_t state;
            uint64_t h = find_or_insert(slots, cap_mask, key[j], inserted, state);
            if (h == ~0ull) { atomicAdd(&ctr->probe_overflow, 1u); continue; }
            if (inserted) uniq++;
            if (state & ST_SAT) {
                hi |= 1u << j;
                nsat++;
                if (count_saturated) atomicAdd(&slots[h].state, 1u);
            } else {
                lo |= 1u << j;
                uint32_t c1 = atomicAdd(&slots[h].count, 1u) + 1;
                if (c1 == depth - 1) blocklist_push(bl_sat, (unsigned long long)h, list_sat, &ctr->n_sat_new);
                if (c1 == depth) {
                    blocklist_push(bl_cross, (unsigned long long)h, list_cross, &ctr->n_cross);
                    filter_set(filter, filter_index(key[j]));
                }
            }
        }
        if (live) wflags[w] = hi | (lo << 16);
    }
    blocklist_flush(bl_sat, list_sat, &ctr->n_sat_new);
    blocklist_flush(bl_cross, list_cross, &ctr->n_cross);
    flush_tallies(uniq, counted, nsat, ctr);
}

// Seeding (sequence_to_hash_zero, normalise_kmers_multi_large.c:1501-1537): keys become present, counts are left alone
template <bool CANON, bool DIRECT>
__global__ void __launch_bounds__(256) k_seed_words(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits,
                                                    uint64_t n_bases, uint64_t n_words, int k, uint32_t *__restrict__ counts,
                                                    Slot *__restrict__ slots, uint64_t cap_mask, Counters *ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const uint64_t nr = (n_words + stride - 1) / stride * stride;
    uint32_t uniq = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nr; w += stride) {
        const bool live = w < n_words;
        if (DIRECT) {
            uint32_t key[16];
            word_keys<uint32_t, CANON>(packed, startbits, w, live, n_bases, k, key);
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (key[j] && atomicOr(&counts[key[j]], SEEDED_BIT) == 0) uniq++;
        } else {
            unsigned long long key[16];
            word_keys<unsigned long long, CANON>(packed, startbits, w, live, n_bases, k, key);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (!key[j]) continue;
                bool inserted; uint32_t state;
                uint64_t h = find_or_insert(slots, cap_mask, key[j], inserted, state);
                if (h == ~0ull) atomicAdd(&ctr->probe_overflow, 1u);
                else if (inserted) uniq++;
            }
        }
    }
    flush_tallies(uniq, 0, 0, ctr);
}

// the same insert for keys that are already made (the owner side of the multi-device exchange)
template <typename KEY, bool DIRECT>
__global__ void __launch_bounds__(256) k_seed_keys(const KEY *__restrict__ keys, uint64_t n, uint32_t *__restrict__ counts, Slot *__restrict__ slots,
                                                   uint64_t cap_mask, Counters *ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t uniq = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const KEY key = keys[i];
        if (!key) continue;
        if (DIRECT) {
            if (atomicOr(&counts[key], SEEDED_BIT) == 0) uniq++;
        } else {
            bool inserted; uint32_t state;
            const uint64_t h = find_or_insert(slots, cap_mask, (uint64_t)key, inserted, state);
            if (h == ~0ull) atomicAdd(&ctr->probe_overflow, 1u);
            else if (inserted) uniq++;
        }
    }
    flush_tallies(uniq, 0, 0, ctr);
}

// K4a on word flags: occurrences marked LOW whose k-mer crossed in this pass -> (id << posbits) | position.
template <bool CANON, bool DIRECT>
__global__ void __launch_bounds__(256) k_collect_words(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ startbits,
                                                       uint64_t n_bases, uint64_t w_begin, uint64_t w_end, int k,
                                                       const uint32_t *__restrict__ wflags, const uint32_t *__restrict__ filter,
                                                       const uint32_t *__restrict__ counts, const Slot *__restrict__ slots, uint64_t cap_mask,
                                                       uint32_t depth, int posbits, unsigned long long *__restrict__ occ, Counters *ctr)
{
    __shared__ OccList bl;
    if (ctr->n_cross == 0) return; // uniform: written by the previous kernel
    if (threadIdx.x == 0) bl.n = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256, n = w_end - w_begin, n_words_total = (n_bases + 15) / 16;
    const uint64_t nr = (n + stride - 1) / stride * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nr; i += stride) {
        const bool live = i < n;
        const uint64_t w = w_begin + i;
        const uint32_t lo = live ? (wflags[w] >> 16) : 0u;
        if (__any(lo != 0)) { // the whole wavefront takes part in the shuffles of word_keys
            if (DIRECT) {
                uint32_t key[16];
                word_keys<uint32_t, CANON>(packed, startbits, w, w < n_words_total, n_bases, k, key);
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (((lo >> j) & 1u) && filter_test(filter, filter_index(key[j])) && (counts[key[j]] & COUNT_MASK) >= depth)
                        blocklist_push(bl, ((unsigned long long)key[j] << posbits) | (w * 16 + j), occ, &ctr->n_occ);
            } else {
                unsigned long long key[16];
                word_keys<unsigned long long, CANON>(packed, startbits, w, w < n_words_total, n_bases, k, key);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (((lo >> j) & 1u) && filter_test(filter, filter_index(key[j]))) {
                        uint64_t h = find_slot(slots, cap_mask, key[j]);
                        if (h != ~0ull && slots[h].count >= depth)
                            blocklist_push(bl, (unsigned long long)(h << posbits) | (w * 16 + j), occ, &ctr->n_occ);
                    }
                }
            }
        }
        __syncthreads();
        uint32_t fill = bl.n;
        __syncthreads(); // every lane has read the same fill level before anyone pushes again
        if (fill > OCC_CAP - 4096) blocklist_flush(bl, occ, &ctr->n_occ);
    }
    blocklist_flush(bl, occ, &ctr->n_occ);
}

// K4b for word flags: same ranking as k_rank, the HIGH bit of the occurrence's word is set with an atomic OR
template <typename TABLE>
__global__ void __launch_bounds__(256) k_rank_words(const unsigned long long *__restrict__ occ, uint32_t n_occ, int posbits, TABLE table,
                                                    uint32_t depth, uint32_t *__restrict__ wflags)
{
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_occ) return;
    unsigned long long e = occ[s];
    unsigned long long id = e >> posbits;
    unsigned long long lo_key = id << posbits, hi_key = (id + 1) << posbits;
    uint32_t lo = 0, hi = s;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (occ[mid] < lo_key) lo = mid + 1; else hi = mid; }
    uint32_t start = lo;
    lo = s + 1; hi = n_occ;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (occ[mid] < hi_key) lo = mid + 1; else hi = mid; }
    uint32_t m = lo - start, i = s - start;
    uint32_t F = table.count_of(id);
    if (F - m + i + 1 >= depth) {
        unsigned long long pos = e & ((1ull << posbits) - 1);
        atomicOr(&wflags[pos >> 4], 1u << (pos & 15));
    }
}

// K5 for word flags: one lane per read; high = HIGH bits inside the read, total = HIGH or LOW bits inside it
__global__ void __launch_bounds__(256) k_reduce_words(const uint32_t *__restrict__ wflags, const uint64_t *__restrict__ offsets,
                                                      uint32_t n_reads, int32_t *__restrict__ high, int32_t *__restrict__ total)
{
    uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t a = offsets[r], b = offsets[r + 1];
    uint32_t h = 0, t = 0;
    for (uint64_t w = a >> 4; w * 16 < b; w++) {
        uint32_t f = wflags[w];
        uint32_t m = 0xffffu;
        if (w * 16 < a) m &= 0xffffu << (a - w * 16);
        if (w * 16 + 16 > b) m &= 0xffffu >> (w * 16 + 16 - b);
        uint32_t hb = f & m, lb = (f >> 16) & m;
        h += __popc(hb);
        t += __popc(hb | lb);
    }
    high[r] = (int32_t)h;
    total[r] = (int32_t)t;
}

// Growth (expand_local_hash_table, normalise_kmers_multi_large.c:1055-1108): every stored key moves to a table
// twice the size.  Keys are distinct, so after the CAS has claimed a slot its other fields are written plainly.
__global__ void __launch_bounds__(256) k_rehash(const Slot *__restrict__ old_slots, uint64_t old_cap, Slot *__restrict__ new_slots,
                                                uint64_t new_mask, Counters *ctr)
{
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < old_cap; i += stride) {
        Slot s = old_slots[i];
        if (!s.key) continue;
        uint64_t h = mix64(s.key) & new_mask;
        uint64_t probes = 0;
        for (; probes <= new_mask; probes++) {
            if (atomicCAS(&new_slots[h].key, 0ull, s.key) == 0ull) {
                new_slots[h].count = s.count;
                new_slots[h].state = s.state;
                break;
            }
            h = (h + 1) & new_mask;
        }
        if (probes > new_mask) atomicAdd(&ctr->probe_overflow, 1u);
    }
}

// ------------------------------------------------------------------------------------------------------
// K5: per read, high = flags equal to HIGH, total = flags that are not NONE, i.e. windows with a non-zero code
// (sequence_to_hash's two counters, normalise_kmers_multi_large.c:1486, 1496).  Sixteen lanes per read.
__global__ void __launch_bounds__(256) k_reduce_reads(const uint8_t *__restrict__ flags, const uint64_t *__restrict__ offsets,
                                                      uint32_t n_reads, int32_t *__restrict__ high, int32_t *__restrict__ total)
{
    const uint32_t r = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = r < n_reads; // no early exit: the lanes of a wavefront shuffle together below
    const int lane = threadIdx.x & 15;
    const uint64_t a = live ? offsets[r] : 0, b = live ? offsets[r + 1] : 0;
    uint32_t h = 0, t = 0;
    if ((reinterpret_cast<uintptr_t>(flags) & 3u) == 0) {
        // four flags per lane and load (aligned words; bytes outside [a, b) are masked off): three rounds for a read
        // of 150 bases.  FLAG_HIGH = 1, FLAG_LOW = 2, so per byte bit 0 = high and (bit 0 | bit 1) = counted.
        static_assert(FLAG_NONE == 0 && FLAG_HIGH == 1 && FLAG_LOW == 2, "flag encoding");
        for (uint64_t p = (a & ~3ull) + 4ull * lane; p < b; p += 64) {
            uint32_t w = *reinterpret_cast<const uint32_t *>(flags + p);
            uint32_t m = 0xffffffffu;
            if (p < a) m &= 0xffffffffu << (8 * (a - p));
            if (p + 4 > b) m &= 0xffffffffu >> (8 * (p + 4 - b));
            w &= m;
            h += __popc(w & 0x01010101u);
            t += __popc((w | (w >> 1)) & 0x01010101u);
        }
    } else {
        for (uint64_t p = a + lane; p < b; p += 16) {
            uint32_t f = flags[p];
            h += f == FLAG_HIGH;
            t += f != FLAG_NONE;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        h += __shfl_xor(h, o);
        t += __shfl_xor(t, o);
    }
    if (live && lane == 0) { high[r] = (int32_t)h; total[r] = (int32_t)t; }
}

// Keep/skip, float32, strict '<' (normalise_kmers_multi_large.c:1641-1646, 1988-1992).  __fdiv_rn is the IEEE
// round-to-nearest divide the host's (float)high / total performs.
__device__ __forceinline__ float ratio_f32(int32_t high, int32_t total)
{
    return total > 0 ? __fdiv_rn((float)high, (float)total) : 0.0f;
}

__global__ void __launch_bounds__(256) k_decide(const int32_t *__restrict__ high, const int32_t *__restrict__ total, uint32_t n_units,
                                                int paired, float coverage, uint8_t *__restrict__ keep)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_units) return;
    if (paired) {
        float rf = ratio_f32(high[2 * i], total[2 * i]), rr = ratio_f32(high[2 * i + 1], total[2 * i + 1]);
        keep[i] = (rf < coverage && rr < coverage) ? 1 : 0;
    } else {
        keep[i] = ratio_f32(high[i], total[i]) < coverage ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------------
// Sharded table (several GPUs): every key has an owning shard; keys are brought into destination order with a
// stable partition, so that what an owner receives from each source is still in that source's stream order.
// One wavefront owns ROUTE_TILE consecutive keys: pass 1 counts them per shard, an exclusive scan over
// [shard][wave] turns the counts into start offsets, pass 2 places each key at offset + (number of earlier keys
// of the same shard in the tile), computed with ballots.  Zero keys (no k-mer) are not sent.
// ------------------------------------------------------------------------------------------------------
constexpr int ROUTE_TILE = 1024;
constexpr uint32_t ROUTE_BALLOT_SHARDS = 2; // up to here a ballot per shard and round ranks the keys, beyond it LDS adds do
constexpr uint32_t ROUTE_META_NONE = 0x8000u, ROUTE_META_HOME = 0x8001u, ROUTE_RANKED_MAX_SHARDS = 32; // (bit 15 free: shard < 32)

__host__ __device__ __forceinline__ uint32_t owner_of(unsigned long long key, uint32_t n_shards)
{
    // 32 mixed bits of the key, scaled to [0, n_shards) with a multiply (a modulo by a run-time divisor costs a division per key)
    return (uint32_t)((((key * 0x9E3779B97F4A7C15ull) >> 32) * n_shards) >> 32);
}

// Where a routing tile's keys come from: a key array (one key per base position), or the packed words of the
// batch (keys recomputed, never stored: 16 per lane, handed over in position order through LDS).
struct RouteSource {
    const void *keys;           // SRC == 0
    const uint32_t *packed;     // SRC != 0
    const uint32_t *startbits;
    uint64_t n_bases;
    int k;
};
constexpr int ROUTE_STAGE = ROUTE_TILE + ROUTE_TILE / 16; // 17 slots per packed word: conflict-free both ways

template <typename KEY, int SRC>
__device__ __forceinline__ void route_stage_tile(const RouteSource &src, uint64_t wv, uint64_t n, bool live_wave, KEY *stage)
{
    if (SRC != 0) {
        const int lane = threadIdx.x & 63;
        if (live_wave) { // wavefront-uniform: every lane takes part in the shuffles of word_keys
            const uint64_t w = wv * (ROUTE_TILE / 16) + lane, n_words = n / 16;
            KEY out[16];
            word_keys<KEY, SRC == 2>(src.packed, src.startbits, w, w < n_words, src.n_bases, src.k, out);
#pragma unroll
            for (int j = 0; j < 16; j++) stage[lane * 17 + j] = out[j];
        }
        __syncthreads();
    }
}

template <typename KEY, int SRC>
__device__ __forceinline__ unsigned long long route_key(const RouteSource &src, const KEY *stage, uint64_t base, int it, int lane, uint64_t n)
{
    if (SRC != 0) {
        const int p = it * 64 + lane;
        return (unsigned long long)stage[p + (p >> 4)];
    }
    const uint64_t i = base + it * 64 + lane;
    return i < n ? (unsigned long long)static_cast<const KEY *>(src.keys)[i] : 0ull;
}

// Ranked routing with a saturation bitmap (sat_bits != nullptr, 32-bit keys): a k-mer whose bit is set has a count of
// at least depth - 1 at its owner, so every further occurrence is "high" whatever the owner would add - it is answered
// where it is read and never sent.  The counting pass looks the bits up (one random 4-byte load per k-mer, the only such
// load of the routing) and leaves them as one 64-bit mask per round of 64 keys for the scatter pass.
template <typename KEY>
__device__ __forceinline__ bool route_stays_home(const uint32_t *__restrict__ sat_bits, KEY key)
{
    if constexpr (sizeof(KEY) == 4) return key && ((sat_bits[key >> 5] >> (key & 31u)) & 1u);
    return false;
}

template <typename KEY, int SRC>
__global__ void __launch_bounds__(256) k_route_count(RouteSource src, uint64_t n, uint32_t n_shards, uint64_t n_waves, int use_adds,
                                                     uint32_t *__restrict__ hist, const uint32_t *__restrict__ sat_bits,
                                                     unsigned long long *__restrict__ sat_mask, Counters *ctr)
{
    __shared__ KEY stage[SRC != 0 ? 4 : 1][SRC != 0 ? ROUTE_STAGE : 1];
    const uint64_t wv = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live_wave = wv < n_waves;
    route_stage_tile<KEY, SRC>(src, wv, n, live_wave, stage[SRC != 0 ? threadIdx.x >> 6 : 0]);
    if (!live_wave) return;
    int lane = threadIdx.x & 63;
    uint64_t base = wv * ROUTE_TILE;
    uint32_t mine = 0; // lane sh accumulates the count of shard sh (n_shards <= 64: one lane each)
    unsigned long long home_mask = 0; // lane `it` keeps the mask of round `it`
    static_assert(ROUTE_TILE / 64 <= 64, "one lane per round for the masks");
    // the key of (round, lane) as the routing sees it: 0 if there is none or it stays home
    auto routed_key = [&](int it) {
        unsigned long long key = route_key<KEY, SRC>(src, stage[SRC != 0 ? threadIdx.x >> 6 : 0], base, it, lane, n);
        if (sat_bits) {
            const bool home = route_stays_home<KEY>(sat_bits, (KEY)key);
            const unsigned long long m = __ballot(home);
            if (lane == it) home_mask = m;
            if (home) key = 0;
        }
        return key;
    };
    if (use_adds) {
        // many shards: one LDS add per key instead of a ballot per shard and round.  Order does not matter for a count, so
        // a shard's counter is spread over `ways` addresses picked by the lane: fewer lanes of an instruction meet at one.
        __shared__ uint32_t cnt[4][64];
        uint32_t *my = cnt[threadIdx.x >> 6];
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(64u / n_shards), way = (uint32_t)lane & ((1u << lg) - 1u);
        my[lane] = 0; // the wavefront's own row: its LDS instructions run in order
        for (int it = 0; it < ROUTE_TILE / 64; it++) {
            unsigned long long key = routed_key(it);
            if (key) atomicAdd(&my[(owner_of(key, n_shards) << lg) | way], 1u);
        }
        wave_lds_handover();
        if ((uint32_t)lane < n_shards)
            for (uint32_t j = 0; j < (1u << lg); j++) mine += __hip_atomic_load(&my[((uint32_t)lane << lg) | j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        for (int it = 0; it < ROUTE_TILE / 64; it++) {
            unsigned long long key = routed_key(it);
            uint32_t own = key ? owner_of(key, n_shards) : 0xffffffffu;
            for (uint32_t sh = 0; sh < n_shards; sh++) {
                unsigned long long m = __ballot(own == sh);
                if ((uint32_t)lane == sh) mine += (uint32_t)__popcll(m);
            }
        }
    }
    if ((uint32_t)lane < n_shards) hist[(uint64_t)lane * n_waves + wv] = mine;
    if (sat_bits) {
        if (lane < ROUTE_TILE / 64) sat_mask[wv * (ROUTE_TILE / 64) + lane] = home_mask;
        const uint32_t stayed = wave_sum((uint32_t)__popcll(home_mask));
        if (lane == 0 && stayed) atomicAdd(&ctr->route_home, stayed);
    }
}

template <typename KEY, int SRC>
__global__ void __launch_bounds__(256) k_route_scatter(RouteSource src, uint64_t n, uint32_t n_shards, uint64_t n_waves, int use_adds,
                                                       const uint32_t *__restrict__ offsets, KEY *__restrict__ send_keys,
                                                       uint32_t *__restrict__ send_src, const unsigned long long *__restrict__ sat_mask,
                                                       uint16_t *__restrict__ meta_out)
{
    // meta_out != nullptr (ranked routing, at most 32 shards): instead of every sent key's base position (send_src, 4 bytes,
    // in send order) the pass leaves 16 bits per base position, in base order: shard << 10 | rank among the tile's keys of
    // that shard, ROUTE_META_NONE where there is no k-mer, ROUTE_META_HOME where the k-mer stayed home (sat_mask).  With
    // the tile's start in every shard's segment (offsets: kept by the caller) that is where the key's flag comes back to.
    __shared__ KEY stage[SRC != 0 ? 4 : 1][SRC != 0 ? ROUTE_STAGE : 1];
    const uint64_t wv = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live_wave = wv < n_waves;
    route_stage_tile<KEY, SRC>(src, wv, n, live_wave, stage[SRC != 0 ? threadIdx.x >> 6 : 0]);
    if (!live_wave) return;
    int lane = threadIdx.x & 63;
    uint64_t base = wv * ROUTE_TILE;
    uint32_t next = (uint32_t)lane < n_shards ? offsets[(uint64_t)lane * n_waves + wv] : 0u; // lane sh: next free slot of shard sh
    const uint32_t first = next;
    unsigned long long home_word = sat_mask && lane < ROUTE_TILE / 64 ? sat_mask[wv * (ROUTE_TILE / 64) + lane] : 0ull; // lane `it`: round `it`
    auto stays_home = [&](int it) { return (bool)((__shfl(home_word, it) >> lane) & 1ull); };
    if (use_adds) {
        // many shards: the tile's keys are first brought into shard order inside LDS and leave as one run per shard, so that
        // neighbouring lanes write neighbouring addresses instead of a few keys here and a few there.  One returning add per
        // key on its shard's counter gives the key's rank among the tile's keys of that shard (the lanes of one instruction
        // that meet at an address are served in lane order, a wavefront's LDS instructions in program order, so the ranks
        // follow the tile's stream order; k_lds_order_check vouches for the device) and leaves the shard's count behind; the
        // prefix of the counts turns ranks into places.
        __shared__ uint32_t cnt[4][64], gdst[4][64];
        __shared__ uint16_t meta[4][ROUTE_TILE];                                  // element of the tile (10 bits) and shard (6 bits) per place
        __shared__ KEY obuf[SRC != 0 ? 1 : 4][SRC != 0 ? 1 : ROUTE_TILE];         // SRC != 0: the staging area is free once the keys are in registers
        const int w = threadIdx.x >> 6;
        KEY *okey = SRC != 0 ? stage[w] : obuf[SRC != 0 ? 0 : w];
        uint32_t *my = cnt[w];
        KEY kk[ROUTE_TILE / 64];
        uint32_t rk[ROUTE_TILE / 128];       // per key: rank (10 bits) and shard (6 bits), two keys to a register
#pragma unroll
        for (int it = 0; it < ROUTE_TILE / 64; it++) kk[it] = (KEY)route_key<KEY, SRC>(src, stage[SRC != 0 ? w : 0], base, it, lane, n);
        my[lane] = 0; // the wavefront's own row: its LDS instructions run in order
#pragma unroll
        for (int it = 0; it < ROUTE_TILE / 64; it++) {
            uint32_t v = 0, m16 = ROUTE_META_NONE;
            if (sat_mask && stays_home(it)) { kk[it] = 0; m16 = ROUTE_META_HOME; }
            if (kk[it]) {
                const uint32_t own = owner_of(kk[it], n_shards);
                v = atomicAdd(&my[own], 1u) | (own << 10);
                m16 = v;
            }
            rk[it >> 1] = (it & 1) ? (rk[it >> 1] | (v << 16)) : v;
            if (meta_out) meta_out[base + it * 64 + lane] = (uint16_t)m16;
        }
        wave_lds_handover();
        const uint32_t c = __hip_atomic_load(&my[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const uint32_t total = __shfl(incl, 63);
        gdst[w][lane] = next - (incl - c);   // place i of the tile's shard order goes to gdst[shard] + i
        wave_lds_handover();
        my[lane] = incl - c;                 // the shard's first place
        wave_lds_handover();
#pragma unroll
        for (int it = 0; it < ROUTE_TILE / 64; it++) {
            if (kk[it]) {
                const uint32_t v = (rk[it >> 1] >> ((it & 1) * 16)) & 0xffffu, own = v >> 10;
                const uint32_t at = my[own] + (v & 1023u);
                okey[at] = kk[it];
                meta[w][at] = (uint16_t)((uint32_t)(it * 64 + lane) | (own << 10));
            }
        }
        wave_lds_handover();
        for (uint32_t at = lane; at < total; at += 64) {
            const uint32_t m = meta[w][at], dst = gdst[w][m >> 10] + at;
            send_keys[dst] = okey[at];
            if (!meta_out) send_src[dst] = (uint32_t)(base + (m & 1023u));
        }
        return;
    }
    for (int it = 0; it < ROUTE_TILE / 64; it++) {
        uint64_t i = base + it * 64 + lane;
        unsigned long long key = route_key<KEY, SRC>(src, stage[SRC != 0 ? threadIdx.x >> 6 : 0], base, it, lane, n);
        uint32_t m16 = ROUTE_META_NONE;
        if (sat_mask && stays_home(it)) { key = 0; m16 = ROUTE_META_HOME; }
        uint32_t own = key ? owner_of(key, n_shards) : 0xffffffffu;
        uint32_t dst = 0;
        for (uint32_t sh = 0; sh < n_shards; sh++) {
            unsigned long long m = __ballot(own == sh);
            uint32_t start = __shfl(next, (int)sh), start0 = __shfl(first, (int)sh);
            if (own == sh) {
                dst = start + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                m16 = (own << 10) | (dst - start0);
            }
            if ((uint32_t)lane == sh) next += (uint32_t)__popcll(m);
        }
        if (key) {
            send_keys[dst] = (KEY)key;
            if (!meta_out) send_src[dst] = (uint32_t)i;
        }
        if (meta_out) meta_out[i] = (uint16_t)m16;
    }
}

// Ranked routing's way home: flags[e] of every base position from what the scatter pass left there.  Eight positions per
// lane: one 16-byte load, eight single-byte gathers (neighbours of one shard read neighbouring bytes), one 8-byte store.
__global__ void __launch_bounds__(256) k_unroute_ranked(const uint8_t *__restrict__ back_flags, const uint16_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ offsets, uint64_t n_waves, uint64_t n_positions,
                                                        uint8_t *__restrict__ flags)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 8;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8; e0 < n_positions; e0 += stride) { // n_positions: a multiple of 16
        const uint4 mv = *reinterpret_cast<const uint4 *>(meta + e0);
        const uint32_t mw[4] = {mv.x, mv.y, mv.z, mv.w};
        const uint64_t tile = e0 >> 10; // ROUTE_TILE = 1024 positions, eight consecutive ones share it
        uint32_t f[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t m = (mw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (m & 0x8000u) f[j] = m == ROUTE_META_HOME ? FLAG_HIGH : FLAG_NONE;
            else f[j] = back_flags[offsets[(uint64_t)(m >> 10) * n_waves + tile] + (m & 1023u)];
        }
        *reinterpret_cast<uint2 *>(flags + e0) = make_uint2(f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24), f[4] | (f[5] << 8) | (f[6] << 16) | (f[7] << 24));
    }
}

// the bits of a list of k-mers that another rank found saturated (k_sat_collect)
__global__ void __launch_bounds__(256) k_sat_apply(const uint32_t *__restrict__ codes, uint64_t n, uint32_t *__restrict__ sat_bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) atomicOr(&sat_bits[codes[i] >> 5], 1u << (codes[i] & 31u));
}

// flags that came back from the owners, in the order the keys were sent, go back to their base positions
__global__ void __launch_bounds__(256) k_unroute_flags(const uint8_t *__restrict__ recv_flags, const uint32_t *__restrict__ send_src,
                                                       uint64_t n_sent, uint8_t *__restrict__ flags)
{
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_sent; i += stride) flags[send_src[i]] = recv_flags[i];
}

// Exclusive scan of a u32 array in three small launches (block sums, scan of the sums, per-block scan + base): used
// where the library must not call anything that might allocate (nkml_route_keys_static).  n <= SCAN_BLOCK * 2048.
constexpr int SCAN_BLOCK = 1024; // entries per workgroup of 256 lanes

__global__ void __launch_bounds__(256) k_scan_block_sums(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t red[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < n) v += in[base + j];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup of 1024 lanes, two sums per lane: sums[i] becomes the total of the blocks before block i
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *__restrict__ sums, uint32_t n_blocks)
{
    __shared__ uint32_t wtot[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t i0 = threadIdx.x * 2;
    const uint32_t a = i0 < n_blocks ? sums[i0] : 0u, b = i0 + 1 < n_blocks ? sums[i0 + 1] : 0u;
    uint32_t incl = a + b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; w++) before += wtot[w];
    const uint32_t excl = before + incl - (a + b);
    if (i0 < n_blocks) sums[i0] = excl;
    if (i0 + 1 < n_blocks) sums[i0 + 1] = excl + a;
}

__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ in, uint32_t n, const uint32_t *__restrict__ sums,
                                                    uint32_t *__restrict__ out)
{
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    uint32_t v[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j] = base + j < n ? in[base + j] : 0u;
        mine += v[j];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t run = sums[blockIdx.x] + incl - mine;
    for (int w = 0; w < wv; w++) run += wtot[w];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

#include "nkml_partition.inc"
#include "nkml_hashpart.inc"
#include "nkml_verify.inc"
#include "nkml_bucketed.inc"

// table dump helpers
__global__ void __launch_bounds__(256) k_dump_direct(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ extra, uint64_t cap, unsigned long long *codes,
                                                     uint32_t *cnts, uint64_t max_entries, unsigned long long *n_out)
{
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += stride) {
        uint32_t v = counts[i];
        if (v) {
            unsigned long long at = atomicAdd(n_out, 1ull);
            if (at < max_entries) { codes[at] = i; cnts[at] = (v & COUNT_MASK) + (extra ? extra[i] : 0u); }
        }
    }
}
__global__ void __launch_bounds__(256) k_dump_hashed(const Slot *__restrict__ slots, uint64_t cap, unsigned long long *codes,
                                                     uint32_t *cnts, uint64_t max_entries, unsigned long long *n_out)
{
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += stride) {
        Slot s = slots[i];
        if (s.key) {
            unsigned long long at = atomicAdd(n_out, 1ull);
            if (at < max_entries) { codes[at] = s.key; cnts[at] = s.count + (s.state & ~ST_SAT); }
        }
    }
}

// The partition passes and the bucket kernels take stream-order ranks from returning LDS adds: lanes of one wavefront
// instruction that meet at an address are served in ascending lane order, and a wavefront's LDS instructions in program
// order (measured on gfx950 over 5e10 adds with 1 to 16 wavefronts per workgroup, tools/microbench/lds_atomic_order.hip;
// not a documented guarantee).  This kernel checks that on the device at hand before a partitioned context is created, in
// the shape the tile kernels use it: workgroups of eight wavefronts, four of them per CU on every CU (the launch asks for
// that many), a row of 256 counters per wavefront, eight returning adds per lane issued back to back with other LDS
// stores and loads of the wavefront between them, digits spread over 1 to 256 counters.  Every value that comes back
// is compared with the number of earlier elements (round, lane) of the wavefront that chose the same counter.
constexpr int LOC_WAVES = 8, LOC_ROUNDS = 8;
__global__ void __launch_bounds__(LOC_WAVES * 64) k_lds_order_check(uint32_t *bad)
{
    __shared__ uint32_t wcnt[LOC_WAVES][256];
    __shared__ uint32_t other[LOC_WAVES][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t nbad = 0, sink = 0;
    for (uint32_t spread = 1; spread <= 256; spread <<= 2) { // 1, 4, 16, 64, 256 counters per wavefront
#pragma unroll
        for (int j = 0; j < 4; j++) { wcnt[wv][j * 64 + lane] = 0; other[wv][j * 64 + lane] = lane; }
        wave_lds_handover();
        uint32_t d[LOC_ROUNDS], got[LOC_ROUNDS];
#pragma unroll
        for (int r = 0; r < LOC_ROUNDS; r++) {
            uint32_t h = (blockIdx.x * (uint32_t)LOC_WAVES + wv) * 0x9E3779B1u ^ (r * 64u + lane) * 0x85ebca6bu ^ spread;
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
            d[r] = h & (spread - 1);
        }
#pragma unroll
        for (int r = 0; r < LOC_ROUNDS; r++) {
            got[r] = atomicAdd(&wcnt[wv][d[r]], 1u);
            other[wv][(lane * 5 + r * 37) & 255] = d[r];          // unrelated stores and loads of the same wavefront in between
            sink ^= other[wv][(lane * 3 + r * 11) & 255];
        }
#pragma unroll
        for (int r = 0; r < LOC_ROUNDS; r++) {
            uint32_t earlier = 0;
#pragma unroll
            for (int r2 = 0; r2 <= r; r2++)
                for (int b = 0; b < 64; b++) {
                    const uint32_t db = __shfl(d[r2], b);
                    earlier += (db == d[r] && (r2 < r || b < lane)) ? 1u : 0u;
                }
            nbad += got[r] != earlier ? 1u : 0u;
        }
        wave_lds_handover();
    }
    asm volatile("" ::"v"(sink));
    if (nbad) atomicAdd(bad, nbad);
}

} // namespace

// ======================================================================================================
// Host side: context and C ABI
// ======================================================================================================
// Scratch of nkml_route_keys_static, the one entry point that never allocates: per-tile shard counts, their scan and
// the scan's block sums live in this fixed array instead of buffers the library allocates on demand.  Every GPU the
// library runs on holds its own copy (4 MiB + 8 KiB) until the process ends; the host reaches it only through the
// per-device address taken once with hipGetSymbolAddress (the array's name is not a device address in host code).
#define NKML_ROUTE_STATIC_ENTRIES NKML_ROUTE_STATIC_MAX_ENTRIES /* (tiles of 1024 keys) x shards, e.g. 128 Mi keys for 4 shards */
static __device__ __align__(256) uint32_t scratch[2 * (NKML_ROUTE_STATIC_ENTRIES + 64) + 2048];
constexpr int MAX_DEVICES = 64;
static uint32_t *g_scratch_ptr[MAX_DEVICES]; // device address of `scratch` on each device, null until first asked for
static int g_lds_order[MAX_DEVICES];         // k_lds_order_check on each device: 0 not run yet, 1 passed, -1 failed
static std::mutex g_device_mutex;            // guards the two arrays above (contexts may be created from several threads)

struct nkml_ctx {
    nkml_config cfg;
    int key_bytes;          // 4 for k <= 16, else 8
    bool direct;
    hipStream_t stream;
    bool own_stream;
    bool lds_ordered;       // this device passed k_lds_order_check (ranks may come from returning LDS adds)
    bool broken;            // a batch failed after part of it had been merged into the table: unusable until nkml_reset
    bool verify;            // NKML_F_VERIFY_RANKS: every partitioned batch is checked by the kernels of nkml_verify.inc
    void *vf_before;        // ... copy of the table as it was when the batch started
    uint64_t vf_before_bytes;
    unsigned long long *vf_keys; // ... hashed table: copy of the second pass's output (the fast bucket kernel queues keys in it)
    int vf_fault;           // test aid (NKML_TEST_VERIFY_FAULT): damage every batch on purpose
    char err[512];

    // table
    uint32_t *counts;       // direct
    uint32_t *sat;          // direct: saturation bitmap
    uint32_t *extra;        // direct, NKML_F_EXACT_COUNTS: occurrences counted after saturation
    uint32_t *filter;       // crossing filter (both table kinds)
    // partitioned counting (k <= 15): one or two bytes per k-mer and the buffers of the two radix passes
    bool partitioned;
    bool pt_wide;           // two-byte table entries (depth > 127)
    void *pt_table;
    uint32_t *pt_keysA, *pt_scanA, *pt_scanB, *pt_pad, *pt_bounds;
    pt_hist_t *pt_histA, *pt_histB;
    uint4 *pt_tally;                             // per bucket: new k-mers, keys counted, keys of saturated k-mers
    uint32_t *pt_chunks, *pt_totals, *pt_startB; // column-scan scratch, keys per digit, group starts of the second pass
    uint16_t *pt_keysB;
    uint32_t *pt_sposA, *pt_sposB; // sorted positions, 12 bits each: what the forward passes leave for the reverse passes
    uint8_t *pt_flagsA, *pt_flagsB;
    uint64_t pt_tiles_a, pt_tiles_b; // strides of the histograms (tiles of the largest batch)
    // the same for the hashed table (k > 16): 8-byte slots grouped in 2^16 buckets, 64-bit hashes through the passes
    bool hashpart;
    unsigned long long *hp_table, *hp_keysA, *hp_keysB;
    uint32_t *hp_done;       // per bucket workgroup: its part of the current batch is committed
    int hp_log2_bucket;      // slots per bucket = 2^hp_log2_bucket; capacity = 2^(16 + hp_log2_bucket)
    int hp_log2_local_max;   // slots one bucket workgroup holds in LDS at most (2^this)
    uint64_t hp_leave_slots; // the table moves to the atomic path once it has more slots than this (0: never)
    uint64_t hp_unique_before; // stored k-mers when the last batch began (what a batch adds is the best guess for the next one)
    uint64_t pt_timed_keys;
    bool pt_timing_pending;
    double pt_stage_ms;
    Slot *slots;            // hashed
    Slot *spare_slots;      // hashed: the next table, allocated ahead of the batch that will need it (or null)
    uint64_t spare_capacity;
    uint64_t used_bound;    // hashed: upper bound on the stored keys (last fetched count + positions enqueued since)
    uint64_t capacity;      // slots / counters
    uint64_t resizes;

    // per-batch buffers
    uint64_t max_words;     // ceil(max_batch_bases / 16)
    uint8_t *d_seq;
    uint64_t *d_offsets;
    uint32_t *d_packed, *d_startbits;
    uint32_t n_shards = 1, shard = 0; // replicated-reads mode: this context counts only the k-mers of its shard
    // ... which it first pulls together into a key array (set up by nkml_set_shard; nullptr: the passes blank the others' k-mers)
    uint32_t *sh_keys, *sh_tile_count, *sh_tile_base;
    uint16_t *sh_own;       // per packed word: which of its 16 windows hold a k-mer of this shard
    uint8_t *sh_flags;
    void *sh_scan_tmp;
    size_t sh_scan_tmp_bytes;
    uint32_t *d_wflags;     // per packed word: HIGH bits (low half) and LOW bits (high half) of its 16 windows
    int32_t *d_high, *d_total;

    // per-chunk buffers
    uint32_t chunk;         // keys per counting pass
    int posbits;
    void *d_list_sat, *d_list_cross;
    unsigned long long *d_occ, *d_occ_sorted;
    void *d_sort_tmp;
    size_t sort_tmp_bytes;

    // routing scratch (sharded path), allocated on first use
    uint32_t *d_route_hist, *d_route_off;
    uint32_t *sat_bits;             // nkml_sat_enable: one bit per k-mer code, set = known to have reached depth - 1 at its owner
    unsigned long long *d_sat_mask; // ... per routing tile and round: which of the 64 positions stayed home
    uint64_t sat_mask_words;
    uint64_t route_home_total;      // positions answered by the bitmap since create / reset
    // bucket-ordered exchange (nkml_xb_enable): the deal of buckets to ranks, the sender's and the owner's buffers
    uint32_t xb_world, xb_rank, xb_buckets;
    uint32_t *xb_order, *xb_place;  // bucket at place i / place of bucket b
    uint32_t *xb_kept, *xb_base;    // sender: travelling keys per place, their exclusive scan (xb_buckets + 1 entries)
    uint32_t *xb_sizes, *xb_bounds2, *xb_run_at; // owner: keys per bucket, their scan; where every received run starts
    uint16_t *xb_keys2;             // owner: what arrived, in bucket order
    uint8_t *xb_flags2;
    uint64_t xb_capacity;           // keys the owner's buffers hold
    void *xb_scan_tmp;
    size_t xb_scan_tmp_bytes;
    uint64_t xb_n;                  // positions of the batch between nkml_xb_front and nkml_xb_finish (0 = none under way)
    uint64_t route_entries;
    void *d_route_tmp;
    size_t route_tmp_bytes;
    uint32_t *h_route_starts; // pinned: shard start offsets of the last routing call (65 entries)

    Counters *d_ctr;
    Counters *h_ctr;        // pinned
    // pinned staging for the host path
    uint8_t *h_seq;
    uint64_t *h_offsets;
    int32_t *h_high, *h_total;
    uint32_t pending_reads; // reads of the submitted, not yet collected batch
    bool pending;
    bool pending_check; // the submitted batch's counters / bad-base tally have not been looked at yet
    hipEvent_t ev_staged; // after the host-to-device copies of the last staged batch
    bool staged_pending;

    // stats
    uint64_t crossing_kmers, crossing_occ;
    hipEvent_t ev[2 * 64];  // start/stop pairs around the counting kernel of up to 64 chunks per batch
    double count_ms;
    uint64_t count_launches, count_keys;
};

namespace {

int fail(nkml_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

// A batch that fails after some of its k-mers have been merged into the table leaves counts that belong to no prefix of
// the stream: counting on would double-count them.  Such a context answers NKML_E_STATE until nkml_reset.
int broken_context(nkml_ctx *c)
{
    return fail(c, NKML_E_STATE, "an earlier batch failed half-way and left the table partly updated: nkml_reset before counting again");
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, e_ == hipErrorOutOfMemory ? NKML_E_NOMEM : NKML_E_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)

inline uint32_t blocks_for(uint64_t n, uint32_t per_block = 256) { return (uint32_t)((n + per_block - 1) / per_block); }
constexpr uint32_t GRID_PERSIST = 256 * 8; // 8 workgroups of 256 lanes per CU, grid-stride beyond
constexpr uint32_t GRID_FUSED = 256 * 8;

uint64_t default_hashed_capacity(int memory_gb)
{
    // the reference sizes its table in 16-byte slots (normalise_kmers_multi_large.c:411-422); default 1 GiB (:137).
    uint64_t gb = memory_gb > 0 ? (uint64_t)memory_gb : 1;
    uint64_t slots = gb << 26; // GiB / 16 B
    uint64_t cap = 1;
    while (cap < slots) cap <<= 1;
    return cap;
}

constexpr int HP_MAX_LOG2_BUCKET = 16;                                        // 2^32 slots = 32 GiB of table
constexpr uint32_t HP_MAX_GRID = 65536u << HB_MAX_SUBS_LOG2;                  // sub-ranges of all buckets at most
constexpr int HP_DEFAULT_LOG2_LOCAL = 12;                                     // 4096 slots x 12 bytes of LDS per bucket workgroup: three workgroups of 512 lanes per CU (measured: 12 > 13 > 11)

// sub-range geometry of a table with 2^log2_bucket slots per bucket
HpGeom hp_geom(const nkml_ctx *c, int log2_bucket);

// bytes of the partitioned path's table: one entry per k-mer, at least one bucket
uint64_t pt_table_bytes(const nkml_ctx *c)
{
    const uint64_t entries = c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity;
    return entries * (c->pt_wide ? 2 : 1);
}

int alloc_table(nkml_ctx *c)
{
    if (c->partitioned) {
        const uint64_t npos = c->max_words * 16;
        if (c->hashpart) {
            c->capacity = default_hashed_capacity(c->cfg.memory_gb);
            c->hp_log2_bucket = 0;
            while ((65536ull << c->hp_log2_bucket) < c->capacity) c->hp_log2_bucket++;
            c->hp_log2_local_max = HP_DEFAULT_LOG2_LOCAL;
            if (const char *e = getenv("NKML_HP_LOG2_LOCAL")) { // tuning aid
                const int v = atoi(e);
                if (v >= 3 && v <= HB_MAX_SLOTS_LOG2) c->hp_log2_local_max = v; // at least one set of slots
            }
            if (const char *e = getenv("NKML_HP_LOG2_BUCKET")) { // test aid: a tiny table, so that small inputs make it grow
                const int v = atoi(e);
                if (v >= 3 && v <= HP_MAX_LOG2_BUCKET) { c->hp_log2_bucket = v; c->capacity = 65536ull << v; } // at least one set of slots
            }
            HIPCHK(c, hipMalloc(&c->hp_table, c->capacity * sizeof(unsigned long long)));
            HIPCHK(c, hipMalloc(&c->hp_done, HP_MAX_GRID * sizeof(uint32_t)));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hp_count_bucket), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(HB_MAX_SLOTS * 12)));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hp_count_bucket_fast<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(HB_MAX_SLOTS * 12)));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hp_count_bucket_fast<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(HB_MAX_SLOTS * 12)));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hp_grow), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HB_MAX_SLOTS * 8)));
        } else {
            c->capacity = 1ull << (2 * c->cfg.ksize);
            if (!(c->cfg.flags & NKML_F_NO_TABLE)) HIPCHK(c, hipMalloc(&c->pt_table, pt_table_bytes(c)));
        }
        c->pt_tiles_a = (npos + PT_TILE - 1) / PT_TILE;
        c->pt_tiles_b = c->pt_tiles_a + 256 + 1;
        const uint64_t na = (c->pt_tiles_b + 1) * PT_TILE;
        if (c->hashpart) {
            HIPCHK(c, hipMalloc(&c->hp_keysA, na * sizeof(unsigned long long)));
            HIPCHK(c, hipMalloc(&c->hp_keysB, npos * sizeof(unsigned long long)));
        } else {
            HIPCHK(c, hipMalloc(&c->pt_keysA, na * sizeof(uint32_t)));
            HIPCHK(c, hipMalloc(&c->pt_keysB, npos * sizeof(uint16_t)));
        }
        HIPCHK(c, hipMalloc(&c->pt_flagsA, na));
        HIPCHK(c, hipMalloc(&c->pt_flagsB, npos));
        HIPCHK(c, hipMalloc(&c->pt_sposA, c->pt_tiles_a * PT_TILE / 8 * 3 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_sposB, na / 8 * 3 * sizeof(uint32_t)));
        // histograms and their scans: one row of 256 per tile plus the all-zero row after the last tile
        HIPCHK(c, hipMalloc(&c->pt_histA, 256 * (c->pt_tiles_a + 1) * sizeof(pt_hist_t)));
        HIPCHK(c, hipMalloc(&c->pt_scanA, 256 * (c->pt_tiles_a + 1) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_histB, 256 * (c->pt_tiles_b + 1) * sizeof(pt_hist_t)));
        HIPCHK(c, hipMalloc(&c->pt_scanB, 256 * (c->pt_tiles_b + 1) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_pad, 260 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_startB, 260 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_bounds, ((c->hashpart ? (uint64_t)65536 : (c->capacity >> PT_LOWBITS)) + 2) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_tally, (c->hashpart ? (uint64_t)HP_MAX_GRID : (c->capacity >> PT_LOWBITS) + 1) * sizeof(uint4)));
        HIPCHK(c, hipMalloc(&c->pt_totals, 256 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->pt_chunks, 256 * ((c->pt_tiles_b + 1 + PT_CH - 1) / PT_CH + 1) * sizeof(uint32_t)));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pt_count_bucket_big<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pt_count_bucket_big<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        return NKML_OK;
    }
    if (c->direct) {
        c->capacity = 1ull << (2 * c->cfg.ksize);
        HIPCHK(c, hipMalloc(&c->counts, c->capacity * sizeof(uint32_t)));
        uint64_t bm = c->capacity / 8 < 256 ? 256 : c->capacity / 8;
        HIPCHK(c, hipMalloc(&c->sat, bm));
        if (c->cfg.flags & NKML_F_EXACT_COUNTS) HIPCHK(c, hipMalloc(&c->extra, c->capacity * sizeof(uint32_t)));
    } else {
        c->capacity = default_hashed_capacity(c->cfg.memory_gb);
        HIPCHK(c, hipMalloc(&c->slots, c->capacity * sizeof(Slot)));
    }
    HIPCHK(c, hipMalloc(&c->filter, FILTER_WORDS * sizeof(uint32_t)));
    return NKML_OK;
}

// lists and sort buffers of the atomic path's order fix-up
int alloc_atomic_lists(nkml_ctx *c)
{
    const uint64_t npos = c->max_words * 16;
    const uint64_t per_chunk = npos < c->chunk ? npos : c->chunk;
    HIPCHK(c, hipMalloc(&c->d_list_sat, per_chunk * c->key_bytes));
    HIPCHK(c, hipMalloc(&c->d_list_cross, per_chunk * c->key_bytes));
    HIPCHK(c, hipMalloc(&c->d_occ, npos * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_occ_sorted, npos * sizeof(unsigned long long)));
    size_t tmp = 0;
    HIPCHK(c, rocprim::radix_sort_keys(nullptr, tmp, c->d_occ, c->d_occ_sorted, (size_t)npos, 0u, 64u, c->stream));
    c->sort_tmp_bytes = tmp ? tmp : 256;
    HIPCHK(c, hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes));
    return NKML_OK;
}

// hipMemsetAsync in pieces of 1 GiB (tables here reach 8 GiB and more)
hipError_t zero_device(void *p, uint64_t bytes, hipStream_t stream)
{
    const uint64_t piece = 1ull << 30;
    for (uint64_t at = 0; at < bytes; at += piece) {
        hipError_t e = hipMemsetAsync(static_cast<char *>(p) + at, 0, bytes - at < piece ? bytes - at : piece, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int clear_table(nkml_ctx *c)
{
    if (c->hashpart) {
        HIPCHK(c, zero_device(c->hp_table, c->capacity * sizeof(unsigned long long), c->stream));
    } else if (c->partitioned) {
        if (c->pt_table) HIPCHK(c, zero_device(c->pt_table, pt_table_bytes(c), c->stream));
    } else if (c->direct) {
        uint64_t bm = c->capacity / 8 < 256 ? 256 : c->capacity / 8;
        HIPCHK(c, zero_device(c->counts, c->capacity * sizeof(uint32_t), c->stream));
        HIPCHK(c, zero_device(c->sat, bm, c->stream));
        if (c->extra) HIPCHK(c, zero_device(c->extra, c->capacity * sizeof(uint32_t), c->stream));
    } else {
        HIPCHK(c, zero_device(c->slots, c->capacity * sizeof(Slot), c->stream));
    }
    if (c->filter) HIPCHK(c, hipMemsetAsync(c->filter, 0, FILTER_WORDS * sizeof(uint32_t), c->stream));
    if (c->sat_bits) HIPCHK(c, zero_device(c->sat_bits, (c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity) / 8, c->stream));
    c->route_home_total = 0;
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream));
    c->crossing_kmers = c->crossing_occ = 0;
    c->count_ms = 0;
    c->pt_stage_ms = 0;
    c->count_launches = c->count_keys = 0;
    return NKML_OK;
}

int fetch_counters(nkml_ctx *c)
{
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NKML_OK;
}

// K0 + K1 on the context's stream: read starts and 2-bit packing of one batch.
int run_prepare(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases)
{
    uint64_t n_words = (n_bases + 15) / 16;
    hipLaunchKernelGGL(k_pack, blocks_for(n_words + 4), 256, 0, c->stream, d_seq, n_bases, c->d_packed, n_words, c->d_ctr, c->d_startbits);
    hipLaunchKernelGGL(k_mark_starts, blocks_for((uint64_t)n_reads + 1), 256, 0, c->stream, d_offsets, n_reads, c->d_startbits);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// K0-K2: keys materialised per base position (the staged form used by the sharded path).
int run_kmerise(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, void *d_keys)
{
    int rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    const uint64_t n_words = (n_bases + 15) / 16;
    const int k = c->cfg.ksize;
    const uint32_t g = blocks_for(n_words);
    if (c->key_bytes == 4) {
        if (c->cfg.canonical)
            hipLaunchKernelGGL((k_kmerise<uint32_t, true>), g, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, (uint32_t *)d_keys);
        else
            hipLaunchKernelGGL((k_kmerise<uint32_t, false>), g, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, (uint32_t *)d_keys);
    } else {
        if (c->cfg.canonical)
            hipLaunchKernelGGL((k_kmerise<unsigned long long, true>), g, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, (unsigned long long *)d_keys);
        else
            hipLaunchKernelGGL((k_kmerise<unsigned long long, false>), g, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, (unsigned long long *)d_keys);
    }
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int rank_collected(nkml_ctx *c, uint8_t *d_flags, uint32_t *d_wflags);
void reserve_next_table(nkml_ctx *c);

// Growth of the hashed table (expand_local_hash_table, normalise_kmers_multi_large.c:1055-1108; load factor 0.8, :143, 933).
// The table must have room for every key a counting pass can add before the pass starts, and a pass of `incoming`
// positions can add that many.  What is stored is only known on the device, so the host keeps an upper bound
// (the last fetched count plus every position enqueued since) and fetches the true count - one short wait - only when
// the bound says the next pass might not fit.  The table then doubles until it does.  The next table is allocated
// ahead of time where possible (reserve_next_table, called between batches), so that a growth inside a batch finds
// its memory waiting and only has to rehash.  While old and new table coexist the path holds 3x the old table's bytes.
// d_flags / d_wflags: where the batch's flags go, if a batch is being counted (see rank_collected).
int ensure_capacity(nkml_ctx *c, uint64_t incoming, uint8_t *d_flags, uint32_t *d_wflags)
{
    if (c->direct) return NKML_OK;
    if ((double)(c->used_bound + incoming) < (double)c->capacity * 0.8) { c->used_bound += incoming; return NKML_OK; }
    int rc = fetch_counters(c); // waits for everything enqueued so far
    if (rc != NKML_OK) return rc;
    uint64_t used = c->h_ctr->unique;
    if ((double)(used + incoming) >= (double)c->capacity * 0.8 && (d_flags || d_wflags)) {
        // the crossing occurrences collected by the batch's earlier passes name their k-mers by slot: rank them before the
        // slots move
        if ((rc = rank_collected(c, d_flags, d_wflags)) != NKML_OK) return rc;
    }
    while ((double)(used + incoming) >= (double)c->capacity * 0.8) {
        uint64_t ncap = c->capacity * 2;
        Slot *ns = nullptr;
        if (c->spare_slots && c->spare_capacity == ncap) {
            ns = c->spare_slots;
            c->spare_slots = nullptr;
            c->spare_capacity = 0;
        } else {
            if (c->spare_slots) { hipFree(c->spare_slots); c->spare_slots = nullptr; c->spare_capacity = 0; }
            HIPCHK(c, hipMalloc(&ns, ncap * sizeof(Slot)));
        }
        HIPCHK(c, zero_device(ns, ncap * sizeof(Slot), c->stream));
        hipLaunchKernelGGL(k_rehash, GRID_PERSIST, 256, 0, c->stream, c->slots, c->capacity, ns, ncap - 1, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->slots));
        c->slots = ns;
        c->capacity = ncap;
        c->resizes++;
    }
    c->used_bound = used + incoming;
    return NKML_OK;
}

// Between batches (the counters have just been fetched): once the table is 40 % full the next doubling is near, so
// its memory is taken now, outside the counting passes.  Failing to get it is not an error here; ensure_capacity asks
// again when it needs it.
void reserve_next_table(nkml_ctx *c)
{
    if (c->direct || c->partitioned || c->spare_slots) return;
    c->used_bound = c->h_ctr->unique;
    if ((double)c->h_ctr->unique < (double)c->capacity * 0.4) return;
    Slot *ns = nullptr;
    if (hipMalloc(&ns, c->capacity * 2 * sizeof(Slot)) != hipSuccess) { (void)hipGetLastError(); return; }
    c->spare_slots = ns;
    c->spare_capacity = c->capacity * 2;
}

// Sort + rank of the crossing occurrences collected so far (the counters have just been fetched): exact stream-order
// ranks for them, HIGH where the rank reaches depth.  Exactly one of d_flags (byte flags per position) / d_wflags (word
// flags) is given.  A k-mer crosses in one counting pass only and is saturated - its count frozen - from the next pass on,
// so this may run after any pass that follows the crossing, not only at the end of the batch.
int rank_collected(nkml_ctx *c, uint8_t *d_flags, uint32_t *d_wflags)
{
    const uint32_t depth = (uint32_t)c->cfg.depth;
    uint32_t n_occ = c->h_ctr->n_occ;
    if (!n_occ) return NKML_OK;
    int idbits = 0;
    while (idbits < 64 && (c->capacity - 1) >> idbits) idbits++;
    if (c->posbits + idbits > 64) return fail(c, NKML_E_ARG, "table and batch too large for 64-bit (k-mer, position) composites");
    size_t tmp = c->sort_tmp_bytes;
    HIPCHK(c, rocprim::radix_sort_keys(c->d_sort_tmp, tmp, c->d_occ, c->d_occ_sorted, (size_t)n_occ, 0u, (unsigned)(c->posbits + idbits),
                                       c->stream));
    const uint32_t g = blocks_for(n_occ);
    if (d_flags) {
        if (c->direct)
            hipLaunchKernelGGL((k_rank<DirectTable>), g, 256, 0, c->stream, c->d_occ_sorted, n_occ, c->posbits, DirectTable{c->counts}, depth, d_flags);
        else
            hipLaunchKernelGGL((k_rank<HashedTable>), g, 256, 0, c->stream, c->d_occ_sorted, n_occ, c->posbits, HashedTable{c->slots}, depth, d_flags);
    } else {
        if (c->direct)
            hipLaunchKernelGGL((k_rank_words<DirectTable>), g, 256, 0, c->stream, c->d_occ_sorted, n_occ, c->posbits, DirectTable{c->counts}, depth, d_wflags);
        else
            hipLaunchKernelGGL((k_rank_words<HashedTable>), g, 256, 0, c->stream, c->d_occ_sorted, n_occ, c->posbits, HashedTable{c->slots}, depth, d_wflags);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(&c->d_ctr->n_occ, 0, sizeof(unsigned int), c->stream));
    c->h_ctr->n_occ = 0;
    c->crossing_occ += n_occ;
    return NKML_OK;
}

// End of a batch's counting passes: the one host wait (for the number of crossing occurrences), then sort + rank.
int finish_counting(nkml_ctx *c, int n_ev, uint64_t keys_timed, uint8_t *d_flags, uint32_t *d_wflags)
{
    int rc;
    if ((rc = fetch_counters(c)) != NKML_OK) return rc;
    if (c->h_ctr->probe_overflow) return fail(c, NKML_E_STATE, "hashed table overflow (capacity %llu)", (unsigned long long)c->capacity);
    for (int i = 0; i < n_ev; i++) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
        c->count_ms += ms;
    }
    c->count_launches += n_ev;
    c->count_keys += keys_timed;
    if (d_flags || d_wflags)
        if ((rc = rank_collected(c, d_flags, d_wflags)) != NKML_OK) return rc;
    reserve_next_table(c);
    return NKML_OK;
}

// after every chunk's counting pass: newly saturated k-mers become visible, filter and list lengths restart
int end_chunk(nkml_ctx *c)
{
    if (c->direct)
        hipLaunchKernelGGL(k_apply_direct, 256, 256, 0, c->stream, (const uint32_t *)c->d_list_sat, (const uint32_t *)c->d_list_cross, c->sat, c->d_ctr);
    else
        hipLaunchKernelGGL(k_apply_hashed, 256, 256, 0, c->stream, (const unsigned long long *)c->d_list_sat,
                           (const unsigned long long *)c->d_list_cross, c->slots, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(c->filter, 0, FILTER_WORDS * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_end_chunk, 1, 1, 0, c->stream, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// Staged K3 + K4 over a key array [0, n) in chunks; n is a multiple of 16.  Per chunk: count, collect the occurrences
// of the k-mers that crossed, make the newly saturated k-mers visible to the next chunk - all enqueued without
// waiting.
int run_count(nkml_ctx *c, const void *d_keys, uint64_t n, uint8_t *d_flags)
{
    if (c->broken) return broken_context(c);
    const uint32_t depth = (uint32_t)c->cfg.depth;
    const bool profile = (c->cfg.flags & NKML_F_PROFILE) != 0;
    const int exact = (c->cfg.flags & NKML_F_EXACT_COUNTS) ? 1 : 0;
    int rc;
    if (n > (1ull << c->posbits)) return fail(c, NKML_E_ARG, "%llu keys exceed the context's batch size", (unsigned long long)n);
    int n_ev = 0;
    uint64_t keys_timed = 0;
    for (uint64_t a = 0; a < n; a += c->chunk) {
        uint64_t len = n - a < c->chunk ? n - a : c->chunk;
        if ((rc = ensure_capacity(c, len, d_flags, nullptr)) != NKML_OK) return rc; // room for every key this pass can add
        const bool timed = profile && n_ev < 64;
        if (timed) HIPCHK(c, hipEventRecord(c->ev[2 * n_ev], c->stream));
        if (c->direct)
            hipLaunchKernelGGL(k_count_direct, GRID_PERSIST, 256, 0, c->stream, (const uint32_t *)d_keys + a, len / 4, d_flags + a, c->counts,
                               c->sat, c->filter, depth, c->extra, c->d_ctr, (uint32_t *)c->d_list_sat, (uint32_t *)c->d_list_cross);
        else
            hipLaunchKernelGGL(k_count_hashed, GRID_PERSIST, 256, 0, c->stream, (const unsigned long long *)d_keys + a, len, d_flags + a, c->slots,
                               c->capacity - 1, c->filter, depth, exact, c->d_ctr, (unsigned long long *)c->d_list_sat,
                               (unsigned long long *)c->d_list_cross);
        HIPCHK(c, hipGetLastError());
        if (timed) {
            HIPCHK(c, hipEventRecord(c->ev[2 * n_ev + 1], c->stream));
            n_ev++;
            keys_timed += len;
        }
        if (c->direct)
            hipLaunchKernelGGL(k_collect_direct, GRID_PERSIST, 256, 0, c->stream, (const uint32_t *)d_keys + a, d_flags + a, len / 4, a, c->filter,
                               c->counts, depth, c->posbits, c->d_occ, c->d_ctr);
        else
            hipLaunchKernelGGL(k_collect_hashed, GRID_PERSIST, 256, 0, c->stream, (const unsigned long long *)d_keys + a, d_flags + a, len, a,
                               c->filter, c->slots, c->capacity - 1, depth, c->posbits, c->d_occ, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        if ((rc = end_chunk(c)) != NKML_OK) return rc;
    }
    return finish_counting(c, n_ev, keys_timed, d_flags, nullptr);
}

// Fused K2+K3+K4 over the prepared batch (packed words), the single-GPU path: keys live in registers only.
int run_fused(nkml_ctx *c, uint64_t n_bases, bool seeding)
{
    if (c->broken) return broken_context(c);
    const uint32_t depth = (uint32_t)c->cfg.depth;
    const bool profile = (c->cfg.flags & NKML_F_PROFILE) != 0 && !seeding;
    const int exact = (c->cfg.flags & NKML_F_EXACT_COUNTS) ? 1 : 0;
    const int k = c->cfg.ksize;
    const bool canon = c->cfg.canonical != 0;
    const uint64_t n_words = (n_bases + 15) / 16;
    int rc;
    if (seeding) {
        if ((rc = ensure_capacity(c, n_words * 16, nullptr, nullptr)) != NKML_OK) return rc;
        const uint64_t mask = c->capacity - 1; // after the table has grown, not before
#define SEED_LAUNCH(CANON, DIRECT) hipLaunchKernelGGL((k_seed_words<CANON, DIRECT>), GRID_PERSIST, 256, 0, c->stream, c->d_packed, c->d_startbits, \
                                                      n_bases, n_words, k, c->counts, c->slots, mask, c->d_ctr)
        if (c->direct) { if (canon) SEED_LAUNCH(true, true); else SEED_LAUNCH(false, true); }
        else { if (canon) SEED_LAUNCH(true, false); else SEED_LAUNCH(false, false); }
#undef SEED_LAUNCH
        HIPCHK(c, hipGetLastError());
        return finish_counting(c, 0, 0, nullptr, c->d_wflags);
    }
    int n_ev = 0;
    uint64_t keys_timed = 0;
    const uint64_t chunk_words = c->chunk / 16;
    for (uint64_t wa = 0; wa < n_words; wa += chunk_words) {
        const uint64_t wb = wa + chunk_words < n_words ? wa + chunk_words : n_words;
        if ((rc = ensure_capacity(c, (wb - wa) * 16, nullptr, c->d_wflags)) != NKML_OK) return rc; // room for every key this pass can add
        const uint64_t mask = c->capacity - 1; // after the table has grown, not before
        const bool timed = profile && n_ev < 64;
        if (timed) HIPCHK(c, hipEventRecord(c->ev[2 * n_ev], c->stream));
        if (c->direct) {
#define COUNT_LAUNCH(CANON) hipLaunchKernelGGL((k_count_words_direct<CANON>), GRID_FUSED, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, wa, wb, k, \
                                               c->d_wflags, c->counts, c->sat, c->filter, depth, c->extra, c->d_ctr, (uint32_t *)c->d_list_sat,          \
                                               (uint32_t *)c->d_list_cross)
            if (canon) COUNT_LAUNCH(true); else COUNT_LAUNCH(false);
#undef COUNT_LAUNCH
        } else {
#define COUNT_LAUNCH(CANON) hipLaunchKernelGGL((k_count_words_hashed<CANON>), GRID_FUSED, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, wa, wb, k, \
                                               c->d_wflags, c->slots, mask, c->filter, depth, exact, c->d_ctr, (unsigned long long *)c->d_list_sat,      \
                                               (unsigned long long *)c->d_list_cross)
            if (canon) COUNT_LAUNCH(true); else COUNT_LAUNCH(false);
#undef COUNT_LAUNCH
        }
        HIPCHK(c, hipGetLastError());
        if (timed) {
            HIPCHK(c, hipEventRecord(c->ev[2 * n_ev + 1], c->stream));
            n_ev++;
            keys_timed += (wb - wa) * 16;
        }
#define COLLECT_LAUNCH(CANON, DIRECT) hipLaunchKernelGGL((k_collect_words<CANON, DIRECT>), GRID_FUSED, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, \
                                                         wa, wb, k, c->d_wflags, c->filter, c->counts, c->slots, mask, depth, c->posbits, c->d_occ, c->d_ctr)
        if (c->direct) { if (canon) COLLECT_LAUNCH(true, true); else COLLECT_LAUNCH(false, true); }
        else { if (canon) COLLECT_LAUNCH(true, false); else COLLECT_LAUNCH(false, false); }
#undef COLLECT_LAUNCH
        HIPCHK(c, hipGetLastError());
        if ((rc = end_chunk(c)) != NKML_OK) return rc;
    }
    return finish_counting(c, n_ev, keys_timed, nullptr, c->d_wflags);
}

int run_reduce(nkml_ctx *c, const uint8_t *d_flags, const uint64_t *d_offsets, uint32_t n_reads, int32_t *d_high, int32_t *d_total)
{
    if (n_reads == 0) return NKML_OK;
    hipLaunchKernelGGL(k_reduce_reads, blocks_for(n_reads, 16), 256, 0, c->stream, d_flags, d_offsets, n_reads, d_high, d_total);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// Partitioned counting of a key array (k <= 15): two stable radix passes, one workgroup per bucket, two reverse
// passes for the flags.  Nothing here waits for the device.
int collect_partition_timing(nkml_ctx *c);
int run_count_shard(nkml_ctx *c, uint64_t n_words, uint64_t n_bases);

// d_keys == nullptr: the keys of the first pass are recomputed from the context's packed words (n_bases of them
// prepared); d_flags == nullptr: the result is the context's word flags (one word per 16 positions).
int run_count_hashpart(nkml_ctx *c, const unsigned long long *d_keys, uint64_t n, uint64_t n_bases, uint8_t *d_flags, bool seeding);
int collect_partition_timing(nkml_ctx *c);

// NKML_F_VERIFY_RANKS: the table as it is now (the batch has not touched it yet) goes to a buffer of its own
int verify_snapshot(nkml_ctx *c, const void *table, uint64_t bytes)
{
    if (c->vf_before_bytes < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->vf_before);
        c->vf_before = nullptr;
        c->vf_before_bytes = 0;
        HIPCHK(c, hipMalloc(&c->vf_before, bytes));
        c->vf_before_bytes = bytes;
        const size_t vf_lds = 3 * VF_WINDOW * sizeof(uint32_t);
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_verify_bucket<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vf_lds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_verify_bucket<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vf_lds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_verify_hbucket), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vf_lds));
    }
    HIPCHK(c, hipMemcpyAsync(c->vf_before, table, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(&c->d_ctr->verify_bad, 0, sizeof(unsigned int), c->stream));
    return NKML_OK;
}

// ... and after the batch's last kernel: wait, read the tally, fail the batch if anything was out of place
int verify_outcome(nkml_ctx *c)
{
    int rc = fetch_counters(c);
    if (rc != NKML_OK) return rc;
    if (c->h_ctr->verify_bad) {
        c->broken = true; // the batch is in the table, its flags cannot be trusted
        return fail(c, NKML_E_STATE, "rank verification: %u elements of the batch are not where a stable partition puts them or carry the wrong flag",
                    c->h_ctr->verify_bad);
    }
    return NKML_OK;
}

// the check of one partition pass (see k_verify_pass); src.keys == nullptr: the keys come from the context's packed words
template <typename KEY, typename KOUT>
void verify_pass(nkml_ctx *c, VerifySource src, uint32_t tiles, int shift, const uint32_t *scanned, const KOUT *out, unsigned long long out_mask)
{
    if (src.keys)
        hipLaunchKernelGGL((k_verify_pass<KEY, KOUT, 0>), tiles, 256, 0, c->stream, src, shift, scanned, out, out_mask, c->d_ctr);
    else if (c->cfg.canonical)
        hipLaunchKernelGGL((k_verify_pass<KEY, KOUT, 2>), tiles, 256, 0, c->stream, src, shift, scanned, out, out_mask, c->d_ctr);
    else
        hipLaunchKernelGGL((k_verify_pass<KEY, KOUT, 1>), tiles, 256, 0, c->stream, src, shift, scanned, out, out_mask, c->d_ctr);
}

// The two partition passes of the direct table's path: the batch's k-mers (from d_keys, or recomputed from the context's
// packed words) end up in pt_keysB grouped by bucket, each bucket in stream order; what the reverse passes need stays in
// pt_sposA / pt_sposB (not when seeding) and the histograms.
static int pt_forward(nkml_ctx *c, const uint32_t *d_keys, uint64_t n, uint64_t n_bases, bool seeding)
{
    const uint64_t ta = (n + PT_TILE - 1) / PT_TILE;       // first-pass tiles of this batch
    const uint64_t tb = ta + 256 + 1;                      // bound on second-pass tiles (each group padded to a tile)
    const uint64_t ra = ta + 1, rb = tb + 1;               // histogram rows: the tiles and the all-zero row after them
    const uint32_t ca = (uint32_t)((ra + PT_CH - 1) / PT_CH), cb = (uint32_t)((rb + PT_CH - 1) / PT_CH);
    const int shift_a = PT_LOWBITS, shift_b = PT_LOWBITS + 8;
    const uint32_t *n_a = c->pt_pad + 256;                 // padded length of the first pass's output, on the device
    const uint64_t n_words = n / 16;
    const int k = c->cfg.ksize;
    uint32_t *spos_a = seeding ? nullptr : c->pt_sposA, *spos_b = seeding ? nullptr : c->pt_sposB;
    if (d_keys)
        hipLaunchKernelGGL((k_pt_hist<uint32_t>), (uint32_t)ta, 256, 0, c->stream, d_keys, n, (const uint32_t *)nullptr, shift_a, ta, c->pt_histA);
    else if (c->cfg.canonical)
        hipLaunchKernelGGL((k_pt_hist_words<true>), (uint32_t)ta, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, shift_a, c->n_shards, c->shard, c->pt_histA);
    else
        hipLaunchKernelGGL((k_pt_hist_words<false>), (uint32_t)ta, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, shift_a, c->n_shards, c->shard, c->pt_histA);
    hipLaunchKernelGGL(k_pt_colsum, ca, 256, 0, c->stream, c->pt_histA, ra, c->pt_chunks);
    hipLaunchKernelGGL(k_pt_chunkscan, 256, 256, 0, c->stream, c->pt_chunks, ca, c->pt_totals);
    hipLaunchKernelGGL(k_pt_starts, 1, 256, 0, c->stream, c->pt_totals, 1, c->pt_pad);
    hipLaunchKernelGGL(k_pt_rowscan, ca, PT_ROWSCAN_THREADS, 0, c->stream, c->pt_histA, ra, c->pt_chunks, c->pt_pad, c->pt_scanA);
    hipLaunchKernelGGL(k_pt_zero_gaps, 256, 256, 0, c->stream, c->pt_totals, c->pt_pad, c->pt_keysA);
    const uint32_t ga = (uint32_t)((ta + 7) / 8 * 8), gb = (uint32_t)((tb + 7) / 8 * 8);
    if (d_keys)
        hipLaunchKernelGGL((k_pt_scatter<uint32_t>), ga, PT_THREADS, 0, c->stream, d_keys, n, (const uint32_t *)nullptr, shift_a, c->pt_scanA,
                           c->pt_keysA, spos_a);
    else if (c->cfg.canonical)
        hipLaunchKernelGGL((k_pt_scatter_words<true>), ga, PT_THREADS, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, shift_a,
                           c->n_shards, c->shard, c->pt_scanA, c->pt_keysA, spos_a);
    else
        hipLaunchKernelGGL((k_pt_scatter_words<false>), ga, PT_THREADS, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, shift_a,
                           c->n_shards, c->shard, c->pt_scanA, c->pt_keysA, spos_a);
    hipLaunchKernelGGL((k_pt_hist<uint32_t>), (uint32_t)tb, 256, 0, c->stream, c->pt_keysA, (uint64_t)0, n_a, shift_b, tb, c->pt_histB);
    hipLaunchKernelGGL(k_pt_colsum, cb, 256, 0, c->stream, c->pt_histB, rb, c->pt_chunks);
    hipLaunchKernelGGL(k_pt_chunkscan, 256, 256, 0, c->stream, c->pt_chunks, cb, c->pt_totals);
    hipLaunchKernelGGL(k_pt_starts, 1, 256, 0, c->stream, c->pt_totals, 0, c->pt_startB);
    hipLaunchKernelGGL(k_pt_rowscan, cb, PT_ROWSCAN_THREADS, 0, c->stream, c->pt_histB, rb, c->pt_chunks, c->pt_startB, c->pt_scanB);
    hipLaunchKernelGGL((k_pt_scatter<uint16_t>), gb, PT_THREADS, 0, c->stream, c->pt_keysA, (uint64_t)0, n_a, shift_b, c->pt_scanB,
                       c->pt_keysB, spos_b);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// The reverse passes: the flags in pt_flagsB (one per element of pt_keysB) go back to base positions - to d_flags (one byte
// each), or, with d_flags == nullptr, to the context's word flags.
static int pt_backward(nkml_ctx *c, uint64_t n, uint8_t *d_flags)
{
    const uint64_t ta = (n + PT_TILE - 1) / PT_TILE, tb = ta + 256 + 1;
    const uint32_t ga = (uint32_t)((ta + 7) / 8 * 8), gb = (uint32_t)((tb + 7) / 8 * 8);
    const uint32_t *n_a = c->pt_pad + 256;
    hipLaunchKernelGGL(k_pt_ungather, gb, PU_THREADS, 0, c->stream, c->pt_sposB, (uint64_t)0, n_a, c->pt_histB, c->pt_scanB,
                       c->pt_flagsB, c->pt_flagsA, (uint8_t)0, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_pt_ungather, ga, PU_THREADS, 0, c->stream, c->pt_sposA, n, (const uint32_t *)nullptr, c->pt_histA, c->pt_scanA,
                       c->pt_flagsA, d_flags, (uint8_t)FLAG_LOW, d_flags ? (uint32_t *)nullptr : c->d_wflags);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// The bucket kernels over keys grouped by bucket (keysB with bounds[b] .. bounds[b + 1] for bucket b): counts into the table,
// one flag per key into flagsB, tallies into the counters.
static int pt_count_buckets(nkml_ctx *c, const uint16_t *keysB, const uint32_t *bounds, uint8_t *flagsB, bool seeding, bool profile)
{
    const uint64_t buckets = (c->capacity >> PT_LOWBITS) ? (c->capacity >> PT_LOWBITS) : 1;
    const uint32_t depth = (uint32_t)c->cfg.depth;
    if (c->pt_wide)
        hipLaunchKernelGGL((k_pt_count_bucket<uint16_t>), (uint32_t)buckets, PB_THREADS, 0, c->stream, keysB, bounds,
                           (uint16_t *)c->pt_table, depth, seeding ? 1 : 0, flagsB, c->pt_tally);
    else
        hipLaunchKernelGGL((k_pt_count_bucket<uint8_t>), (uint32_t)buckets, PB_THREADS, 0, c->stream, keysB, bounds,
                           (uint8_t *)c->pt_table, depth, seeding ? 1 : 0, flagsB, c->pt_tally);
    if (profile) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    hipLaunchKernelGGL(k_pt_tally_reduce, (uint32_t)((buckets + 1023) / 1024 < 64 ? (buckets + 1023) / 1024 : 64), 1024, 0, c->stream, c->pt_tally, (uint32_t)buckets, c->d_ctr);
    const uint32_t big_group = buckets >= 256 * 1024 ? 256u : buckets >= 2048 ? (uint32_t)(buckets / 1024) : 1u;
    if (c->pt_wide)
        hipLaunchKernelGGL((k_pt_count_bucket_big<uint16_t>), (uint32_t)((buckets + big_group - 1) / big_group), 256, 65536, c->stream, keysB, bounds,
                           (uint16_t *)c->pt_table, depth, seeding ? 1 : 0, flagsB, c->d_ctr, (uint32_t)buckets, big_group);
    else
        hipLaunchKernelGGL((k_pt_count_bucket_big<uint8_t>), (uint32_t)((buckets + big_group - 1) / big_group), 256, 65536, c->stream, keysB, bounds,
                           (uint8_t *)c->pt_table, depth, seeding ? 1 : 0, flagsB, c->d_ctr, (uint32_t)buckets, big_group);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int run_count_partitioned(nkml_ctx *c, const void *d_keys_any, uint64_t n, uint64_t n_bases, uint8_t *d_flags, bool seeding)
{
    if (c->broken) return broken_context(c);
    if (n == 0) return NKML_OK;
    if (c->hashpart) return run_count_hashpart(c, static_cast<const unsigned long long *>(d_keys_any), n, n_bases, d_flags, seeding);
    const uint32_t *d_keys = static_cast<const uint32_t *>(d_keys_any);
    if (!c->pt_table) return fail(c, NKML_E_STATE, "the context was created without a table (NKML_F_NO_TABLE)");
    if (n > c->max_words * 16) return fail(c, NKML_E_ARG, "%llu keys exceed the context's batch size", (unsigned long long)n);
    const bool profile = (c->cfg.flags & NKML_F_PROFILE) != 0 && !seeding;
    int rc;
    if (profile && c->pt_timing_pending && (rc = collect_partition_timing(c)) != NKML_OK) return rc; // the events are about to be recorded again
    const uint64_t ta = (n + PT_TILE - 1) / PT_TILE, tb = ta + 256 + 1;
    const int shift_a = PT_LOWBITS, shift_b = PT_LOWBITS + 8;
    const uint32_t *n_a = c->pt_pad + 256;
    const uint64_t n_words = n / 16;
    const int k = c->cfg.ksize;
    const bool verify = c->verify && !seeding;
    if (verify) {
        const uint64_t entries = c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity;
        if ((rc = verify_snapshot(c, c->pt_table, entries * (c->pt_wide ? 2 : 1))) != NKML_OK) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->pt_histA + 256 * ta, 0, 256 * sizeof(pt_hist_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->pt_histB + 256 * tb, 0, 256 * sizeof(pt_hist_t), c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if ((rc = pt_forward(c, d_keys, n, n_bases, seeding)) != NKML_OK) return rc;
    if (verify) {
        if (c->vf_fault == 2) hipLaunchKernelGGL(k_verify_fault_swap<uint32_t>, 1, 1, 0, c->stream, c->pt_keysA, n_a);
        VerifySource sa{d_keys, n, nullptr, c->d_packed, c->d_startbits, n_bases, n_words, k, 0, c->n_shards, c->shard};
        verify_pass<uint32_t, uint32_t>(c, sa, (uint32_t)ta, shift_a, c->pt_scanA, c->pt_keysA, 0xffffffffull);
        VerifySource sb{c->pt_keysA, 0, n_a, nullptr, nullptr, 0, 0, k, 0, 1, 0};
        verify_pass<uint32_t, uint16_t>(c, sb, (uint32_t)tb, shift_b, c->pt_scanB, c->pt_keysB, (1ull << PT_LOWBITS) - 1);
        HIPCHK(c, hipGetLastError());
    }
    const uint64_t buckets = (c->capacity >> PT_LOWBITS) ? (c->capacity >> PT_LOWBITS) : 1;
    const uint32_t depth = (uint32_t)c->cfg.depth;
    if (profile) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_pt_bucket_bounds, (uint32_t)(buckets / 256 + 1), 256, 0, c->stream, c->pt_scanB, c->pt_pad, (uint32_t)buckets, c->pt_bounds);
    if ((rc = pt_count_buckets(c, c->pt_keysB, c->pt_bounds, c->pt_flagsB, seeding, profile)) != NKML_OK) return rc;
    if (verify) {
        const size_t vf_lds = 3 * VF_WINDOW * sizeof(uint32_t);
        if (c->vf_fault == 1) hipLaunchKernelGGL(k_verify_fault_flag, 1, 1, 0, c->stream, c->pt_bounds, (uint32_t)buckets, c->pt_flagsB);
        if (c->pt_wide)
            hipLaunchKernelGGL((k_verify_bucket<uint16_t>), (uint32_t)buckets, 512, vf_lds, c->stream, c->pt_keysB, c->pt_bounds, (const uint16_t *)c->pt_table,
                               (const uint16_t *)c->vf_before, depth, c->pt_flagsB, c->d_ctr);
        else
            hipLaunchKernelGGL((k_verify_bucket<uint8_t>), (uint32_t)buckets, 512, vf_lds, c->stream, c->pt_keysB, c->pt_bounds, (const uint8_t *)c->pt_table,
                               (const uint8_t *)c->vf_before, depth, c->pt_flagsB, c->d_ctr);
        HIPCHK(c, hipGetLastError());
    }
    if (!seeding && (rc = pt_backward(c, n, d_flags)) != NKML_OK) return rc;
#ifdef PB_STAMP
    hipLaunchKernelGGL(k_pt_stamp_print, 1, 1, 0, c->stream, (uint32_t)buckets);
#endif
    if (profile) {
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        c->pt_timed_keys = n;
        c->pt_timing_pending = true;
    }
    if (verify) return verify_outcome(c);
    return NKML_OK;
}

// Doubles the hashed partitioned table (every bucket gets twice the slots).  Waits for the device: the old table is
// freed afterwards.  remap: a batch is under way, its per-workgroup "done" marks and tallies follow the slots.
__global__ void __launch_bounds__(256) k_hp_remap_done(const uint32_t *__restrict__ done_old, const uint4 *__restrict__ tally_old, uint32_t n_old,
                                                       uint32_t *__restrict__ done_new, uint4 *__restrict__ tally_new)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_old) return;
    // workgroup i's sub-range of slots becomes the two sub-ranges 2i and 2i + 1
    done_new[2 * i] = done_new[2 * i + 1] = done_old[i];
    tally_new[2 * i] = done_old[i] ? tally_old[i] : make_uint4(0u, 0u, 0u, 0u);
    tally_new[2 * i + 1] = make_uint4(0u, 0u, 0u, 0u);
}

HpGeom hp_geom(const nkml_ctx *c, int log2_bucket)
{
    int ll = log2_bucket < c->hp_log2_local_max ? log2_bucket : c->hp_log2_local_max;
    if (log2_bucket - ll > HB_MAX_SUBS_LOG2) ll = log2_bucket - HB_MAX_SUBS_LOG2;
    return HpGeom{log2_bucket, ll};
}

int hp_grow(nkml_ctx *c, bool batch_under_way)
{
    if (c->hp_log2_bucket + 1 > HP_MAX_LOG2_BUCKET || hp_geom(c, c->hp_log2_bucket + 1).log2_local > HB_MAX_SLOTS_LOG2)
        return fail(c, NKML_E_NOMEM, "hashed table cannot grow beyond 2^%d slots", 16 + c->hp_log2_bucket);
    const HpGeom og = hp_geom(c, c->hp_log2_bucket), ng = hp_geom(c, c->hp_log2_bucket + 1);
    const uint64_t ncap = c->capacity * 2;
    unsigned long long *nt = nullptr;
    HIPCHK(c, hipMalloc(&nt, ncap * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_hp_grow, 65536u << (ng.log2_bucket - ng.log2_local), HB_THREADS, ((size_t)8) << ng.log2_local, c->stream, c->hp_table, og, nt, ng,
                       c->d_ctr);
    HIPCHK(c, hipGetLastError());
    if (batch_under_way && ng.log2_bucket - ng.log2_local > og.log2_bucket - og.log2_local) { // the sub-ranges per bucket double
        const uint32_t n_old = 65536u << (og.log2_bucket - og.log2_local);
        uint32_t *d2 = nullptr;
        uint4 *t2 = nullptr;
        hipError_t e = hipMalloc(&d2, (uint64_t)n_old * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&t2, (uint64_t)n_old * sizeof(uint4));
        if (e == hipSuccess) e = hipMemcpyAsync(d2, c->hp_done, (uint64_t)n_old * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(t2, c->pt_tally, (uint64_t)n_old * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_hp_remap_done, blocks_for(n_old), 256, 0, c->stream, d2, t2, n_old, c->hp_done, c->pt_tally);
            e = hipGetLastError();
        }
        const hipError_t e_sync = hipStreamSynchronize(c->stream); // also on failure: nothing may still read d2 / t2 / nt
        hipFree(d2);
        hipFree(t2);
        if (e != hipSuccess || e_sync != hipSuccess) {
            hipFree(nt);
            HIPCHK(c, e != hipSuccess ? e : e_sync);
        }
    }
    {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipFree(nt); HIPCHK(c, e); }
    }
    HIPCHK(c, hipFree(c->hp_table));
    c->hp_table = nt;
    c->capacity = ncap;
    c->hp_log2_bucket++;
    c->resizes++;
    return NKML_OK;
}

// The partitioned path sweeps the whole table through LDS once per batch, the atomic path touches one slot per k-mer
// occurrence: once the table has grown well beyond a batch's positions, the sweep costs more than the random accesses it
// replaced (measured on 150-base pairs: 2 M-pair batches break even between 2^31 and 2^32 slots, 0.7 M-pair batches between 2^29 and 2^30).  From
// then on the context counts on the atomic path: every stored k-mer moves to a table of 16-byte slots (same number of
// slots; count as it stands; saturated if count >= depth - 1, which is what both paths hold between batches), the
// partition buffers go, the atomic path's lists come.  Called between batches only.  If the memory for the move is not
// to be had the context stays where it is.
__global__ void __launch_bounds__(256) k_hp_to_slots(const unsigned long long *__restrict__ table, int log2_bucket, Slot *__restrict__ slots,
                                                     uint64_t cap_mask, uint32_t depth, Counters *ctr)
{
    const uint64_t cap = (uint64_t)65536 << log2_bucket, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += stride) {
        const unsigned long long v = table[i];
        if (!v) continue;
        const unsigned long long code = hp_unhash(((i >> log2_bucket) << 48) | (v & HP_KEYMASK));
        const uint32_t cnt = (uint32_t)(v >> HP_COUNT_SHIFT) & HP_CMAX;
        uint64_t h = mix64(code) & cap_mask, probes = 0; // where find_or_insert starts
        for (; probes <= cap_mask; probes++) {
            if (atomicCAS(&slots[h].key, 0ull, code) == 0ull) {
                slots[h].count = cnt;
                slots[h].state = cnt + 1 >= depth ? ST_SAT : 0u;
                break;
            }
            h = (h + 1) & cap_mask;
        }
        if (probes > cap_mask) atomicAdd(&ctr->probe_overflow, 1u);
    }
}

int hp_leave(nkml_ctx *c)
{
    int rc;
    if ((rc = fetch_counters(c)) != NKML_OK) return rc; // waits for the stream
    if ((rc = collect_partition_timing(c)) != NKML_OK) return rc;
    c->hp_leave_slots = 0;
    Slot *ns = nullptr;
    uint32_t *filter = nullptr;
    if (hipMalloc(&ns, c->capacity * sizeof(Slot)) != hipSuccess || hipMalloc(&filter, FILTER_WORDS * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(ns);
        return NKML_OK;
    }
    // the partition buffers first: the lists need room too
    void *gone[] = {c->hp_keysA, c->hp_keysB, c->hp_done, c->pt_flagsA, c->pt_flagsB, c->pt_sposA, c->pt_sposB, c->pt_histA, c->pt_scanA,
                    c->pt_histB, c->pt_scanB, c->pt_pad, c->pt_startB, c->pt_tally, c->pt_totals, c->pt_chunks};
    for (void *p : gone) hipFree(p);
    c->hp_keysA = c->hp_keysB = nullptr;
    c->hp_done = nullptr;
    c->pt_flagsA = c->pt_flagsB = nullptr;
    c->pt_sposA = c->pt_sposB = c->pt_scanA = c->pt_scanB = c->pt_pad = c->pt_startB = c->pt_totals = c->pt_chunks = nullptr;
    c->pt_histA = c->pt_histB = nullptr;
    c->pt_tally = nullptr;
    c->slots = ns;
    c->filter = filter;
    HIPCHK(c, zero_device(ns, c->capacity * sizeof(Slot), c->stream));
    HIPCHK(c, hipMemsetAsync(filter, 0, FILTER_WORDS * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_hp_to_slots, GRID_PERSIST, 256, 0, c->stream, c->hp_table, c->hp_log2_bucket, ns, c->capacity - 1, (uint32_t)c->cfg.depth, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(c->hp_table));
    c->hp_table = nullptr;
    c->partitioned = c->hashpart = false;
    c->used_bound = c->h_ctr->unique;
    return alloc_atomic_lists(c);
}

// before a batch: is it time to leave the partitioned path?
inline int hp_maybe_leave(nkml_ctx *c)
{
    if (!c->hashpart || !c->hp_leave_slots || c->capacity <= c->hp_leave_slots) return NKML_OK;
    return hp_leave(c);
}

// The hashed table's partitioned counting of one batch: histogram + two partition passes on the hashes' top 16 bits,
// the bucket kernel (relaunched after a doubling of the table for the buckets that did not fit), two reverse passes.
// Waits for the device once per batch, after the bucket kernel (whether every bucket fitted is only known there).
int run_count_hashpart(nkml_ctx *c, const unsigned long long *d_keys, uint64_t n, uint64_t n_bases, uint8_t *d_flags, bool seeding)
{
    if (n > c->max_words * 16) return fail(c, NKML_E_ARG, "%llu keys exceed the context's batch size", (unsigned long long)n);
    if (d_keys && (reinterpret_cast<uintptr_t>(d_keys) & 15u)) return fail(c, NKML_E_ARG, "the key array must be 16-byte aligned");
    int rc;
    // between batches: the table doubles while this batch, if it adds as many k-mers as the last one did, would take it
    // beyond 75 % full (store_kmer grows at 80 %, normalise_kmers_multi_large.c:143, 933).  Every batch sweeps the whole
    // table through LDS, so a table larger than needed costs bandwidth; one that turns out too small costs a relaunch.
    {
        const uint64_t now = c->h_ctr->unique, added = now > c->hp_unique_before ? now - c->hp_unique_before : 0;
        static const double grow_load = [] { // tuning aid: NKML_HP_GROW_LOAD=0.5 .. 0.95
            const char *e = getenv("NKML_HP_GROW_LOAD");
            const double v = e ? atof(e) : 0.0;
            return v >= 0.5 && v <= 0.95 ? v : 0.75;
        }();
        while ((double)(now + added) >= grow_load * (double)c->capacity && c->hp_log2_bucket < HP_MAX_LOG2_BUCKET) {
            if (hp_grow(c, false) != NKML_OK) break; // no room for a larger table now: the bucket kernel will say if it is needed
        }
        c->hp_unique_before = now;
    }
    const bool profile = (c->cfg.flags & NKML_F_PROFILE) != 0 && !seeding;
    if (profile && c->pt_timing_pending && (rc = collect_partition_timing(c)) != NKML_OK) return rc;
    const uint64_t ta = (n + PT_TILE - 1) / PT_TILE, tb = ta + 256 + 1, ra = ta + 1, rb = tb + 1;
    const uint32_t ca = (uint32_t)((ra + PT_CH - 1) / PT_CH), cb = (uint32_t)((rb + PT_CH - 1) / PT_CH);
    const uint32_t *n_a = c->pt_pad + 256;
    const bool verify = c->verify && !seeding;
    const HpGeom geo_before = hp_geom(c, c->hp_log2_bucket);
    if (verify && (rc = verify_snapshot(c, c->hp_table, c->capacity * sizeof(unsigned long long))) != NKML_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->pt_histA + 256 * ta, 0, 256 * sizeof(pt_hist_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->pt_histB + 256 * tb, 0, 256 * sizeof(pt_hist_t), c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    const uint64_t n_words = n / 16;
    const int k = c->cfg.ksize;
    uint32_t *spos_a = seeding ? nullptr : c->pt_sposA, *spos_b = seeding ? nullptr : c->pt_sposB;
    if (d_keys)
        hipLaunchKernelGGL((k_hp_hist<true>), (uint32_t)ta, 256, 0, c->stream, d_keys, n, (const uint32_t *)nullptr, HP_SHIFT_A, c->pt_histA);
    else if (c->cfg.canonical)
        hipLaunchKernelGGL((k_hp_hist_words<true>), (uint32_t)ta, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, c->pt_histA);
    else
        hipLaunchKernelGGL((k_hp_hist_words<false>), (uint32_t)ta, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, c->pt_histA);
    hipLaunchKernelGGL(k_pt_colsum, ca, 256, 0, c->stream, c->pt_histA, ra, c->pt_chunks);
    hipLaunchKernelGGL(k_pt_chunkscan, 256, 256, 0, c->stream, c->pt_chunks, ca, c->pt_totals);
    hipLaunchKernelGGL(k_pt_starts, 1, 256, 0, c->stream, c->pt_totals, 1, c->pt_pad);
    hipLaunchKernelGGL(k_pt_rowscan, ca, PT_ROWSCAN_THREADS, 0, c->stream, c->pt_histA, ra, c->pt_chunks, c->pt_pad, c->pt_scanA);
    hipLaunchKernelGGL(k_hp_zero_gaps, 256, 256, 0, c->stream, c->pt_totals, c->pt_pad, c->hp_keysA);
    const uint32_t ga = (uint32_t)((ta + 7) / 8 * 8), gb = (uint32_t)((tb + 7) / 8 * 8);
    if (d_keys)
        hipLaunchKernelGGL((k_hp_scatter<true>), ga, PT_THREADS, 0, c->stream, d_keys, n, (const uint32_t *)nullptr, HP_SHIFT_A, c->pt_scanA, c->hp_keysA, spos_a);
    else if (c->cfg.canonical)
        hipLaunchKernelGGL((k_hp_scatter_words<true>), ga, PT_THREADS, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, c->pt_scanA, c->hp_keysA, spos_a);
    else
        hipLaunchKernelGGL((k_hp_scatter_words<false>), ga, PT_THREADS, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, c->pt_scanA, c->hp_keysA, spos_a);
    hipLaunchKernelGGL((k_hp_hist<false>), (uint32_t)tb, 256, 0, c->stream, c->hp_keysA, (uint64_t)0, n_a, HP_SHIFT_B, c->pt_histB);
    hipLaunchKernelGGL(k_pt_colsum, cb, 256, 0, c->stream, c->pt_histB, rb, c->pt_chunks);
    hipLaunchKernelGGL(k_pt_chunkscan, 256, 256, 0, c->stream, c->pt_chunks, cb, c->pt_totals);
    hipLaunchKernelGGL(k_pt_starts, 1, 256, 0, c->stream, c->pt_totals, 0, c->pt_startB);
    hipLaunchKernelGGL(k_pt_rowscan, cb, PT_ROWSCAN_THREADS, 0, c->stream, c->pt_histB, rb, c->pt_chunks, c->pt_startB, c->pt_scanB);
    hipLaunchKernelGGL((k_hp_scatter<false>), gb, PT_THREADS, 0, c->stream, c->hp_keysA, (uint64_t)0, n_a, HP_SHIFT_B, c->pt_scanB, c->hp_keysB, spos_b);
    HIPCHK(c, hipGetLastError());
    if (verify) {
        if (c->vf_fault == 2) hipLaunchKernelGGL(k_verify_fault_swap<unsigned long long>, 1, 1, 0, c->stream, c->hp_keysA, n_a);
        VerifySource sa{d_keys, n, nullptr, c->d_packed, c->d_startbits, n_bases, n_words, k, 1, 1, 0};
        verify_pass<unsigned long long, unsigned long long>(c, sa, (uint32_t)ta, HP_SHIFT_A, c->pt_scanA, c->hp_keysA, ~0ull);
        VerifySource sb{c->hp_keysA, 0, n_a, nullptr, nullptr, 0, 0, k, 0, 1, 0};
        verify_pass<unsigned long long, unsigned long long>(c, sb, (uint32_t)tb, HP_SHIFT_B, c->pt_scanB, c->hp_keysB, ~0ull);
        HIPCHK(c, hipGetLastError());
        if (!c->vf_keys) HIPCHK(c, hipMalloc(&c->vf_keys, c->max_words * 16 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemcpyAsync(c->vf_keys, c->hp_keysB, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    }
    {
        const HpGeom g0 = hp_geom(c, c->hp_log2_bucket);
        HIPCHK(c, hipMemsetAsync(c->hp_done, 0, ((uint64_t)65536 << (g0.log2_bucket - g0.log2_local)) * sizeof(uint32_t), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(&c->d_ctr->hp_left, 0, sizeof(unsigned int), c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_pt_bucket_bounds, 65536u / 256 + 1, 256, 0, c->stream, c->pt_scanB, c->pt_pad, 65536u, c->pt_bounds);
    uint32_t n_ranges = 0;
    const bool general_only = getenv("NKML_HP_GENERAL_KERNEL") != nullptr; // tuning / test aid: every bucket through the general kernel
    for (int attempt = 0;; attempt++) {
        const HpGeom geo = hp_geom(c, c->hp_log2_bucket);
        n_ranges = 65536u << (geo.log2_bucket - geo.log2_local);
        const size_t lds = ((size_t)12) << geo.log2_local;
        // the first pass's output is free by now: the workgroups regroup their buckets' keys into it
        if (!general_only) {
            if (geo.log2_bucket > geo.log2_local)
                hipLaunchKernelGGL(k_hp_count_bucket_fast<true>, 65536u, HB_THREADS, lds, c->stream, c->hp_keysB, c->pt_bounds, c->hp_table, geo, (uint32_t)c->cfg.depth,
                                   seeding ? 1 : 0, attempt == 0 ? 1 : 0, c->pt_flagsB, c->pt_tally, c->hp_done, c->d_ctr, c->hp_keysA);
            else
                hipLaunchKernelGGL(k_hp_count_bucket_fast<false>, 65536u, HB_THREADS, lds, c->stream, c->hp_keysB, c->pt_bounds, c->hp_table, geo, (uint32_t)c->cfg.depth,
                                   seeding ? 1 : 0, attempt == 0 ? 1 : 0, c->pt_flagsB, c->pt_tally, c->hp_done, c->d_ctr, c->hp_keysA);
            HIPCHK(c, hipGetLastError());
        }
        // from here on part of the batch is in the table: a failure leaves the context unusable until nkml_reset
        c->broken = true;
        if (profile && attempt == 0) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        if ((rc = fetch_counters(c)) != NKML_OK) return rc; // did every bucket fit?  did the fast kernel leave any?
        if (general_only || c->h_ctr->hp_left) {
            // buckets with more keys than 16-bit counters hold (a k-mer repeated throughout the batch): the general kernel
            // takes every sub-range that is not marked done
            hipLaunchKernelGGL(k_hp_count_bucket, 65536u, HB_THREADS, lds, c->stream, c->hp_keysB, c->pt_scanB, c->pt_pad, c->hp_table, geo,
                               (uint32_t)c->cfg.depth, seeding ? 1 : 0, c->pt_flagsB, c->pt_tally, c->hp_done, c->d_ctr, c->hp_keysA);
            HIPCHK(c, hipGetLastError());
            if ((rc = fetch_counters(c)) != NKML_OK) return rc;
        }
        if (!c->h_ctr->probe_overflow) break;
        if (attempt >= HP_MAX_LOG2_BUCKET) return fail(c, NKML_E_STATE, "hashed table overflow (capacity %llu)", (unsigned long long)c->capacity);
        HIPCHK(c, hipMemsetAsync(&c->d_ctr->probe_overflow, 0, sizeof(unsigned int), c->stream));
        c->h_ctr->probe_overflow = 0;
        if ((rc = hp_grow(c, true)) != NKML_OK) return rc;
    }
    c->broken = false; // every sub-range of the batch is committed
    if (verify) {
        if (c->vf_fault == 1) hipLaunchKernelGGL(k_verify_fault_flag, 1, 1, 0, c->stream, c->pt_bounds, 65536u, c->pt_flagsB);
        hipLaunchKernelGGL(k_verify_hbucket, 65536u, 512, 3 * VF_WINDOW * sizeof(uint32_t), c->stream, c->vf_keys, c->pt_bounds, c->hp_table,
                           hp_geom(c, c->hp_log2_bucket), static_cast<const unsigned long long *>(c->vf_before), geo_before, (uint32_t)c->cfg.depth,
                           c->pt_flagsB, c->d_ctr);
        HIPCHK(c, hipGetLastError());
    }
#ifdef HB_STAMP
    hipLaunchKernelGGL(k_hb_stamp_print, 1, 1, 0, c->stream);
#endif
    hipLaunchKernelGGL(k_pt_tally_reduce, (n_ranges + 1023) / 1024 < 64 ? (n_ranges + 1023) / 1024 : 64, 1024, 0, c->stream, c->pt_tally, n_ranges, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    if (!seeding) {
        hipLaunchKernelGGL(k_pt_ungather, gb, PU_THREADS, 0, c->stream, c->pt_sposB, (uint64_t)0, n_a, c->pt_histB, c->pt_scanB,
                           c->pt_flagsB, c->pt_flagsA, (uint8_t)0, (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_pt_ungather, ga, PU_THREADS, 0, c->stream, c->pt_sposA, n, (const uint32_t *)nullptr, c->pt_histA, c->pt_scanA,
                           c->pt_flagsA, d_flags, (uint8_t)FLAG_LOW, d_flags ? (uint32_t *)nullptr : c->d_wflags);
        HIPCHK(c, hipGetLastError());
    }
    if (profile) {
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        c->pt_timed_keys = n;
        c->pt_timing_pending = true;
    }
    if (verify) return verify_outcome(c);
    return NKML_OK;
}

// Replicated-reads mode: this shard's k-mers of the batch are pulled together (stream order kept) and counted as a key
// array; the flags go back to the packed words' flag words.  Waits once for the number of k-mers the shard owns.
int run_count_shard(nkml_ctx *c, uint64_t n_words, uint64_t n_bases)
{
    const uint32_t tiles = (uint32_t)((n_words + 255) / 256);
    const int k = c->cfg.ksize;
#define SH_LAUNCH(KERNEL, ...)                                                                                                               \
    do {                                                                                                                                     \
        if (c->cfg.canonical)                                                                                                                \
            hipLaunchKernelGGL((KERNEL<true>), tiles, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, __VA_ARGS__);      \
        else                                                                                                                                 \
            hipLaunchKernelGGL((KERNEL<false>), tiles, 256, 0, c->stream, c->d_packed, c->d_startbits, n_bases, n_words, k, __VA_ARGS__);     \
    } while (0)
    HIPCHK(c, hipMemsetAsync(c->sh_tile_count + tiles, 0, sizeof(uint32_t), c->stream));
    SH_LAUNCH(k_sh_count, c->n_shards, c->shard, c->sh_own, c->sh_tile_count);
    HIPCHK(c, hipGetLastError());
    size_t tmp = c->sh_scan_tmp_bytes;
    HIPCHK(c, rocprim::exclusive_scan(c->sh_scan_tmp, tmp, c->sh_tile_count, c->sh_tile_base, 0u, (size_t)tiles + 1, rocprim::plus<uint32_t>(), c->stream));
    uint32_t *h_own = c->h_route_starts + 70;
    HIPCHK(c, hipMemcpyAsync(h_own, c->sh_tile_base + tiles, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t n_own = *h_own, n16 = (n_own + 15) & ~15ull;
    if (n16 > n_own) HIPCHK(c, hipMemsetAsync(c->sh_keys + n_own, 0, (n16 - n_own) * sizeof(uint32_t), c->stream));
    SH_LAUNCH(k_sh_compact, c->sh_own, c->sh_tile_base, c->sh_keys);
    HIPCHK(c, hipGetLastError());
    int rc = NKML_OK;
    if (n16 && (rc = run_count_partitioned(c, c->sh_keys, n16, 0, c->sh_flags, false)) != NKML_OK) return rc;
    hipLaunchKernelGGL(k_sh_expand, tiles, 256, 0, c->stream, c->sh_own, n_words, c->sh_tile_base, c->sh_flags, c->d_wflags);
    HIPCHK(c, hipGetLastError());
#undef SH_LAUNCH
    return NKML_OK;
}

// read back the event pairs of the last partitioned batch (after the stream has been waited for)
int collect_partition_timing(nkml_ctx *c)
{
    if (!c->pt_timing_pending) return NKML_OK;
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    float whole = 0, bucket = 0;
    HIPCHK(c, hipEventElapsedTime(&whole, c->ev[0], c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&bucket, c->ev[2], c->ev[3]));
    c->count_ms += bucket;
    c->pt_stage_ms += whole;
    c->count_launches++;
    c->count_keys += c->pt_timed_keys;
    c->pt_timing_pending = false;
    return NKML_OK;
}

int run_reduce_words(nkml_ctx *c, const uint64_t *d_offsets, uint32_t n_reads, int32_t *d_high, int32_t *d_total)
{
    if (n_reads == 0) return NKML_OK;
    hipLaunchKernelGGL(k_reduce_words, blocks_for(n_reads), 256, 0, c->stream, c->d_wflags, d_offsets, n_reads, d_high, d_total);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int check_batch(nkml_ctx *c, uint32_t n_reads, uint64_t n_bases)
{
    if (n_reads > c->cfg.max_batch_reads) return fail(c, NKML_E_ARG, "batch of %u reads exceeds max_batch_reads %u", n_reads, c->cfg.max_batch_reads);
    if (n_bases > c->cfg.max_batch_bases) return fail(c, NKML_E_ARG, "batch of %llu bases exceeds max_batch_bases %llu", (unsigned long long)n_bases, (unsigned long long)c->cfg.max_batch_bases);
    return NKML_OK;
}

int check_bad_bases(nkml_ctx *c)
{
    if (c->h_ctr->bad_bases) {
        unsigned int n = c->h_ctr->bad_bases;
        hipMemsetAsync(&c->d_ctr->bad_bases, 0, sizeof(unsigned int), c->stream);
        return fail(c, NKML_E_BADBASE, "%u sequence byte(s) outside ACGTN", n);
    }
    return NKML_OK;
}

} // namespace

extern "C" {

int nkml_abi_version(void) { return NKML_ABI_VERSION; }

const char *nkml_strerror(int code)
{
    switch (code) {
    case NKML_OK: return "ok";
    case NKML_E_ARG: return "invalid argument";
    case NKML_E_HIP: return "HIP runtime error";
    case NKML_E_NOMEM: return "out of memory";
    case NKML_E_BADBASE: return "sequence byte outside ACGTN";
    case NKML_E_STATE: return "invalid call order or internal state";
    case NKML_E_NODEVICE: return "no gfx950 device";
    default: return "unknown error";
    }
}

const char *nkml_last_error(const nkml_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int nkml_create(const nkml_config *cfg, nkml_ctx **out)
{
    if (!cfg || !out) return NKML_E_ARG;
    *out = nullptr;
    if (cfg->ksize < 5 || cfg->ksize > 31) return NKML_E_ARG; // normalise_kmers_multi_large.c:724
    if (cfg->depth < 2) return NKML_E_ARG;                    // :734
    if (cfg->max_batch_bases == 0 || cfg->max_batch_bases >= (1ull << 32) || cfg->max_batch_reads == 0) return NKML_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return NKML_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return NKML_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return NKML_E_NODEVICE;
    if (hipSetDevice(cfg->device) != hipSuccess) return NKML_E_NODEVICE;

    nkml_ctx *c = new (std::nothrow) nkml_ctx();
    if (!c) return NKML_E_NOMEM;
    memset(c, 0, sizeof *c);
    c->cfg = *cfg;
    c->direct = cfg->ksize <= 16;
    {
        // the partitioned path needs 30-bit codes, saturating counts of 7 or 15 bits and no exact counts past depth; it pays
        // where a batch spreads over many 2^14-code buckets (4^15 / 2^14 = 65536 at k = 15)
        // (and batches whose padded first-pass output still fits 32-bit positions)
        const bool can = cfg->ksize <= 15 && cfg->depth <= (int32_t)PT_MAX_DEPTH_WIDE && !(cfg->flags & NKML_F_EXACT_COUNTS) &&
                         cfg->max_batch_bases < (1ull << 32) - (4ull << 20);
        c->pt_wide = cfg->depth > 127;
        const char *env = getenv("NKML_PARTITIONED");
        // chosen by default where a full batch leaves the buckets well inside the fast bucket kernel's reach
        // (fewer than 65535 keys each): k = 15 always, k = 14 up to about 2 M pairs of 150 bases per batch
        const uint64_t buckets = (1ull << (2 * cfg->ksize)) >> 14;
        bool want = cfg->ksize >= 14 && cfg->ksize <= 15 && cfg->max_batch_bases / buckets <= 40000;
        if (env && (env[0] == '0' || env[0] == '1')) want = env[0] == '1';
        if (cfg->flags & NKML_F_COUNT_PARTITIONED) want = true;
        if (cfg->flags & NKML_F_COUNT_ATOMIC) want = false;
        c->partitioned = can && want;
        // the hashed table's partitioned path (k > 16): same conditions on depth, flags and batch size
        const bool can_hp = cfg->ksize > 16 && cfg->depth <= (int32_t)PT_MAX_DEPTH_WIDE && !(cfg->flags & NKML_F_EXACT_COUNTS) &&
                            cfg->max_batch_bases < (1ull << 32) - (4ull << 20);
        bool want_hp = true;
        if (env && (env[0] == '0' || env[0] == '1')) want_hp = env[0] == '1';
        if (cfg->flags & NKML_F_COUNT_PARTITIONED) want_hp = true;
        if (cfg->flags & NKML_F_COUNT_ATOMIC) want_hp = false;
        if (can_hp && want_hp) c->partitioned = c->hashpart = true;
        if ((cfg->flags & NKML_F_NO_TABLE) && (!c->partitioned || c->hashpart)) {
            delete c;
            return NKML_E_ARG; // a context without a table only makes sense as a sender of the bucket-ordered exchange (k <= 15)
        }
    }
    {
        const char *env = getenv("NKML_VERIFY_RANKS");
        c->verify = (cfg->flags & NKML_F_VERIFY_RANKS) != 0 || (env && env[0] == '1');
        if (const char *f = getenv("NKML_TEST_VERIFY_FAULT")) c->vf_fault = atoi(f);
    }
    c->n_shards = 1; // (the memset above also wiped the member initialisers)
    c->shard = 0;
    c->key_bytes = c->direct ? 4 : 8;
    uint32_t chunk = cfg->chunk_bases ? cfg->chunk_bases : (32u << 20);
    chunk = (chunk + 63) & ~63u;
    if (chunk > (1u << 30)) chunk = 1u << 30;
    c->chunk = chunk;
    c->max_words = (cfg->max_batch_bases + 15) / 16;
    const uint64_t npos = c->max_words * 16;
    c->posbits = 0;
    while ((1ull << c->posbits) < npos) c->posbits++;
    if (c->hashpart) {
        // chosen by default: stay while the table has at most 4 slots per position of a full batch (hp_leave);
        // asked for by flag or environment: stay for good
        const char *env = getenv("NKML_PARTITIONED");
        const bool forced = (cfg->flags & NKML_F_COUNT_PARTITIONED) || (env && env[0] == '1');
        c->hp_leave_slots = forced ? 0 : 4 * npos;
        if (const char *e = getenv("NKML_HP_LEAVE_SLOTS")) c->hp_leave_slots = strtoull(e, nullptr, 10); // test aid
    }

#define CREATE_CHK(call)                                                   \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != NKML_OK) { *out = c; return rc_; }                      \
    } while (0)
#define CREATE_HIP(call)                                                                                         \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int rc_ = fail(c, e_ == hipErrorOutOfMemory ? NKML_E_NOMEM : NKML_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
            *out = c;                                                                                            \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)

    CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    if (cfg->flags & NKML_F_PROFILE)
        for (int i = 0; i < 2 * 64; i++) CREATE_HIP(hipEventCreate(&c->ev[i]));
    CREATE_HIP(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
    CREATE_HIP(hipMalloc(&c->d_ctr, sizeof(Counters)));
    CREATE_HIP(hipHostMalloc(&c->h_ctr, sizeof(Counters), hipHostMallocDefault));
    memset(c->h_ctr, 0, sizeof(Counters));
    CREATE_HIP(hipHostMalloc(&c->h_route_starts, 72 * sizeof(uint32_t), hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lock(g_device_mutex);
        if (cfg->device < MAX_DEVICES && !g_scratch_ptr[cfg->device]) {
            // this device's copy of the static scratch array (the device is current: hipSetDevice above)
            void *p = nullptr;
            CREATE_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[cfg->device] = static_cast<uint32_t *>(p);
        }
        // stream-order ranks come from returning LDS adds (partitioned counting, routing to more than two shards): make sure
        // this device serves them in lane order, with every CU as busy with such workgroups as the tile kernels keep it
        int &state = g_lds_order[cfg->device < MAX_DEVICES ? cfg->device : MAX_DEVICES - 1];
        if (state == 0 || cfg->device >= MAX_DEVICES) {
            uint32_t bad = 0;
            CREATE_HIP(hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream));
            const uint32_t cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
            hipLaunchKernelGGL(k_lds_order_check, cus * 4u, LOC_WAVES * 64, 0, c->stream, reinterpret_cast<uint32_t *>(c->d_ctr));
            CREATE_HIP(hipGetLastError());
            CREATE_HIP(hipMemcpyAsync(&bad, c->d_ctr, sizeof bad, hipMemcpyDeviceToHost, c->stream));
            CREATE_HIP(hipStreamSynchronize(c->stream));
            state = bad ? -1 : 1;
            if (getenv("NKML_TEST_LDS_DISORDER")) state = -1; // test aid: behave like a device that fails the check
        }
        c->lds_ordered = state > 0;
        if (state < 0 && c->partitioned) {
            if (cfg->flags & NKML_F_COUNT_PARTITIONED) {
                *out = c;
                return fail(c, NKML_E_NODEVICE, "this device does not serve returning LDS adds of one instruction in lane order: "
                                                "the partitioned counting paths cannot rank here (NKML_F_COUNT_ATOMIC selects the other path)");
            }
            // not asked for by flag: count on the atomic path, whose ranks come from a sort
            fprintf(stderr, "nkml: device %d does not serve returning LDS adds in lane order; counting on the atomic path\n", cfg->device);
            c->partitioned = c->hashpart = false;
            c->hp_leave_slots = 0;
        }
    }
    CREATE_CHK(alloc_table(c));
    CREATE_HIP(hipMalloc(&c->d_seq, npos + 64));
    CREATE_HIP(hipMalloc(&c->d_offsets, ((uint64_t)cfg->max_batch_reads + 1) * sizeof(uint64_t)));
    CREATE_HIP(hipMalloc(&c->d_packed, (c->max_words + 4) * sizeof(uint32_t)));
    CREATE_HIP(hipMalloc(&c->d_startbits, (c->max_words + 4) * sizeof(uint32_t)));
    CREATE_HIP(hipMalloc(&c->d_wflags, (c->max_words + 4) * sizeof(uint32_t)));
    CREATE_HIP(hipMalloc(&c->d_high, (uint64_t)cfg->max_batch_reads * sizeof(int32_t)));
    CREATE_HIP(hipMalloc(&c->d_total, (uint64_t)cfg->max_batch_reads * sizeof(int32_t)));
    if (!c->partitioned) CREATE_CHK(alloc_atomic_lists(c));
    CREATE_HIP(hipHostMalloc(&c->h_seq, npos + 64, hipHostMallocDefault));
    CREATE_HIP(hipHostMalloc(&c->h_offsets, ((uint64_t)cfg->max_batch_reads + 1) * sizeof(uint64_t), hipHostMallocDefault));
    CREATE_HIP(hipHostMalloc(&c->h_high, (uint64_t)cfg->max_batch_reads * sizeof(int32_t), hipHostMallocDefault));
    CREATE_HIP(hipHostMalloc(&c->h_total, (uint64_t)cfg->max_batch_reads * sizeof(int32_t), hipHostMallocDefault));
    CREATE_CHK(clear_table(c));
    CREATE_HIP(hipStreamSynchronize(c->stream));
#undef CREATE_CHK
#undef CREATE_HIP
    *out = c;
    return NKML_OK;
}

void nkml_destroy(nkml_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    hipFree(c->counts); hipFree(c->sat); hipFree(c->extra); hipFree(c->filter); hipFree(c->slots); hipFree(c->spare_slots);
    hipFree(c->hp_table); hipFree(c->hp_keysA); hipFree(c->hp_keysB); hipFree(c->hp_done);
    hipFree(c->pt_table); hipFree(c->pt_keysA); hipFree(c->pt_keysB); hipFree(c->pt_flagsA);
    hipFree(c->pt_flagsB); hipFree(c->pt_histA); hipFree(c->pt_scanA); hipFree(c->pt_histB); hipFree(c->pt_scanB); hipFree(c->pt_pad); hipFree(c->pt_bounds);
    hipFree(c->pt_tally); hipFree(c->pt_chunks); hipFree(c->pt_totals); hipFree(c->pt_startB); hipFree(c->pt_sposA); hipFree(c->pt_sposB);
    hipFree(c->d_seq); hipFree(c->d_offsets); hipFree(c->d_packed); hipFree(c->d_startbits);
    hipFree(c->d_wflags); hipFree(c->d_high); hipFree(c->d_total);
    hipFree(c->d_list_sat); hipFree(c->d_list_cross); hipFree(c->d_occ); hipFree(c->d_occ_sorted); hipFree(c->d_sort_tmp);
    hipFree(c->d_route_hist); hipFree(c->d_route_off); hipFree(c->d_route_tmp);
    hipFree(c->sat_bits); hipFree(c->d_sat_mask);
    hipFree(c->xb_order); hipFree(c->xb_place); hipFree(c->xb_kept); hipFree(c->xb_base); hipFree(c->xb_sizes); hipFree(c->xb_bounds2);
    hipFree(c->xb_run_at); hipFree(c->xb_keys2); hipFree(c->xb_flags2); hipFree(c->xb_scan_tmp);
    hipFree(c->sh_keys); hipFree(c->sh_own); hipFree(c->sh_flags); hipFree(c->sh_tile_count); hipFree(c->sh_tile_base); hipFree(c->sh_scan_tmp);
    hipFree(c->vf_before); hipFree(c->vf_keys);
    hipFree(c->d_ctr);
    if (c->h_ctr) hipHostFree(c->h_ctr);
    if (c->h_route_starts) hipHostFree(c->h_route_starts);
    if (c->h_seq) hipHostFree(c->h_seq);
    if (c->h_offsets) hipHostFree(c->h_offsets);
    if (c->h_high) hipHostFree(c->h_high);
    if (c->h_total) hipHostFree(c->h_total);
    for (int i = 0; i < 2 * 64; i++)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->ev_staged) hipEventDestroy(c->ev_staged);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int nkml_reset(nkml_ctx *c)
{
    if (!c) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = clear_table(c);
    if (rc != NKML_OK) return rc;
    memset(c->h_ctr, 0, sizeof(Counters));
    c->pending = false;
    c->pending_check = false;
    c->pt_timing_pending = false;
    c->staged_pending = false;
    c->used_bound = 0;
    c->hp_unique_before = 0;
    c->broken = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NKML_OK;
}

int nkml_set_stream(nkml_ctx *c, void *hip_stream)
{
    if (!c) return NKML_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) HIPCHK(c, hipStreamDestroy(c->stream));
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return NKML_OK;
}

int nkml_synchronize(nkml_ctx *c)
{
    if (!c) return NKML_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return collect_partition_timing(c);
}

int nkml_get_stats(nkml_ctx *c, nkml_stats *out)
{
    if (!c || !out) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = fetch_counters(c);
    if (rc != NKML_OK) return rc;
    if ((rc = collect_partition_timing(c)) != NKML_OK) return rc;
    memset(out, 0, sizeof *out);
    out->unique_kmers = c->h_ctr->unique;
    out->capacity = c->capacity;
    out->table_bytes = c->hashpart ? c->capacity * sizeof(unsigned long long) : c->partitioned ? c->capacity * (c->pt_wide ? 2 : 1) : c->direct ? c->capacity * 4 + c->capacity / 8 : c->capacity * sizeof(Slot);
    // (occurrences the saturation bitmap answered where they were read never reached a table: they count as saturated here)
    out->kmers_counted = c->h_ctr->counted + c->route_home_total;
    out->kmers_saturated = c->h_ctr->saturated + c->route_home_total;
    out->crossing_kmers = c->h_ctr->total_cross;
    out->crossing_occurrences = c->crossing_occ;
    out->resizes = c->resizes;
    out->count_kernel_ms = c->count_ms;
    out->count_kernel_launches = c->count_launches;
    out->count_kernel_keys = c->count_keys;
    out->count_stage_ms = c->pt_stage_ms;
    out->partitioned = c->partitioned ? 1 : 0;
    return NKML_OK;
}

int nkml_key_bytes(const nkml_ctx *c) { return c ? c->key_bytes : NKML_E_ARG; }

int nkml_kmerise_device(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, void *d_keys)
{
    if (!c || !d_seq || !d_offsets || !d_keys) return NKML_E_ARG;
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return run_kmerise(c, d_seq, d_offsets, n_reads, n_bases, d_keys);
}

int nkml_count_keys_device(nkml_ctx *c, const void *d_keys, uint64_t n_keys, uint8_t *d_flags)
{
    if (!c || !d_keys || !d_flags || (n_keys & 15)) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (int rc = hp_maybe_leave(c)) return rc;
    if (c->partitioned) return run_count_partitioned(c, d_keys, n_keys, 0, d_flags, false);
    return run_count(c, d_keys, n_keys, d_flags);
}

int nkml_seed_keys_device(nkml_ctx *c, const void *d_keys, uint64_t n_keys)
{
    if (!c || !d_keys || (n_keys & 15)) return NKML_E_ARG;
    if (n_keys == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc;
    if ((rc = hp_maybe_leave(c)) != NKML_OK) return rc;
    if (c->partitioned) {
        if ((rc = run_count_partitioned(c, d_keys, n_keys, 0, nullptr, true)) != NKML_OK) return rc;
        return fetch_counters(c);
    }
    if ((rc = ensure_capacity(c, n_keys, nullptr, nullptr)) != NKML_OK) return rc;
    if (c->direct)
        hipLaunchKernelGGL((k_seed_keys<uint32_t, true>), GRID_PERSIST, 256, 0, c->stream, (const uint32_t *)d_keys, n_keys, c->counts, c->slots, 0ull, c->d_ctr);
    else
        hipLaunchKernelGGL((k_seed_keys<unsigned long long, false>), GRID_PERSIST, 256, 0, c->stream, (const unsigned long long *)d_keys, n_keys,
                           c->counts, c->slots, c->capacity - 1, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    return finish_counting(c, 0, 0, nullptr, nullptr);
}

int nkml_reduce_reads_device(nkml_ctx *c, const void *d_keys, const uint8_t *d_flags, const uint64_t *d_offsets, uint32_t n_reads,
                             int32_t *d_high, int32_t *d_total)
{
    if (!c || !d_keys || !d_flags || !d_offsets || !d_high || !d_total) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return run_reduce(c, d_flags, d_offsets, n_reads, d_high, d_total);
}

// defer_check (partitioned path only): nothing here waits for the device; the caller fetches the counters and looks
// at the bad-base tally once it has waited for the stream (nkml_collect)
static int process_device(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases,
                          int32_t *d_high, int32_t *d_total, bool defer_check)
{
    if (!c || !d_seq || !d_offsets || !d_high || !d_total) return NKML_E_ARG;
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    if (n_reads == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if ((rc = hp_maybe_leave(c)) != NKML_OK) return rc;
    if (c->partitioned) {
        const uint64_t n_keys = (n_bases + 15) / 16 * 16;
        if ((rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
        if ((rc = run_count_partitioned(c, nullptr, n_keys, n_bases, nullptr, false)) != NKML_OK) return rc;
        if ((rc = run_reduce_words(c, d_offsets, n_reads, d_high, d_total)) != NKML_OK) return rc;
        if (defer_check) { c->pending_check = true; return NKML_OK; }
        if ((rc = fetch_counters(c)) != NKML_OK) return rc; // bad-base tally; the one wait of the batch
        if ((rc = collect_partition_timing(c)) != NKML_OK) return rc;
        return check_bad_bases(c);
    }
    if ((rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
    if ((rc = run_fused(c, n_bases, false)) != NKML_OK) return rc;
    if ((rc = check_bad_bases(c)) != NKML_OK) return rc;
    return run_reduce_words(c, d_offsets, n_reads, d_high, d_total);
}

int nkml_process_device(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases,
                        int32_t *d_high, int32_t *d_total)
{
    return process_device(c, d_seq, d_offsets, n_reads, n_bases, d_high, d_total, false);
}

int nkml_set_shard(nkml_ctx *c, uint32_t shard, uint32_t n_shards)
{
    if (!c) return NKML_E_ARG;
    if (n_shards == 0 || shard >= n_shards || n_shards > 65535) return fail(c, NKML_E_ARG, "shard %u of %u", shard, n_shards);
    if (n_shards > 1 && (!c->partitioned || c->hashpart)) return fail(c, NKML_E_STATE, "table shards need the partitioned counting path of the direct table (k <= 15)");
    c->n_shards = n_shards;
    c->shard = shard;
    const char *env = getenv("NKML_SHARD_COMPACT");
    if (n_shards > 1 && !c->sh_keys && !(env && env[0] == '0')) {
        // the shard's k-mers are pulled together before counting (run_count_shard): key and flag array for the worst case
        // (every k-mer of the batch in this shard), one count per tile of 256 packed words
        HIPCHK(c, hipSetDevice(c->cfg.device));
        const uint64_t npos = c->max_words * 16, tiles = (c->max_words + 255) / 256 + 1;
        HIPCHK(c, hipMalloc(&c->sh_keys, (npos + 16) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->sh_flags, npos + 16));
        HIPCHK(c, hipMalloc(&c->sh_own, (c->max_words + 256) * sizeof(uint16_t)));
        HIPCHK(c, hipMalloc(&c->sh_tile_count, tiles * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->sh_tile_base, tiles * sizeof(uint32_t)));
        size_t tmp = 0;
        HIPCHK(c, rocprim::exclusive_scan(nullptr, tmp, c->sh_tile_count, c->sh_tile_base, 0u, (size_t)tiles, rocprim::plus<uint32_t>(), c->stream));
        c->sh_scan_tmp_bytes = tmp ? tmp : 256;
        HIPCHK(c, hipMalloc(&c->sh_scan_tmp, c->sh_scan_tmp_bytes));
    }
    return NKML_OK;
}

int nkml_shard_of(uint64_t key, uint32_t n_shards)
{
    if (n_shards == 0 || n_shards > 65535 || key >> 32) return NKML_E_ARG;
    return (int)pt_shard_of((uint32_t)key, n_shards);
}

int nkml_pack_device(nkml_ctx *c, const uint8_t *d_seq, uint64_t n_bases, uint32_t *d_words)
{
    if (!c || !d_seq || !d_words) return NKML_E_ARG;
    if (n_bases == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_words = (n_bases + 15) / 16;
    hipLaunchKernelGGL(k_pack, blocks_for(n_words), 256, 0, c->stream, d_seq, n_bases, d_words, n_words, c->d_ctr, (uint32_t *)nullptr);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int nkml_process_packed_device(nkml_ctx *c, const uint32_t *d_words, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases,
                               int32_t *d_high, int32_t *d_total)
{
    if (!c || !d_words || !d_offsets || !d_high || !d_total) return NKML_E_ARG;
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    if (n_reads == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_words = (n_bases + 15) / 16;
    hipLaunchKernelGGL(k_take_words, blocks_for(n_words + 4), 256, 0, c->stream, d_words, n_words, c->d_packed, c->d_startbits);
    hipLaunchKernelGGL(k_mark_starts, blocks_for((uint64_t)n_reads + 1), 256, 0, c->stream, d_offsets, n_reads, c->d_startbits);
    HIPCHK(c, hipGetLastError());
    if ((rc = hp_maybe_leave(c)) != NKML_OK) return rc;
    if (c->partitioned) {
        if (c->n_shards > 1 && c->sh_keys) rc = run_count_shard(c, n_words, n_bases);
        else rc = run_count_partitioned(c, nullptr, n_words * 16, n_bases, nullptr, false);
        if (rc != NKML_OK) return rc;
        if ((rc = run_reduce_words(c, d_offsets, n_reads, d_high, d_total)) != NKML_OK) return rc;
        if ((rc = fetch_counters(c)) != NKML_OK) return rc; // the one wait of the batch
        if ((rc = collect_partition_timing(c)) != NKML_OK) return rc;
        return check_bad_bases(c);
    }
    if ((rc = run_fused(c, n_bases, false)) != NKML_OK) return rc;
    if ((rc = check_bad_bases(c)) != NKML_OK) return rc;
    return run_reduce_words(c, d_offsets, n_reads, d_high, d_total);
}

static int stage_host_batch(nkml_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads, uint64_t *n_bases_out)
{
    if (!c || !seq || !offsets) return NKML_E_ARG;
    if (offsets[0] != 0) return fail(c, NKML_E_ARG, "offsets[0] must be 0");
    uint64_t n_bases = offsets[n_reads];
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    const int k = c->cfg.ksize;
    for (uint32_t i = 0; i < n_reads; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] < (uint64_t)k)
            return fail(c, NKML_E_ARG, "read %u is shorter than k=%d (the caller drops such records)", i, k);
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (seq != c->h_seq) memcpy(c->h_seq, seq, n_bases);
    if (offsets != c->h_offsets) memcpy(c->h_offsets, offsets, ((uint64_t)n_reads + 1) * sizeof(uint64_t));
    HIPCHK(c, hipMemcpyAsync(c->d_seq, c->h_seq, n_bases, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_offsets, c->h_offsets, ((uint64_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_staged, c->stream));
    c->staged_pending = true;
    *n_bases_out = n_bases;
    return NKML_OK;
}

int nkml_wait_staged(nkml_ctx *c)
{
    if (!c) return NKML_E_ARG;
    if (c->staged_pending) {
        HIPCHK(c, hipEventSynchronize(c->ev_staged));
        c->staged_pending = false;
    }
    return NKML_OK;
}

int nkml_host_staging(nkml_ctx *c, uint8_t **seq, uint64_t **offsets)
{
    if (!c || !seq || !offsets) return NKML_E_ARG;
    *seq = c->h_seq;
    *offsets = c->h_offsets;
    return NKML_OK;
}

int nkml_submit(nkml_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads)
{
    if (!c) return NKML_E_ARG;
    if (c->pending) return fail(c, NKML_E_STATE, "previous batch not collected");
    if (n_reads == 0) { c->pending = true; c->pending_reads = 0; return NKML_OK; }
    uint64_t n_bases = 0;
    int rc = stage_host_batch(c, seq, offsets, n_reads, &n_bases);
    if (rc != NKML_OK) return rc;
    if ((rc = process_device(c, c->d_seq, c->d_offsets, n_reads, n_bases, c->d_high, c->d_total, true)) != NKML_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_high, c->d_high, (uint64_t)n_reads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, (uint64_t)n_reads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    c->pending = true;
    c->pending_reads = n_reads;
    return NKML_OK;
}

int nkml_collect(nkml_ctx *c, int32_t *high, int32_t *total)
{
    if (!c || !high || !total) return NKML_E_ARG;
    if (!c->pending) return fail(c, NKML_E_STATE, "nothing submitted");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pending = false;
    if (c->pending_check) { // the partitioned path left its end-of-batch checks to this wait
        c->pending_check = false;
        int rc = fetch_counters(c);
        if (rc != NKML_OK) return rc;
        if ((rc = collect_partition_timing(c)) != NKML_OK) return rc;
        if ((rc = check_bad_bases(c)) != NKML_OK) return rc;
    }
    memcpy(high, c->h_high, (uint64_t)c->pending_reads * sizeof(int32_t));
    memcpy(total, c->h_total, (uint64_t)c->pending_reads * sizeof(int32_t));
    return NKML_OK;
}

int nkml_seed(nkml_ctx *c, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads)
{
    if (!c) return NKML_E_ARG;
    if (n_reads == 0) return NKML_OK;
    uint64_t n_bases = 0;
    int rc = stage_host_batch(c, seq, offsets, n_reads, &n_bases);
    if (rc != NKML_OK) return rc;
    if ((rc = hp_maybe_leave(c)) != NKML_OK) return rc;
    if (c->partitioned) {
        if ((rc = run_prepare(c, c->d_seq, c->d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
        if ((rc = run_count_partitioned(c, nullptr, (n_bases + 15) / 16 * 16, n_bases, nullptr, true)) != NKML_OK) return rc;
        if ((rc = fetch_counters(c)) != NKML_OK) return rc;
        return check_bad_bases(c);
    }
    if ((rc = run_prepare(c, c->d_seq, c->d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
    if ((rc = run_fused(c, n_bases, true)) != NKML_OK) return rc;
    return check_bad_bases(c);
}

static inline float host_ratio(int32_t high, int32_t total)
{
    volatile float r = total > 0 ? (float)high / (float)total : 0.0f;
    return r;
}

int nkml_keep_pair(int32_t fwd_high, int32_t fwd_total, int32_t rev_high, int32_t rev_total, float coverage)
{
    return host_ratio(fwd_high, fwd_total) < coverage && host_ratio(rev_high, rev_total) < coverage;
}

int nkml_keep_single(int32_t high, int32_t total, float coverage) { return host_ratio(high, total) < coverage; }

int nkml_decide_device(nkml_ctx *c, const int32_t *d_high, const int32_t *d_total, uint32_t n_units, int paired, float coverage,
                       uint8_t *d_keep)
{
    if (!c || !d_high || !d_total || !d_keep) return NKML_E_ARG;
    if (n_units == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(k_decide, blocks_for(n_units), 256, 0, c->stream, d_high, d_total, n_units, paired, coverage, d_keep);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int nkml_owner_of(uint64_t key, uint32_t n_shards) { return n_shards ? (int)owner_of(key, n_shards) : NKML_E_ARG; }

// Stable partition of one batch's keys by owning shard.  d_keys != nullptr: from a key array; else from the packed
// words the context prepared (n_bases of them).
// hist / off: (tiles x shards + 1) entries each.  sums == nullptr: the scan is rocPRIM's, with the context's temporary
// storage; else the three-launch scan above with `sums` for its block totals (nothing that could allocate).
// meta != nullptr: ranked routing (see k_route_scatter), d_send_src unused; sat_bits != nullptr: with the saturation bitmap.
static int route_run(nkml_ctx *c, const void *d_keys, uint64_t n_keys, uint64_t n_bases, uint32_t n_shards, void *d_send_keys,
                     uint32_t *d_send_src, uint64_t *send_counts, uint32_t *hist, uint32_t *off, uint32_t *sums,
                     uint16_t *meta = nullptr, const uint32_t *sat_bits = nullptr, unsigned long long *sat_mask = nullptr)
{
    const uint64_t n_waves = (n_keys + ROUTE_TILE - 1) / ROUTE_TILE;
    const uint64_t entries = n_waves * n_shards;
    const uint32_t grid = blocks_for(n_waves, 4);
    HIPCHK(c, hipMemsetAsync(hist + entries, 0, sizeof(uint32_t), c->stream)); // one extra entry: the scan's last output = total
    RouteSource src = {d_keys, c->d_packed, c->d_startbits, n_bases, c->cfg.ksize};
    const int mode = d_keys ? 0 : c->cfg.canonical ? 2 : 1;
    // beyond two shards one LDS add per key ranks it (where the device serves such adds in lane order: checked at nkml_create)
    const int use_adds = n_shards > ROUTE_BALLOT_SHARDS && c->lds_ordered;
#define ROUTE_COUNT(KEY, SRC) hipLaunchKernelGGL((k_route_count<KEY, SRC>), grid, 256, 0, c->stream, src, n_keys, n_shards, n_waves, use_adds, hist, sat_bits, sat_mask, c->d_ctr)
#define ROUTE_SCATTER(KEY, SRC) hipLaunchKernelGGL((k_route_scatter<KEY, SRC>), grid, 256, 0, c->stream, src, n_keys, n_shards, n_waves, use_adds, off, \
                                                   (KEY *)d_send_keys, d_send_src, sat_bits ? sat_mask : nullptr, meta)
#define ROUTE_DISPATCH(WHAT)                                                                                      \
    do {                                                                                                          \
        if (c->key_bytes == 4) { if (mode == 0) WHAT(uint32_t, 0); else if (mode == 1) WHAT(uint32_t, 1); else WHAT(uint32_t, 2); } \
        else { if (mode == 0) WHAT(unsigned long long, 0); else if (mode == 1) WHAT(unsigned long long, 1); else WHAT(unsigned long long, 2); } \
    } while (0)
    ROUTE_DISPATCH(ROUTE_COUNT);
    HIPCHK(c, hipGetLastError());
    if (sums) {
        const uint32_t n = (uint32_t)(entries + 1), nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(k_scan_block_sums, nb, 256, 0, c->stream, hist, n, sums);
        hipLaunchKernelGGL(k_scan_sums, 1, 1024, 0, c->stream, sums, nb);
        hipLaunchKernelGGL(k_scan_apply, nb, 256, 0, c->stream, hist, n, sums, off);
        HIPCHK(c, hipGetLastError());
    } else {
        size_t tmp = c->route_tmp_bytes;
        HIPCHK(c, rocprim::exclusive_scan(c->d_route_tmp, tmp, hist, off, 0u, (size_t)(entries + 1), rocprim::plus<uint32_t>(), c->stream));
    }
    ROUTE_DISPATCH(ROUTE_SCATTER);
#undef ROUTE_DISPATCH
#undef ROUTE_SCATTER
#undef ROUTE_COUNT
    HIPCHK(c, hipGetLastError());
    // per-shard totals = differences of the shard start offsets (through pinned memory: no staging by the runtime)
    for (uint32_t s = 0; s <= n_shards; s++)
        HIPCHK(c, hipMemcpyAsync(&c->h_route_starts[s], off + (uint64_t)s * n_waves, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t s = 0; s < n_shards; s++) send_counts[s] = c->h_route_starts[s + 1] - c->h_route_starts[s];
    return NKML_OK;
}

static int route_common(nkml_ctx *c, const void *d_keys, uint64_t n_keys, uint64_t n_bases, uint32_t n_shards, void *d_send_keys,
                        uint32_t *d_send_src, uint64_t *send_counts)
{
    if (n_keys >= (1ull << 32)) return fail(c, NKML_E_ARG, "too many keys for 32-bit source positions");
    for (uint32_t s = 0; s < n_shards; s++) send_counts[s] = 0;
    if (n_keys == 0) return NKML_OK;
    const uint64_t n_waves = (n_keys + ROUTE_TILE - 1) / ROUTE_TILE;
    const uint64_t entries = n_waves * n_shards;
    if (entries >= (1ull << 31)) return fail(c, NKML_E_ARG, "routing table too large");
    if (entries + 1 > c->route_entries) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_route_hist); hipFree(c->d_route_off); hipFree(c->d_route_tmp);
        c->d_route_hist = c->d_route_off = nullptr; c->d_route_tmp = nullptr;
        c->route_entries = entries + 1;
        HIPCHK(c, hipMalloc(&c->d_route_hist, c->route_entries * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->d_route_off, c->route_entries * sizeof(uint32_t)));
        size_t tmp = 0;
        HIPCHK(c, rocprim::exclusive_scan(nullptr, tmp, c->d_route_hist, c->d_route_off, 0u, (size_t)c->route_entries, rocprim::plus<uint32_t>(), c->stream));
        c->route_tmp_bytes = tmp ? tmp : 256;
        HIPCHK(c, hipMalloc(&c->d_route_tmp, c->route_tmp_bytes));
    }
    return route_run(c, d_keys, n_keys, n_bases, n_shards, d_send_keys, d_send_src, send_counts, c->d_route_hist, c->d_route_off, nullptr);
}

int nkml_route_keys_device(nkml_ctx *c, const void *d_keys, uint64_t n_keys, uint32_t n_shards, void *d_send_keys,
                           uint32_t *d_send_src, uint64_t *send_counts)
{
    if (!c || !d_keys || !d_send_keys || !d_send_src || !send_counts || n_shards == 0 || n_shards > 64) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return route_common(c, d_keys, n_keys, 0, n_shards, d_send_keys, d_send_src, send_counts);
}

int nkml_route_keys_static(nkml_ctx *c, const void *d_keys, uint64_t n_keys, uint32_t n_shards, void *d_send_keys,
                           uint32_t *d_send_src, uint64_t *send_counts)
{
    if (!c || !d_keys || !d_send_keys || !d_send_src || !send_counts || n_shards == 0 || n_shards > 64) return NKML_E_ARG;
    if (n_keys >= (1ull << 32)) return fail(c, NKML_E_ARG, "too many keys for 32-bit source positions");
    const uint64_t n_waves = (n_keys + ROUTE_TILE - 1) / ROUTE_TILE, entries = n_waves * n_shards;
    if (entries > NKML_ROUTE_STATIC_ENTRIES)
        return fail(c, NKML_E_ARG, "nkml_route_keys_static: %llu keys for %u shards need %llu scratch entries, the fixed scratch holds %u "
                                   "(use nkml_route_keys_device, which allocates)",
                    (unsigned long long)n_keys, n_shards, (unsigned long long)entries, NKML_ROUTE_STATIC_ENTRIES);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    uint32_t *sp = c->cfg.device < MAX_DEVICES ? g_scratch_ptr[c->cfg.device] : nullptr;
    if (!sp) return fail(c, NKML_E_STATE, "no static scratch on device %d", c->cfg.device);
    for (uint32_t s = 0; s < n_shards; s++) send_counts[s] = 0;
    if (n_keys == 0) return NKML_OK;
    // every entry the kernels read is written first by this call (counts: one per (tile, shard); the extra entry and
    // the block sums: below), so what the last call left in the array does not matter
    uint32_t *hist = sp, *off = sp + (NKML_ROUTE_STATIC_ENTRIES + 64), *sums = sp + 2 * (NKML_ROUTE_STATIC_ENTRIES + 64);
    return route_run(c, d_keys, n_keys, 0, n_shards, d_send_keys, d_send_src, send_counts, hist, off, sums);
}

int nkml_route_reads_device(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t n_shards,
                            void *d_send_keys, uint32_t *d_send_src, uint64_t *send_counts)
{
    if (!c || !d_seq || !d_offsets || !d_send_keys || !d_send_src || !send_counts || n_shards == 0 || n_shards > 64) return NKML_E_ARG;
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    for (uint32_t s = 0; s < n_shards; s++) send_counts[s] = 0;
    if (n_reads == 0) return NKML_OK;
    if ((rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
    if ((rc = route_common(c, nullptr, (n_bases + 15) / 16 * 16, n_bases, n_shards, d_send_keys, d_send_src, send_counts)) != NKML_OK) return rc;
    if ((rc = fetch_counters(c)) != NKML_OK) return rc; // the stream has been waited for: the bad-base tally is final
    return check_bad_bases(c);
}

int nkml_unroute_flags_device(nkml_ctx *c, const uint8_t *d_recv_flags, const uint32_t *d_send_src, uint64_t n_sent, uint8_t *d_flags,
                              uint64_t n_keys)
{
    if (!c || !d_flags || (n_sent && (!d_recv_flags || !d_send_src))) return NKML_E_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemsetAsync(d_flags, 0, n_keys, c->stream));
    if (n_sent) {
        hipLaunchKernelGGL(k_unroute_flags, GRID_PERSIST, 256, 0, c->stream, d_recv_flags, d_send_src, n_sent, d_flags);
        HIPCHK(c, hipGetLastError());
    }
    return NKML_OK;
}

// ---- ranked routing and the saturation bitmap (exchange mode) -----------------------------------------------------------
// first place of rank r in the deal of the bucket-ordered exchange
static uint32_t xb_first(uint32_t buckets, uint32_t world, uint32_t r) { return (uint32_t)((uint64_t)buckets * r / world); }

int nkml_sat_enable(nkml_ctx *c)
{
    if (!c) return NKML_E_ARG;
    if (!c->partitioned || c->hashpart) return fail(c, NKML_E_STATE, "the saturation bitmap needs the partitioned direct table (k <= 15)");
    if (c->sat_bits) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t entries = c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity;
    HIPCHK(c, hipMalloc(&c->sat_bits, entries / 8));
    HIPCHK(c, hipMemsetAsync(c->sat_bits, 0, entries / 8, c->stream));
    return NKML_OK;
}

int nkml_route_reads_ranked(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, uint32_t n_shards,
                            void *d_send_keys, uint16_t *d_meta, uint32_t *d_tile_off, uint64_t *send_counts, uint64_t *n_home)
{
    if (!c || !d_seq || !d_offsets || !d_send_keys || !d_meta || !d_tile_off || !send_counts || n_shards == 0 || n_shards > ROUTE_RANKED_MAX_SHARDS)
        return NKML_E_ARG;
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    for (uint32_t s = 0; s < n_shards; s++) send_counts[s] = 0;
    if (n_home) *n_home = 0;
    if (n_reads == 0) return NKML_OK;
    const uint64_t n_keys = (n_bases + 15) / 16 * 16;
    if (n_keys >= (1ull << 32)) return fail(c, NKML_E_ARG, "too many keys for 32-bit positions");
    const uint64_t n_waves = (n_keys + ROUTE_TILE - 1) / ROUTE_TILE, entries = n_waves * n_shards;
    if (entries >= (1ull << 31)) return fail(c, NKML_E_ARG, "routing table too large");
    if ((rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
    if (entries + 1 > c->route_entries) { // the counts and the scan's temporary storage; the offsets go to the caller's d_tile_off
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_route_hist); hipFree(c->d_route_off); hipFree(c->d_route_tmp);
        c->d_route_hist = c->d_route_off = nullptr; c->d_route_tmp = nullptr;
        c->route_entries = entries + 1;
        HIPCHK(c, hipMalloc(&c->d_route_hist, c->route_entries * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(&c->d_route_off, c->route_entries * sizeof(uint32_t)));
        size_t tmp = 0;
        HIPCHK(c, rocprim::exclusive_scan(nullptr, tmp, c->d_route_hist, c->d_route_off, 0u, (size_t)c->route_entries, rocprim::plus<uint32_t>(), c->stream));
        c->route_tmp_bytes = tmp ? tmp : 256;
        HIPCHK(c, hipMalloc(&c->d_route_tmp, c->route_tmp_bytes));
    }
    if (c->sat_bits && n_waves * (ROUTE_TILE / 64) > c->sat_mask_words) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_sat_mask);
        c->d_sat_mask = nullptr;
        c->sat_mask_words = n_waves * (ROUTE_TILE / 64);
        HIPCHK(c, hipMalloc(&c->d_sat_mask, c->sat_mask_words * sizeof(unsigned long long)));
    }
    if (c->sat_bits) HIPCHK(c, hipMemsetAsync(&c->d_ctr->route_home, 0, sizeof(unsigned int), c->stream));
    if ((rc = route_run(c, nullptr, n_keys, n_bases, n_shards, d_send_keys, nullptr, send_counts, c->d_route_hist, d_tile_off, nullptr, d_meta,
                        c->sat_bits, c->d_sat_mask)) != NKML_OK)
        return rc;
    if ((rc = fetch_counters(c)) != NKML_OK) return rc; // the stream has been waited for: the tallies are final
    c->route_home_total += c->h_ctr->route_home;
    if (n_home) *n_home = c->sat_bits ? c->h_ctr->route_home : 0;
    return check_bad_bases(c);
}

int nkml_unroute_ranked(nkml_ctx *c, const uint8_t *d_back_flags, const uint16_t *d_meta, const uint32_t *d_tile_off, uint64_t n_positions,
                        uint8_t *d_flags)
{
    if (!c || !d_meta || !d_tile_off || !d_flags || (n_positions & 15u)) return NKML_E_ARG;
    if (n_positions == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_waves = (n_positions + ROUTE_TILE - 1) / ROUTE_TILE;
    hipLaunchKernelGGL(k_unroute_ranked, blocks_for((n_positions + 7) / 8), 256, 0, c->stream, d_back_flags, d_meta, d_tile_off, n_waves, n_positions, d_flags);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int nkml_sat_collect(nkml_ctx *c, uint32_t *d_codes, uint64_t max_codes, uint64_t *n_codes)
{
    if (!c || !n_codes || (max_codes && !d_codes) || max_codes >= (1ull << 32)) return NKML_E_ARG;
    if (!c->sat_bits) return fail(c, NKML_E_STATE, "nkml_sat_enable first");
    if (!c->pt_table) return fail(c, NKML_E_STATE, "the context was created without a table (NKML_F_NO_TABLE)");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t entries = c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity;
    HIPCHK(c, hipMemsetAsync(&c->d_ctr->sat_new, 0, sizeof(unsigned int), c->stream));
    // the buckets of the table this context counts into: all of them, or those it owns in the bucket-ordered exchange
    uint32_t grid = (uint32_t)(entries >> PT_LOWBITS), first = 0;
    if (c->xb_order) {
        first = xb_first(c->xb_buckets, c->xb_world, c->xb_rank);
        grid = xb_first(c->xb_buckets, c->xb_world, c->xb_rank + 1) - first;
    }
    const uint32_t n_buckets = grid;
    if (grid > 1024) grid = 1024; // four workgroups per CU, each with 32 KiB of LDS for its new k-mers, buckets in turns
    if (c->pt_wide)
        hipLaunchKernelGGL((k_sat_collect<uint16_t>), grid, 256, 0, c->stream, (const uint16_t *)c->pt_table, c->xb_order, first, n_buckets, (uint32_t)c->cfg.depth,
                           c->sat_bits, d_codes, (uint32_t)max_codes, c->d_ctr);
    else
        hipLaunchKernelGGL((k_sat_collect<uint8_t>), grid, 256, 0, c->stream, (const uint8_t *)c->pt_table, c->xb_order, first, n_buckets, (uint32_t)c->cfg.depth,
                           c->sat_bits, d_codes, (uint32_t)max_codes, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    if (int rc = fetch_counters(c)) return rc;
    // More than the list holds: the bits are set here, the list only feeds other ranks' bitmaps, which may lag for good -
    // what did not fit stays a travelling k-mer on the other ranks (results do not depend on it).
    *n_codes = c->h_ctr->sat_new < max_codes ? c->h_ctr->sat_new : max_codes;
    return NKML_OK;
}

int nkml_sat_apply(nkml_ctx *c, const uint32_t *d_codes, uint64_t n_codes)
{
    if (!c || (n_codes && !d_codes)) return NKML_E_ARG;
    if (!c->sat_bits) return fail(c, NKML_E_STATE, "nkml_sat_enable first");
    if (n_codes == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(k_sat_apply, blocks_for(n_codes), 256, 0, c->stream, d_codes, n_codes, c->sat_bits);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

// ---- bucket-ordered exchange (see nkml_bucketed.inc) ---------------------------------------------------------------------

int nkml_xb_enable(nkml_ctx *c, uint32_t world, uint32_t rank, uint64_t recv_capacity)
{
    if (!c || world == 0 || rank >= world) return NKML_E_ARG;
    if (!c->partitioned || c->hashpart) return fail(c, NKML_E_STATE, "the bucket-ordered exchange needs the partitioned direct table (k <= 15)");
    if (c->n_shards > 1) return fail(c, NKML_E_STATE, "the context counts a table shard of its own (nkml_set_shard)");
    if (c->xb_order) return fail(c, NKML_E_STATE, "nkml_xb_enable was called before");
    const uint32_t buckets = (uint32_t)((c->capacity >> PT_LOWBITS) ? (c->capacity >> PT_LOWBITS) : 1);
    if (buckets < world) return fail(c, NKML_E_ARG, "%u buckets cannot be dealt to %u ranks", buckets, world);
    if (recv_capacity >= (1ull << 32)) return fail(c, NKML_E_ARG, "more keys than 32-bit positions address");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    // the deal: buckets in the order of a mixed value of their number (the same on every rank)
    std::vector<uint32_t> order(buckets), place(buckets);
    for (uint32_t b = 0; b < buckets; b++) order[b] = b;
    auto mixed = [](uint32_t b) { // (a finaliser of its own: the deal must not change if a table's hash does)
        uint64_t x = (uint64_t)b + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t ma = mixed(a), mb = mixed(b);
        return ma != mb ? ma < mb : a < b;
    });
    for (uint32_t i = 0; i < buckets; i++) place[order[i]] = i;
    const uint32_t mine = xb_first(buckets, world, rank + 1) - xb_first(buckets, world, rank);
    HIPCHK(c, hipMalloc(&c->xb_order, buckets * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_place, buckets * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_kept, ((uint64_t)buckets + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_base, ((uint64_t)buckets + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_sizes, ((uint64_t)buckets + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_bounds2, ((uint64_t)buckets + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_run_at, ((uint64_t)world * mine + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->xb_keys2, (recv_capacity + 16) * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc(&c->xb_flags2, recv_capacity + 16));
    size_t t1 = 0, t2 = 0;
    HIPCHK(c, rocprim::exclusive_scan(nullptr, t1, c->xb_kept, c->xb_base, 0u, (size_t)buckets + 1, rocprim::plus<uint32_t>(), c->stream));
    HIPCHK(c, rocprim::exclusive_scan(nullptr, t2, c->xb_run_at, c->xb_run_at, 0u, (size_t)world * mine + 1, rocprim::plus<uint32_t>(), c->stream));
    c->xb_scan_tmp_bytes = (t1 > t2 ? t1 : t2) + 256;
    HIPCHK(c, hipMalloc(&c->xb_scan_tmp, c->xb_scan_tmp_bytes));
    HIPCHK(c, hipMemcpy(c->xb_order, order.data(), buckets * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->xb_place, place.data(), buckets * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->xb_world = world;
    c->xb_rank = rank;
    c->xb_buckets = buckets;
    c->xb_capacity = recv_capacity;
    return NKML_OK;
}

int nkml_xb_front(nkml_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, uint32_t n_reads, uint64_t n_bases, uint16_t *d_send_keys,
                  uint32_t *d_send_counts, uint64_t *send_totals, uint64_t *n_home)
{
    if (!c || (n_reads && (!d_seq || !d_offsets)) || !d_send_keys || !d_send_counts || !send_totals) return NKML_E_ARG;
    if (!c->xb_order) return fail(c, NKML_E_STATE, "nkml_xb_enable first");
    if (c->broken) return broken_context(c);
    int rc = check_batch(c, n_reads, n_bases);
    if (rc != NKML_OK) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint32_t W = c->xb_world, B = c->xb_buckets;
    for (uint32_t r = 0; r < W; r++) send_totals[r] = 0;
    if (n_home) *n_home = 0;
    c->xb_n = 0;
    if (n_reads == 0) {
        HIPCHK(c, hipMemsetAsync(d_send_counts, 0, (uint64_t)B * sizeof(uint32_t), c->stream));
        return NKML_OK;
    }
    const uint64_t n = (n_bases + 15) / 16 * 16;
    if (n > c->max_words * 16) return fail(c, NKML_E_ARG, "%llu positions exceed the context's batch size", (unsigned long long)n);
    if ((rc = run_prepare(c, d_seq, d_offsets, n_reads, n_bases)) != NKML_OK) return rc;
    const uint64_t ta = (n + PT_TILE - 1) / PT_TILE, tb = ta + 256 + 1;
    HIPCHK(c, hipMemsetAsync(c->pt_histA + 256 * ta, 0, 256 * sizeof(pt_hist_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->pt_histB + 256 * tb, 0, 256 * sizeof(pt_hist_t), c->stream));
    if ((rc = pt_forward(c, nullptr, n, n_bases, false)) != NKML_OK) return rc;
    if (c->verify) { // NKML_F_VERIFY_RANKS: both passes against a stable partition built without the LDS's lane order
        HIPCHK(c, hipMemsetAsync(&c->d_ctr->verify_bad, 0, sizeof(unsigned int), c->stream));
        VerifySource sa{nullptr, n, nullptr, c->d_packed, c->d_startbits, n_bases, n / 16, c->cfg.ksize, 0, 1, 0};
        verify_pass<uint32_t, uint32_t>(c, sa, (uint32_t)ta, PT_LOWBITS, c->pt_scanA, c->pt_keysA, 0xffffffffull);
        VerifySource sb{c->pt_keysA, 0, c->pt_pad + 256, nullptr, nullptr, 0, 0, c->cfg.ksize, 0, 1, 0};
        verify_pass<uint32_t, uint16_t>(c, sb, (uint32_t)tb, PT_LOWBITS + 8, c->pt_scanB, c->pt_keysB, (1ull << PT_LOWBITS) - 1);
        HIPCHK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_pt_bucket_bounds, B / 256 + 1, 256, 0, c->stream, c->pt_scanB, c->pt_pad, B, c->pt_bounds);
    HIPCHK(c, hipMemsetAsync(c->xb_kept + B, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_xb_mark, B, 64, 0, c->stream, c->pt_keysB, c->pt_bounds, c->sat_bits, c->xb_place, c->xb_kept);
    HIPCHK(c, hipGetLastError());
    size_t tmp = c->xb_scan_tmp_bytes;
    HIPCHK(c, rocprim::exclusive_scan(c->xb_scan_tmp, tmp, c->xb_kept, c->xb_base, 0u, (size_t)B + 1, rocprim::plus<uint32_t>(), c->stream));
    hipLaunchKernelGGL(k_xb_pack<true>, B, 64, 0, c->stream, c->pt_keysB, c->pt_bounds, c->pt_flagsB, c->sat_bits, c->xb_place, c->xb_base, d_send_keys, (const uint8_t *)nullptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(d_send_counts, c->xb_kept, (uint64_t)B * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    // every owner's total = differences of the scan at the first places of the ranks (through pinned memory)
    if (W + 2 > 70) return fail(c, NKML_E_ARG, "more ranks than the staging area of the totals holds");
    for (uint32_t r = 0; r <= W; r++)
        HIPCHK(c, hipMemcpyAsync(&c->h_route_starts[r], c->xb_base + xb_first(B, W, r), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&c->h_route_starts[W + 1], c->pt_bounds + B, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)); // k-mers of the batch
    if ((rc = fetch_counters(c)) != NKML_OK) return rc; // waits for the stream: the totals and the tallies are final
    for (uint32_t r = 0; r < W; r++) send_totals[r] = c->h_route_starts[r + 1] - c->h_route_starts[r];
    const uint64_t home = c->h_route_starts[W + 1] - (c->h_route_starts[W] - c->h_route_starts[0]); // what did not travel stayed home
    c->route_home_total += home;
    if (n_home) *n_home = home;
    c->xb_n = n;
    if (c->verify && c->h_ctr->verify_bad) // (the counters were fetched above, after the checks had run)
        return fail(c, NKML_E_STATE, "rank verification: %u elements of the batch are not where a stable partition puts them", c->h_ctr->verify_bad);
    return check_bad_bases(c);
}

int nkml_xb_count(nkml_ctx *c, const uint16_t *d_recv_keys, const uint32_t *d_recv_counts, uint64_t n_recv, uint8_t *d_recv_flags)
{
    if (!c || !d_recv_counts || (n_recv && (!d_recv_keys || !d_recv_flags))) return NKML_E_ARG;
    if (!c->xb_order) return fail(c, NKML_E_STATE, "nkml_xb_enable first");
    if (!c->pt_table) return fail(c, NKML_E_STATE, "the context was created without a table (NKML_F_NO_TABLE)");
    if (c->broken) return broken_context(c);
    if (n_recv > c->xb_capacity) return fail(c, NKML_E_ARG, "%llu keys arrived, the context was sized for %llu", (unsigned long long)n_recv, (unsigned long long)c->xb_capacity);
    if (n_recv == 0) return NKML_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint32_t W = c->xb_world, B = c->xb_buckets, first = xb_first(B, W, c->xb_rank), mine = xb_first(B, W, c->xb_rank + 1) - first;
    HIPCHK(c, hipMemsetAsync(c->xb_sizes, 0, ((uint64_t)B + 1) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_xb_sizes, blocks_for(mine), 256, 0, c->stream, d_recv_counts, W, mine, c->xb_order, first, c->xb_sizes);
    HIPCHK(c, hipGetLastError());
    size_t tmp = c->xb_scan_tmp_bytes;
    HIPCHK(c, rocprim::exclusive_scan(c->xb_scan_tmp, tmp, c->xb_sizes, c->xb_bounds2, 0u, (size_t)B + 1, rocprim::plus<uint32_t>(), c->stream));
    tmp = c->xb_scan_tmp_bytes; // (the last input entry of an exclusive scan does not reach the output: reading one past the counts is harmless only
    // if it is readable - the caller's array has world x mine entries, so the scan stops there and the runs' ends follow from the counts)
    HIPCHK(c, rocprim::exclusive_scan(c->xb_scan_tmp, tmp, d_recv_counts, c->xb_run_at, 0u, (size_t)W * mine, rocprim::plus<uint32_t>(), c->stream));
    hipLaunchKernelGGL(k_xb_merge<true>, mine * W, 64, 0, c->stream, d_recv_keys, (uint8_t *)nullptr, d_recv_counts, c->xb_run_at, W, mine, c->xb_order, first, c->xb_bounds2,
                       c->xb_keys2, (const uint8_t *)nullptr);
    HIPCHK(c, hipGetLastError());
    int rc;
    if (c->verify) { // NKML_F_VERIFY_RANKS: the bucket kernels' work against a copy of the table from before the step
        const uint64_t entries = c->capacity < (1ull << PT_LOWBITS) ? (1ull << PT_LOWBITS) : c->capacity;
        if ((rc = verify_snapshot(c, c->pt_table, entries * (c->pt_wide ? 2 : 1))) != NKML_OK) return rc;
    }
    c->broken = true; // from here on part of the step is in the table: a failure leaves the context unusable until nkml_reset
    if ((rc = pt_count_buckets(c, c->xb_keys2, c->xb_bounds2, c->xb_flags2, false, false)) != NKML_OK) return rc;
    c->broken = false;
    if (c->verify) {
        const size_t vf_lds = 3 * VF_WINDOW * sizeof(uint32_t);
        if (c->pt_wide)
            hipLaunchKernelGGL((k_verify_bucket<uint16_t>), B, 512, vf_lds, c->stream, c->xb_keys2, c->xb_bounds2, (const uint16_t *)c->pt_table,
                               (const uint16_t *)c->vf_before, (uint32_t)c->cfg.depth, c->xb_flags2, c->d_ctr);
        else
            hipLaunchKernelGGL((k_verify_bucket<uint8_t>), B, 512, vf_lds, c->stream, c->xb_keys2, c->xb_bounds2, (const uint8_t *)c->pt_table,
                               (const uint8_t *)c->vf_before, (uint32_t)c->cfg.depth, c->xb_flags2, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        if ((rc = verify_outcome(c)) != NKML_OK) return rc;
    }
    hipLaunchKernelGGL(k_xb_merge<false>, mine * W, 64, 0, c->stream, (const uint16_t *)nullptr, d_recv_flags, d_recv_counts, c->xb_run_at, W, mine, c->xb_order, first,
                       c->xb_bounds2, (uint16_t *)nullptr, c->xb_flags2);
    HIPCHK(c, hipGetLastError());
    return NKML_OK;
}

int nkml_xb_finish(nkml_ctx *c, const uint8_t *d_back_flags, const uint64_t *d_offsets, uint32_t n_reads, int32_t *d_high, int32_t *d_total)
{
    if (!c || (n_reads && (!d_offsets || !d_high || !d_total))) return NKML_E_ARG;
    if (!c->xb_order) return fail(c, NKML_E_STATE, "nkml_xb_enable first");
    if (n_reads == 0) return NKML_OK;
    if (c->xb_n == 0) return fail(c, NKML_E_STATE, "nkml_xb_finish without a batch from nkml_xb_front");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint32_t B = c->xb_buckets;
    hipLaunchKernelGGL(k_xb_pack<false>, B, 64, 0, c->stream, c->pt_keysB, c->pt_bounds, c->pt_flagsB, (const uint32_t *)nullptr, c->xb_place, c->xb_base, (uint16_t *)nullptr, d_back_flags);
    HIPCHK(c, hipGetLastError());
    int rc = pt_backward(c, c->xb_n, nullptr);
    if (rc != NKML_OK) return rc;
    c->xb_n = 0;
    return run_reduce_words(c, d_offsets, n_reads, d_high, d_total);
}

static int dump_table_into(nkml_ctx *c, unsigned long long *d_codes, uint32_t *d_cnts, unsigned long long *d_n, uint64_t *codes,
                           uint32_t *counts, uint64_t max_entries, uint64_t *n_entries)
{
    HIPCHK(c, hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream));
    if (c->hashpart) {
        hipLaunchKernelGGL(k_hp_dump, GRID_PERSIST, 256, 0, c->stream, c->hp_table, c->hp_log2_bucket, d_codes, d_cnts, max_entries, d_n);
    } else if (c->partitioned) {
        if (c->pt_wide)
            hipLaunchKernelGGL((k_pt_dump<uint16_t>), GRID_PERSIST, 256, 0, c->stream, (const uint16_t *)c->pt_table, c->capacity, d_codes, d_cnts, max_entries, d_n);
        else
            hipLaunchKernelGGL((k_pt_dump<uint8_t>), GRID_PERSIST, 256, 0, c->stream, (const uint8_t *)c->pt_table, c->capacity, d_codes, d_cnts, max_entries, d_n);
    } else if (c->direct) {
        hipLaunchKernelGGL(k_dump_direct, GRID_PERSIST, 256, 0, c->stream, c->counts, c->extra, c->capacity, d_codes, d_cnts, max_entries, d_n);
    } else {
        hipLaunchKernelGGL(k_dump_hashed, GRID_PERSIST, 256, 0, c->stream, c->slots, c->capacity, d_codes, d_cnts, max_entries, d_n);
    }
    HIPCHK(c, hipGetLastError());
    unsigned long long n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t take = n < max_entries ? n : max_entries;
    if (take) {
        HIPCHK(c, hipMemcpy(codes, d_codes, take * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(counts, d_cnts, take * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    *n_entries = n;
    return NKML_OK;
}

int nkml_dump_table(nkml_ctx *c, uint64_t *codes, uint32_t *counts, uint64_t max_entries, uint64_t *n_entries)
{
    if (!c || !n_entries || (max_entries && (!codes || !counts))) return NKML_E_ARG;
    if (c->partitioned && !c->hashpart && !c->pt_table) return fail(c, NKML_E_STATE, "the context was created without a table (NKML_F_NO_TABLE)");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long *d_codes = nullptr, *d_n = nullptr;
    uint32_t *d_cnts = nullptr;
    const uint64_t room = max_entries ? max_entries : 1;
    int rc = NKML_OK;
    hipError_t e;
    // the three device buffers are freed on every way out
    if ((e = hipMalloc(&d_codes, room * sizeof(unsigned long long))) != hipSuccess || (e = hipMalloc(&d_cnts, room * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc(&d_n, sizeof(unsigned long long))) != hipSuccess)
        rc = fail(c, e == hipErrorOutOfMemory ? NKML_E_NOMEM : NKML_E_HIP, "table dump: %s", hipGetErrorString(e));
    else
        rc = dump_table_into(c, d_codes, d_cnts, d_n, codes, counts, max_entries, n_entries);
    hipFree(d_codes); hipFree(d_cnts); hipFree(d_n);
    return rc;
}

} // extern "C"
