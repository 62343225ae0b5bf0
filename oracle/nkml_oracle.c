/*
 * CPU oracle for the k-mer coverage-normalisation hot path.  TEST INFRASTRUCTURE ONLY:
 * never linked into or called by the product library (see nkml_oracle.h).
 *
 * Each function names the lines of the reference program (normalise_kmers_multi_large.c,
 * "ref" below) whose behaviour it restates.  Parity status: pinned by
 * tests/test_oracle_golden.py against ref:56-70 and vectors captured from the compiled reference.
 */
#include "nkml_oracle.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * 2-bit code of one base, ref:150-153 (base_map): A=0 C=1 G=2 T=3, everything else 0.
 * 'N' never reaches the encoder in the reference because replacestr (ref:1406) rewrote it to 'A'.
 * ---------------------------------------------------------------------------------------- */
static inline int base_code(unsigned char c)
{
    switch (c) {
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default:  return 0; /* 'A', and 'N' after masking */
    }
}

/* valid_dna, ref:1144-1158, applied after N->A: only A, C, G, T pass. */
static inline int base_ok(unsigned char c)
{
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N';
}

/* encode_kmer_plain, ref:1118-1126: first base ends up in the most significant bits. */
uint64_t nkml_oracle_encode(const char *kmer, int k)
{
    uint64_t e = 0;
    for (int i = 0; i < k; i++)
        e = (e << 2) | (uint64_t)base_code((unsigned char)kmer[i]);
    return e;
}

/* get_canonical_kmer + reverse_complement, ref:1160-1180: the lexicographically smaller of the
 * k-mer and its reverse complement; A<C<G<T in ASCII equals the 2-bit order, so that is the
 * smaller code.  Equal strings (palindromes) give the same code either way. */
uint64_t nkml_oracle_encode_canonical(const char *kmer, int k)
{
    char rc[64];
    for (int i = 0; i < k; i++) {
        char c = kmer[i], d;
        switch (c) {
        case 'A': d = 'T'; break;
        case 'T': d = 'A'; break;
        case 'C': d = 'G'; break;
        case 'G': d = 'C'; break;
        default:  d = 0;   break; /* ref complement[] maps anything else to NUL */
        }
        rc[k - 1 - i] = d;
    }
    rc[k] = '\0';
    char km[64];
    memcpy(km, kmer, (size_t)k);
    km[k] = '\0';
    return nkml_oracle_encode(strcmp(km, rc) < 0 ? km : rc, k);
}

/* ------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t key; /* 0 = empty, ref:948 */
    int32_t count;
    int32_t pad;
} slot_t; /* 16 bytes like kmer_t, ref:156-161 */

struct nkml_oracle {
    int k, depth, canonical, model;
    uint64_t mask;    /* 2k low bits */
    uint64_t used;    /* hash_table_t.used, ref:166 */
    /* EXACT model, k <= 15: one counter per possible code (what the reference degenerates to when
     * capacity == 4^k); larger k: a conventional correct open-addressed map. */
    uint32_t *direct;
    slot_t *map;
    uint64_t map_cap; /* power of two */
    /* QUIRK model */
    slot_t *q;
    uint64_t q_cap;
};

static uint64_t pow4(int k) { return (uint64_t)1 << (2 * k); }

nkml_oracle *nkml_oracle_create(int k, int depth, int canonical, int model, uint64_t capacity)
{
    if (k < 1 || k > 31 || depth < 1)
        return NULL;
    nkml_oracle *o = calloc(1, sizeof *o);
    if (!o)
        return NULL;
    o->k = k;
    o->depth = depth;
    o->canonical = canonical ? 1 : 0;
    o->model = model;
    o->mask = pow4(k) - 1;
    if (model == NKML_ORACLE_QUIRK) {
        /* ref:676-684: default INITIAL_CAPACITY 67108879 (ref:137), clamped to 4^k */
        uint64_t cap = capacity ? capacity : 67108879ULL;
        if (pow4(k) < cap)
            cap = pow4(k);
        o->q_cap = cap;
        o->q = calloc(cap, sizeof(slot_t)); /* init_hash_table, ref:890-906 */
        if (!o->q) { free(o); return NULL; }
    } else if (k <= 15) {
        o->direct = calloc(pow4(k), sizeof(uint32_t));
        if (!o->direct) { free(o); return NULL; }
    } else {
        o->map_cap = (uint64_t)1 << 20;
        o->map = calloc(o->map_cap, sizeof(slot_t));
        if (!o->map) { free(o); return NULL; }
    }
    return o;
}

void nkml_oracle_destroy(nkml_oracle *o)
{
    if (!o)
        return;
    free(o->direct);
    free(o->map);
    free(o->q);
    free(o);
}

uint64_t nkml_oracle_unique(const nkml_oracle *o) { return o->used; }

/* --- EXACT model ------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void map_grow(nkml_oracle *o)
{
    uint64_t ncap = o->map_cap * 2;
    slot_t *n = calloc(ncap, sizeof(slot_t));
    if (!n)
        abort();
    for (uint64_t i = 0; i < o->map_cap; i++) {
        if (!o->map[i].key)
            continue;
        uint64_t j = mix64(o->map[i].key) & (ncap - 1);
        while (n[j].key)
            j = (j + 1) & (ncap - 1);
        n[j] = o->map[i];
    }
    free(o->map);
    o->map = n;
    o->map_cap = ncap;
}

/* returns pointer to the counter of `code`, inserting it (count 0) if absent */
static inline int32_t *exact_slot(nkml_oracle *o, uint64_t code)
{
    if (o->direct) {
        return (int32_t *)&o->direct[code];
    }
    if (o->used * 10 >= o->map_cap * 7)
        map_grow(o);
    uint64_t j = mix64(code) & (o->map_cap - 1);
    while (o->map[j].key && o->map[j].key != code)
        j = (j + 1) & (o->map_cap - 1);
    if (!o->map[j].key) {
        o->map[j].key = code;
        o->map[j].count = 0;
        o->map[j].pad = 0; /* pad = 1 once the key is counted in `used` */
    }
    return &o->map[j].count;
}

/* --- QUIRK model: expand_local_hash_table, ref:1055-1108 ---------------------------------- */
static void quirk_expand(nkml_oracle *o)
{
    uint64_t ncap = (uint64_t)((double)o->q_cap + (double)o->q_cap * 0.5); /* ref:1058 */
    if (o->q_cap >= ncap)
        return;
    slot_t *n = calloc(ncap, sizeof(slot_t));
    if (!n)
        abort();
    uint64_t nused = 0;
    for (uint64_t i = 0; i < o->q_cap; i++) {
        if (o->q[i].key != 0) {                    /* key-less ghost counters are dropped, ref:1079 */
            uint64_t j = o->q[i].key % ncap;        /* ref:1082 */
            while (n[j].key != 0)
                j = (j + 1) % ncap;                 /* linear here, ref:1083-1086 */
            n[j] = o->q[i];
            nused++;
        }
    }
    free(o->q);
    o->q = n;
    o->q_cap = ncap;
    o->used = nused; /* ref:1097 */
}

/* store_kmer, ref:929-1053.  Returns the slot index whose count the caller then reads (ref:1494). */
static uint64_t quirk_store(nkml_oracle *o, uint64_t h, int do_init)
{
    if ((double)o->used >= (double)o->q_cap * 0.8) /* ref:933, TABLE_LOAD_FACTOR ref:143 */
        quirk_expand(o);
    uint64_t i = h % o->q_cap; /* ref:936 */
    if (o->q[i].key == 0) {    /* ref:948-971 */
        o->q[i].key = h;
        o->q[i].count = do_init ? 0 : 1;
        o->used++;
        return i;
    }
    if (o->q[i].key == h) {    /* ref:972-1008 */
        if (!do_init)
            o->q[i].count++;
        return i;
    }
    /* ref:1009-1048: the step grows by collisions^2 cumulatively; every slot landed on has its count
     * touched whether or not it holds this key; an empty landing slot never receives the key. */
    int collisions = 0;
    while (o->q[i].key != 0 && o->q[i].key != h) {
        collisions++;
        i = (i + (uint64_t)((int64_t)collisions * collisions)) % o->q_cap; /* ref:1028 */
        if (!do_init)
            o->q[i].count++;   /* ref:1043-1044 */
        else
            o->q[i].count = 0; /* ref:1045-1046 */
    }
    return i;
}

uint32_t nkml_oracle_count_of(const nkml_oracle *o, uint64_t code)
{
    if (o->model == NKML_ORACLE_QUIRK) {
        uint64_t i = code % o->q_cap;
        return (o->q[i].key == code) ? (uint32_t)o->q[i].count : 0;
    }
    if (o->direct)
        return o->direct[code] & 0x7fffffffu;
    uint64_t j = mix64(code) & (o->map_cap - 1);
    while (o->map[j].key && o->map[j].key != code)
        j = (j + 1) & (o->map_cap - 1);
    return o->map[j].key ? (uint32_t)o->map[j].count : 0;
}

/* ------------------------------------------------------------------------------------------
 * The window loop, sequence_to_hash ref:1459-1499 (do_init = 0) and sequence_to_hash_zero
 * ref:1501-1537 (do_init = 1).  The reference re-encodes each window from its K characters;
 * the code below keeps a rolling forward code and a rolling reverse-complement code, which give the
 * same value for every window (checked against nkml_oracle_encode* by the tests).
 * ---------------------------------------------------------------------------------------- */
static void window_loop(nkml_oracle *o, const char *seq, int len, int do_init, int *high, int *total)
{
    const int k = o->k;
    uint64_t fwd = 0, rc = 0;
    const int rc_shift = 2 * (k - 1);
    for (int p = 0; p < len; p++) {
        uint64_t b = (uint64_t)base_code((unsigned char)seq[p]);
        fwd = ((fwd << 2) | b) & o->mask;
        rc = (rc >> 2) | ((3 - b) << rc_shift);
        if (p < k - 1)
            continue;
        uint64_t code = fwd;
        if (o->canonical && rc < fwd) /* ref:1472-1476 */
            code = rc;
        if (code == 0) /* ref:1483-1484 (and :1526-1527) */
            continue;
        if (o->model == NKML_ORACLE_QUIRK) {
            uint64_t idx = quirk_store(o, code, do_init);
            if (!do_init) {
                (*total)++; /* ref:1486 */
                if (o->q[idx].count >= o->depth) /* ref:1494-1497 */
                    (*high)++;
            }
        } else {
            int32_t *c = exact_slot(o, code);
            if (o->direct) {
                /* a direct counter has no separate "present" bit: presence = count > 0 or seeded.
                 * Seeded-but-uncounted keys are tracked in the top bit. */
                uint32_t v = (uint32_t)*c;
                if (v == 0)
                    o->used++; /* ref:965 */
                if (do_init) {
                    if (v == 0)
                        *c = (int32_t)0x80000000u; /* present, count 0 (ref:963 do_init) */
                } else {
                    uint32_t cnt = (v & 0x7fffffffu) + 1;
                    *c = (int32_t)(cnt | 0x80000000u);
                    (*total)++;
                    if ((int64_t)cnt >= o->depth)
                        (*high)++;
                }
            } else {
                slot_t *s = (slot_t *)((char *)c - offsetof(slot_t, count));
                if (!s->pad) {
                    s->pad = 1;
                    o->used++;
                }
                if (!do_init) {
                    (*c)++;
                    (*total)++;
                    if (*c >= o->depth)
                        (*high)++;
                }
            }
        }
    }
}

int nkml_oracle_keys(int k, int canonical, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads, uint64_t *keys)
{
    const uint64_t mask = pow4(k) - 1;
    const int rc_shift = 2 * (k - 1);
    for (uint32_t r = 0; r < n_reads; r++) {
        uint64_t a = offsets[r], b = offsets[r + 1];
        uint64_t fwd = 0, rc = 0;
        for (uint64_t p = a; p < b; p++) {
            if (!base_ok(seq[p]))
                return -1;
            keys[p] = 0;
            uint64_t c = (uint64_t)base_code(seq[p]);
            fwd = ((fwd << 2) | c) & mask;
            rc = (rc >> 2) | ((3 - c) << rc_shift);
            if (p + 1 < a + (uint64_t)k)
                continue;
            keys[p + 1 - (uint64_t)k] = (canonical && rc < fwd) ? rc : fwd;
        }
    }
    return 0;
}

void nkml_oracle_count_keys(nkml_oracle *o, const uint64_t *keys, uint64_t n, uint8_t *flags)
{
    for (uint64_t i = 0; i < n; i++) {
        flags[i] = 0;
        if (!keys[i])
            continue;
        int32_t *c = exact_slot(o, keys[i]);
        uint32_t cnt;
        if (o->direct) {
            uint32_t v = (uint32_t)*c;
            if (v == 0)
                o->used++;
            cnt = (v & 0x7fffffffu) + 1;
            *c = (int32_t)(cnt | 0x80000000u);
        } else {
            slot_t *s = (slot_t *)((char *)c - offsetof(slot_t, count));
            if (!s->pad) { s->pad = 1; o->used++; }
            cnt = (uint32_t)++(*c);
        }
        flags[i] = ((int64_t)cnt >= o->depth) ? 1 : 2;
    }
}

static int validate(const char *seq, int len)
{
    for (int i = 0; i < len; i++)
        if (!base_ok((unsigned char)seq[i]))
            return 0;
    return 1;
}

int nkml_oracle_read(nkml_oracle *o, const char *seq, int len, int *high, int *total)
{
    /* is_valid_sequence_single, ref:1404-1423: length test first, then the alphabet test */
    if (len < o->k)
        return 0;
    if (!validate(seq, len))
        return -1;
    *high = 0; /* ref:1545-1546 */
    *total = 0;
    window_loop(o, seq, len, 0, high, total);
    return 1;
}

int nkml_oracle_pair(nkml_oracle *o, const char *fwd, int fwd_len, const char *rev, int rev_len,
                     int *fwd_high, int *fwd_total, int *rev_high, int *rev_total)
{
    /* is_valid_sequence_pair, ref:1424-1457: both lengths, then both alphabets, before any counting */
    if (fwd_len < o->k || rev_len < o->k)
        return 0;
    if (!validate(fwd, fwd_len) || !validate(rev, rev_len))
        return -1;
    *fwd_high = *fwd_total = 0; /* ref:1557-1559 */
    window_loop(o, fwd, fwd_len, 0, fwd_high, fwd_total);
    *rev_high = *rev_total = 0; /* ref:1561-1563 */
    window_loop(o, rev, rev_len, 0, rev_high, rev_total);
    return 1;
}

int nkml_oracle_seed_read(nkml_oracle *o, const char *seq, int len)
{
    if (len <= o->k) /* strlen(record[1]) > K, ref:1347 */
        return 0;
    if (!validate(seq, len))
        return -1;
    int h = 0, t = 0;
    window_loop(o, seq, len, 1, &h, &t);
    return 1;
}

int nkml_oracle_batch(nkml_oracle *o, const uint8_t *seq, const uint64_t *offsets, uint32_t n_reads,
                      int32_t *high, int32_t *total)
{
    for (uint32_t r = 0; r < n_reads; r++) {
        int h = 0, t = 0;
        int len = (int)(offsets[r + 1] - offsets[r]);
        int rcode = nkml_oracle_read(o, (const char *)seq + offsets[r], len, &h, &t);
        if (rcode < 0)
            return -1;
        high[r] = h;
        total[r] = t;
    }
    return 0;
}

/* ref:1641-1646: ratio = total > 0 ? (float)high / total : 0 ; keep iff both ratios < coverage */
static inline float ratio_f32(int high, int total)
{
    volatile float r = total > 0 ? (float)high / (float)total : 0.0f;
    return r;
}

int nkml_oracle_keep_pair(int fwd_high, int fwd_total, int rev_high, int rev_total, float coverage)
{
    return ratio_f32(fwd_high, fwd_total) < coverage && ratio_f32(rev_high, rev_total) < coverage;
}

int nkml_oracle_keep_single(int high, int total, float coverage) /* ref:1988-1992 */
{
    return ratio_f32(high, total) < coverage;
}
