/*
 * nkml_amd - command-line host of the MI355X k-mer coverage normaliser.
 *
 * Drop-in for the reference program's command line (normalise_kmers_multi_large.c, "ref"): same options
 * (ref:543-565), same validation (ref:704-743), same input framing (4 lines per FASTQ record, 2 per FASTA
 * record, ref:1572), same output files in the current directory (ref:834-850, 2286-2297), same kept-record bytes
 * (sequence line with N rewritten to A, ref:1406, 1661-1665; FASTQ->FASTA header rule ref:852-876), same
 * counters in the per-file and final report lines (ref:1911, 2415-2419).  The per-read work
 * (ref:1551-1566 and everything below it) is done by the HIP library through the C ABI of nkml_hip.h:
 * records are framed from the mmap'd inputs, their sequence lines staged into one batch, the batch counted
 * on the GPU, and the float32 keep/skip test (ref:1641-1646) applied to the returned counters in input order.
 *
 * Differences, all deliberate:
 *   --cpu is accepted and ignored: there is one shared k-mer table on the device, so the depth is not divided
 *   by the thread count (ref:674) and there is a single output file pair named ..._thread0.fastq;
 *   counts are exact: that is what the reference computes only while its own table is collision-free (capacity == 4^k:
 *   k <= 13 always, -m 4 at k=14, -m 16 at k=15, ref:676-684).  With a smaller table - its default at k >= 14, any table
 *   at k > 16 - the reference's probe path (ref:1009-1047) perturbs counts and the reported unique k-mers, so outputs are
 *   byte-identical to it only where those collisions do not flip a decision (every fixture the reference ships; not
 *   guaranteed on large inputs: 0.07 % of 2 M synthetic pairs differ at k=15, 0.017 % at k=25).  A one-line notice on
 *   stderr says when the run is in that regime;
 *   lines longer than 1023 bytes are an error instead of being split (ref:139, 394-409).
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <getopt.h>
#include <locale.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "nkml_hip.h"

#define NKML_VERSION 20240823 /* the reference version this command line mirrors (ref:1) */
#define MAX_LINE 1023         /* longest line the reference can hold (MAX_LINE_LENGTH - 1, ref:139) */
#define MAX_CPUS 256          /* ref:142 */
#define SEED_RECORDS 3e6      /* ref:146 */
#define REF_DEFAULT_CAPACITY 67108879UL /* ref:137 */

struct config {
    char **fwd;
    int n_fwd;
    char **rev;
    int n_rev;
    int ksize, depth, cpus, verbose, debug, canonical, memory, single, doprint;
    float coverage;
    bool fastq_in, fastq_out;
};
static struct config cfg;

struct mapped {
    const char *data;
    size_t size;
};

struct totals {
    size_t processed, printed, skipped;
};

/* ---------------------------------------------------------------------------------------------------- */
static void usage(void)
{
    fprintf(stderr,
            "Usage:\n\n\t\tMandatory:\n"
            "\t\t* --forward|-f file1 [file2+]\tList of forward (read1) sequence files\n"
            "\t\t* --reverse|-r file1 [file2+]\tList of reverse (read2) sequence files\n"
            "\n\t\tOptional:\n"
            "\t\t[--single|-s] data are single ended, any --forward files not matched with --reverse are single-end\n"
            "\t\t[--ksize|-k (integer 5-31; def. 15)]\tk-mer size\n"
            "\t\t[--depth|-d (integer; def. 100)]\tcount at which a k-mer is tagged as high coverage\n"
            "\t\t[--coverage|-g (float 0-1; def. 0.9)]\tproportion of a read's k-mers that must be high coverage to drop it\n"
            "\t\t[--canonical|-c]\t\t\tmerge k-mers with their reverse complement\n"
            "\t\t[--filetype|-t (fq|fa; def. fq)]\tinput format\n"
            "\t\t[--outformat|-o (fq|fa; def. fq)]\toutput format\n"
            "\t\t[--memory_start|-m (integer; def. 1)]\tGb of k-mer table to start with (k > 16; k <= 16 uses 4^k counters)\n"
            "\t\t[--cpu|-p (int; def 1)]\t\t\taccepted for compatibility; the GPU path has one shared table\n"
            "\t\t[--verbose|-e] [--debug|-b N] [--print|-P] [--version|-v] [--help|-h]\n");
}

static void add_files(char ***list, int *count, char *first, char **more, int *optind_io)
{
    /* ref:747-832: the option's own argument, then every following argv entry up to the next one starting
     * with '-', each kept only if readable. */
    char *cand = first;
    for (;;) {
        if (access(cand, R_OK) == 0) {
            *list = realloc(*list, (size_t)(*count + 1) * sizeof(char *));
            if (!*list) { fprintf(stderr, "Memory allocation failed\n"); exit(1); }
            (*list)[(*count)++] = strdup(cand);
        } else {
            fprintf(stderr, "Warning: File '%s' does not exist or is not readable. Skipping.\n", cand);
        }
        if (*more == NULL || (*more)[0] == '-') break;
        cand = *more++;
        (*optind_io)++;
    }
}

static bool parse_format(const char *s, bool *is_fastq)
{
    if (!strcasecmp(s, "fa") || !strcasecmp(s, "fasta") || !strcasecmp(s, "fsa") || !strcasecmp(s, "fas")) { *is_fastq = false; return true; }
    if (!strcasecmp(s, "fq") || !strcasecmp(s, "fastq") || !strcasecmp(s, "fsq")) { *is_fastq = true; return true; }
    return false;
}

static size_t pow4(int k) { return (size_t)1 << (2 * k); }

/* initial_hash_size as the reference would report it (ref:416-422, 676-684); informational here */
static size_t reference_capacity(int memory_gb, int cpus, int k)
{
    size_t cap = REF_DEFAULT_CAPACITY;
    if (memory_gb > 0) {
        size_t number = (size_t)((float)memory_gb * 1073741824.0f);
        float total = (float)(number / 16);
        size_t per_cpu = (size_t)(total / (float)cpus);
        cap = (per_cpu % 2 == 0) ? per_cpu + 1 : per_cpu;
    }
    if (pow4(k) < cap) cap = pow4(k);
    return cap;
}

static int parse_arguments(int argc, char **argv)
{
    cfg.coverage = 0.9f;
    cfg.fastq_in = cfg.fastq_out = true;
    cfg.cpus = 1;
    cfg.ksize = 15;
    cfg.depth = 100;
    static struct option longopts[] = {
        {"forward", required_argument, 0, 'f'}, {"reverse", required_argument, 0, 'r'}, {"ksize", required_argument, 0, 'k'},
        {"depth", required_argument, 0, 'd'}, {"coverage", required_argument, 0, 'g'}, {"filetype", required_argument, 0, 't'},
        {"outformat", required_argument, 0, 'o'}, {"cpu", required_argument, 0, 'p'}, {"memory_start", required_argument, 0, 'm'},
        {"debug", required_argument, 0, 'b'}, {"verbose", no_argument, 0, 'e'}, {"help", no_argument, 0, 'h'},
        {"canonical", no_argument, 0, 'c'}, {"version", no_argument, 0, 'v'}, {"single", no_argument, 0, 's'},
        {"print", no_argument, 0, 'P'}, {0, 0, 0, 0}};
    int opt, idx = 0;
    while ((opt = getopt_long(argc, argv, "f:r:k:d:g:t:o:p:m:b:ehcvsP", longopts, &idx)) != -1) {
        switch (opt) {
        case 'P': cfg.doprint = 1; break;
        case 's': cfg.single = 1; break;
        case 'c': cfg.canonical = 1; break;
        case 'm':
            cfg.memory = atoi(optarg);
            if (cfg.memory < 1) { printf("Memory cannot be less than 1 Gb %'d\n", cfg.memory); return 0; }
            break;
        case 'b': cfg.debug = atoi(optarg); break;
        case 'h': usage(); exit(EXIT_SUCCESS);
        case 'p': cfg.cpus = atoi(optarg); break;
        case 'f': add_files(&cfg.fwd, &cfg.n_fwd, optarg, &argv[optind], &optind); break;
        case 'r': add_files(&cfg.rev, &cfg.n_rev, optarg, &argv[optind], &optind); break;
        case 'k': cfg.ksize = atoi(optarg); break;
        case 'd': cfg.depth = atoi(optarg); break;
        case 'g': cfg.coverage = (float)atof(optarg); break;
        case 'v': printf("%d\n", NKML_VERSION); exit(EXIT_SUCCESS);
        case 'e': cfg.verbose = 1; break;
        case 't':
            if (!parse_format(optarg, &cfg.fastq_in)) { printf("Input file format must be either fa or fq, not %s\n", optarg); return 0; }
            break;
        case 'o':
            if (!parse_format(optarg, &cfg.fastq_out)) { printf("Output file format must be either fa or fq, not %s\n", optarg); return 0; }
            break;
        default: fprintf(stderr, "Unexpected error in option processing\n"); return 0;
        }
    }
    if (cfg.verbose) {
        printf("\nVERSION: %d, CMD: ", NKML_VERSION);
        for (int i = 0; i < argc; i++) printf("%s ", argv[i]);
        printf("\n\n");
    }
    if (cfg.cpus > 0 && cfg.ksize >= 1 && cfg.ksize <= 31) {
        /* ref:676-685: the line reports the reference's own table sizing rule; the device table is sized by
         * nkml_create (4^k counters for k <= 16, --memory_start GiB of 16-byte slots otherwise) */
        size_t cap = cfg.memory > 0 ? reference_capacity(cfg.memory, cfg.cpus, 31) : REF_DEFAULT_CAPACITY;
        float mem_per_cpu = (float)cap * 16 / 1073741824;
        int total = cfg.memory;
        if (pow4(cfg.ksize) < cap) {
            cap = pow4(cfg.ksize);
            mem_per_cpu = (float)cap * 16 / 1073741824;
            total = (int)(mem_per_cpu * (float)cfg.cpus);
        }
        printf("Initial hash table size set to %'zu (maximum for k=%d is %'zu); memory ~ %'0.2f Gb for each of %d threads (~ %'d Gb total))\n\n",
               cap, cfg.ksize, pow4(cfg.ksize), mem_per_cpu, cfg.cpus, total);
    }
    if (cfg.cpus > 0 && cfg.ksize >= 5 && cfg.ksize <= 31 && reference_capacity(cfg.memory, cfg.cpus, cfg.ksize) < pow4(cfg.ksize))
        fprintf(stderr, "Note: k-mers are counted exactly here; the reference program with this --memory_start holds %'zu slots for 4^%d "
                        "possible k-mers, so its own counts (and, rarely, keep/skip decisions on large inputs) are perturbed by table collisions\n",
                reference_capacity(cfg.memory, cfg.cpus, cfg.ksize), cfg.ksize);
    /* ref:704-743 */
    if (cfg.n_fwd == 0 || (cfg.n_rev == 0 && !cfg.single)) {
        fprintf(stderr, "Error: no fwd (%d) or reverse (%d) files provided\n", cfg.n_fwd, cfg.n_rev);
        return 0;
    }
    if (!cfg.fastq_in && cfg.fastq_out) { fprintf(stderr, "Error: cannot request an output format of FASTQ when input is FASTA\n"); return 0; }
    if (!cfg.single && cfg.n_fwd != cfg.n_rev) {
        fprintf(stderr, "Error: Number of forward (%d) and reverse files (%d) must match\n", cfg.n_fwd, cfg.n_rev);
        return 0;
    }
    if (cfg.cpus <= 0 || cfg.cpus > MAX_CPUS) {
        fprintf(stderr, "Error: CPU count (%d) must be a positive integer and up to %d\n", cfg.cpus, MAX_CPUS);
        return 0;
    }
    if (cfg.ksize < 5 || cfg.ksize > 31) { fprintf(stderr, "Error: Only kmer sizes (%d) of 5 to 31 are supported\n", cfg.ksize); return 0; }
    if (cfg.coverage > 1 || cfg.coverage < 0.001) {
        fprintf(stderr, "Error: Coverage (%3.f) is the proportion of the sequence covered by high kmers and must be between 0 and 1\n", cfg.coverage);
        return 0;
    }
    if (cfg.depth < 2) {
        fprintf(stderr, "Error: Depth (%d) is the number of times a kmer needs to be found before being flagged as high coverage, it must be above 1\n", cfg.depth);
        return 0;
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------------- */
static struct mapped map_file(const char *name)
{
    struct mapped m = {NULL, 0};
    int fd = open(name, O_RDONLY);
    if (fd < 0) { perror("Error opening file"); return m; }
    struct stat sb;
    if (fstat(fd, &sb) < 0) { perror("Error getting file size"); close(fd); return m; }
    m.size = (size_t)sb.st_size;
    if (m.size == 0) { close(fd); m.data = ""; return m; }
    void *p = mmap(NULL, m.size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { perror("Error mapping file"); m.size = 0; return m; }
    madvise(p, m.size, MADV_SEQUENTIAL);
    m.data = p;
    return m;
}

static void unmap_file(struct mapped *m)
{
    if (m->data && m->size) munmap((void *)m->data, m->size);
    m->data = NULL;
    m->size = 0;
}

static const uint8_t base_class[256] = {['A'] = 1, ['C'] = 1, ['G'] = 1, ['T'] = 1, ['N'] = 2}; /* valid_dna after N->A, ref:1144-1158, 1406 */

/* Eight bytes at a time: every byte must be one of A C G T N.  (x ^ c) has a zero byte exactly where x holds c; the
 * zero-byte test used here has no carries between bytes, so a mark (0x80) stands for that byte alone. */
static inline uint64_t bytes_equal(uint64_t x, uint8_t c)
{
    const uint64_t v = x ^ (0x0101010101010101ULL * c), m = 0x7f7f7f7f7f7f7f7fULL;
    return ~(((v & m) + m) | v | m);
}
static bool valid_bases(const char *s, uint32_t n)
{
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, s + i, 8);
        const uint64_t hit = bytes_equal(x, 'A') | bytes_equal(x, 'C') | bytes_equal(x, 'G') | bytes_equal(x, 'T') | bytes_equal(x, 'N');
        if (hit != 0x8080808080808080ULL) return false;
    }
    uint8_t ok = 1;
    for (; i < n; i++) ok &= (uint8_t)(base_class[(uint8_t)s[i]] != 0);
    return ok;
}

static void fatal_sequence(const char *which, const char *s, uint32_t n)
{
    /* ref:1418, 1447, 1452 (the reference prints the N-masked line) */
    char buf[MAX_LINE + 1];
    for (uint32_t i = 0; i < n; i++) buf[i] = s[i] == 'N' ? 'A' : s[i];
    buf[n] = 0;
    fflush(NULL);
    fprintf(stderr, "FATAL: %s sequence does not appear to be a DNA sequence\n%s\n\n", which, buf);
}

/* ---------------------------------------------------------------------------------------------------- */
/* Host threads.  The record loop is a sequence of data-parallel phases over one window of the input files
 * (index the lines, size and stage the records, [GPU], format the kept records), each run on n_threads threads. */
#define MAX_THREADS 64
static int n_threads = 1;
typedef void (*par_fn)(int tid, int nt, void *arg);
struct par_call { par_fn fn; void *arg; int tid, nt; };
static void *par_tramp(void *p)
{
    struct par_call *c = p;
    c->fn(c->tid, c->nt, c->arg);
    return NULL;
}
static void parallel_run(par_fn fn, void *arg)
{
    pthread_t th[MAX_THREADS];
    struct par_call pc[MAX_THREADS];
    const int nt = n_threads;
    for (int t = 1; t < nt; t++) {
        pc[t] = (struct par_call){fn, arg, t, nt};
        if (pthread_create(&th[t], NULL, par_tramp, &pc[t]) != 0) { perror("pthread_create"); exit(EXIT_FAILURE); }
    }
    fn(0, nt, arg);
    for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
}

/* Line index of one window of a file: nl[g] = offset of the newline that ends line g. */
struct findex {
    const char *base;
    size_t size;
    uint32_t *nl;
    size_t n_nl, cap_nl;
    size_t n_lines; /* newline-terminated lines, plus the unterminated last line of the file (ref:394-409) */
};
static inline size_t line_start(const struct findex *fi, size_t g) { return g ? (size_t)fi->nl[g - 1] + 1 : 0; }
static inline size_t line_end(const struct findex *fi, size_t g) { return g < fi->n_nl ? (size_t)fi->nl[g] : fi->size; }

/* One scan per thread over its share of the bytes: newline offsets go straight into the thread's own stretch of nl[]
 * (room for one newline every `min_line` bytes would be wasteful: the stretch is sized from the density seen so far and
 * the scan falls back to counting if it runs out), then the stretches are closed up in place. */
struct index_part { size_t at, n, cap; bool overflow; };
static struct index_part ipart[MAX_THREADS];

static void index_scan(int tid, int nt, void *arg)
{
    struct findex *fi = arg;
    const char *p = fi->base + fi->size * (size_t)tid / (size_t)nt, *e = fi->base + fi->size * (size_t)(tid + 1) / (size_t)nt;
    struct index_part *ip = &ipart[tid];
    uint32_t *out = fi->nl + ip->at;
    size_t c = 0;
    while (p < e) {
        const char *q = memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        if (c < ip->cap) out[c] = (uint32_t)(q - fi->base);
        c++;
        p = q + 1;
    }
    ip->n = c;
    ip->overflow = c > ip->cap;
}
static void index_fill(int tid, int nt, void *arg)
{
    struct findex *fi = arg;
    const char *p = fi->base + fi->size * (size_t)tid / (size_t)nt, *e = fi->base + fi->size * (size_t)(tid + 1) / (size_t)nt;
    size_t at = ipart[tid].at;
    while (p < e) {
        const char *q = memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        fi->nl[at++] = (uint32_t)(q - fi->base);
        p = q + 1;
    }
}
static void build_index(struct findex *fi, const char *base, size_t size, bool at_eof)
{
    fi->base = base;
    fi->size = size;
    /* lines of FASTQ / FASTA records are rarely shorter than 20 bytes on average; the density of the last window refines it */
    static double bytes_per_line = 20.0;
    {
        const size_t per = (size_t)((double)(size / (size_t)n_threads + 1) / bytes_per_line * 1.1) + 64;
        if (per * (size_t)n_threads > fi->cap_nl) {
            free(fi->nl);
            fi->cap_nl = per * (size_t)n_threads;
            fi->nl = malloc(fi->cap_nl * sizeof(uint32_t));
            if (!fi->nl) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
        }
        for (int t = 0; t < n_threads; t++) ipart[t] = (struct index_part){(size_t)t * per, 0, per, false};
        parallel_run(index_scan, fi);
        size_t n = 0;
        bool overflow = false;
        for (int t = 0; t < n_threads; t++) { n += ipart[t].n; overflow |= ipart[t].overflow; }
        if (overflow) { /* denser than expected: the counts are known now, so the second scan fits exactly */
            if (n > fi->cap_nl) {
                free(fi->nl);
                fi->cap_nl = n + n / 8 + 1024;
                fi->nl = malloc(fi->cap_nl * sizeof(uint32_t));
                if (!fi->nl) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
            }
            size_t at = 0;
            for (int t = 0; t < n_threads; t++) { ipart[t].at = at; at += ipart[t].n; }
            parallel_run(index_fill, fi);
        } else { /* close the gaps between the threads' stretches (thread 0's is in place already) */
            size_t at = ipart[0].n;
            for (int t = 1; t < n_threads; t++) {
                memmove(fi->nl + at, fi->nl + ipart[t].at, ipart[t].n * sizeof(uint32_t));
                at += ipart[t].n;
            }
        }
        fi->n_nl = n;
        if (n > 1024) bytes_per_line = (double)size / (double)n;
    }
    fi->n_lines = fi->n_nl + (at_eof && size > 0 && base[size - 1] != '\n' ? 1 : 0);
}

/* ---------------------------------------------------------------------------------------------------- */
/* One batch in flight: the records of one window, their sequence lines staged in the library's pinned area. */
struct batch {
    uint8_t *seq;       /* pinned staging of the library (nkml_host_staging) */
    uint64_t *offsets;
    uint32_t cap_reads;
    uint64_t cap_bases;
};

/* One window of the input: its line index, its records and their results.  Two of them alternate, so that the kept
 * records of one window are formatted and written while the GPU counts the next one. */
struct window {
    struct findex F, R;
    size_t n_rec;       /* records taken from the window */
    uint8_t *use;       /* per record: long enough to be counted (ref:1630-1631) */
    size_t cap_use;
    uint32_t *rec;      /* per unit (pair or single read): its record number in the window */
    int32_t *high, *total;
    uint32_t n_reads, n_units;
    uint64_t n_bases;
};

struct plan {
    bool paired, to_fasta;
    int lines;
    uint32_t K;
    /* sizing pass, per thread */
    size_t t_units[MAX_THREADS], t_bases[MAX_THREADS], t_bad[MAX_THREADS];
    size_t t_seedable[2][MAX_THREADS], t_unused[MAX_THREADS]; /* records whose fwd / rev read is longer than k; records not counted */
    int t_bad_rev[MAX_THREADS], long_line;
    /* where the alphabet is checked: in the sizing pass (seeding pass: it has to know before it stages anything), or in the
     * staging pass on the bytes just copied (record loop: one pass over the sequence lines instead of two) */
    bool size_validates;
    size_t t_bad_unit[MAX_THREADS]; /* staging pass: units staged before the thread's first bad record */
    /* formatting pass, per thread; two sets of buffers take turns, so that one window's kept records can still be on
     * their way to the files while the next window's are formatted */
    int set;
    char *obuf[2][2][MAX_THREADS];
    size_t ocap[2][2][MAX_THREADS], olen[2][2][MAX_THREADS], t_printed[MAX_THREADS];
};

static struct batch B;
static struct plan P;
static struct window WIN[3], *W = &WIN[0]; /* W: the window the running phase works on (WIN[2]: spare for a seeding pass in mid-loop) */

/* NKML_TIMING=1: wall time of each phase of the host pipeline on stderr at exit */
static int timing_on;
static double t_phase[8];
static const char *t_name[8] = {"index", "size", "stage", "gpu", "format", "write", "seed", "setup"};
static inline double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define TIMED(slot, stmt) do { double t0_ = timing_on ? now_s() : 0; stmt; if (timing_on) t_phase[slot] += now_s() - t0_; } while (0)
static nkml_ctx *ctx;     /* one device ... */
static nkml_multi *mctx;  /* ... or one table over several (NKML_DEVICES): exactly one of the two is set */
static int out_fd[2] = {-1, -1};  /* forward / reverse output files */
static char *out_name[2];
/* Seeding of the first file (pair) folded into the record loop: a record that is going to be counted needs no seeding
 * (its k-mers enter the table anyway, and a key inserted with count 0 changes nothing else), so only the reads the
 * loop drops - a mate longer than k whose partner is too short - are seeded, when their window comes up.  The
 * reference's own seeding pass over that file (ref:2242-2250) is only run for real if the loop meets a record the
 * reference would have died on during seeding, to die the same way (before any output file exists). */
static bool lazy_seed;
static size_t lazy_limit, lazy_seen[2];      /* records with a read longer than k seen so far in the fwd / rev file */
static bool unlink_outputs_on_seed_fatal;
static void seed_file(const char *name, int limit);
static off_t out_off[2];          /* bytes written so far */
static struct totals tot;

static void die_nkml(const char *what, int rc)
{
    fflush(NULL);
    fprintf(stderr, "Error: %s: %s (%s)\n", what, nkml_strerror(rc), mctx ? nkml_multi_last_error(mctx) : ctx ? nkml_last_error(ctx) : "");
    exit(EXIT_FAILURE);
}

/* the engine behind the record loop: the single-device context or the multi-device one, same calls */
static int eng_host_staging(uint8_t **seq, uint64_t **off) { return mctx ? nkml_multi_host_staging(mctx, seq, off) : nkml_host_staging(ctx, seq, off); }
static int eng_wait_staged(void) { return mctx ? nkml_multi_wait_staged(mctx) : nkml_wait_staged(ctx); }
static int eng_submit(const uint8_t *seq, const uint64_t *off, uint32_t n) { return mctx ? nkml_multi_submit(mctx, seq, off, n) : nkml_submit(ctx, seq, off, n); }
static int eng_collect(int32_t *high, int32_t *total) { return mctx ? nkml_multi_collect(mctx, high, total) : nkml_collect(ctx, high, total); }
static int eng_seed(const uint8_t *seq, const uint64_t *off, uint32_t n) { return mctx ? nkml_multi_seed(mctx, seq, off, n) : nkml_seed(ctx, seq, off, n); }
static int eng_stats(nkml_stats *st) { return mctx ? nkml_multi_get_stats(mctx, st) : nkml_get_stats(ctx, st); }
static int eng_dump(uint64_t *codes, uint32_t *counts, uint64_t max, uint64_t *n)
{
    return mctx ? nkml_multi_dump_table(mctx, codes, counts, max, n) : nkml_dump_table(ctx, codes, counts, max, n);
}

static void batch_alloc(uint32_t cap_reads, uint64_t cap_bases)
{
    B.cap_reads = cap_reads;
    B.cap_bases = cap_bases;
    int rc = eng_host_staging(&B.seq, &B.offsets);
    if (rc != NKML_OK) die_nkml("staging", rc);
    for (int w = 0; w < 3; w++) {
        WIN[w].rec = malloc((size_t)cap_reads * sizeof(uint32_t));
        WIN[w].high = malloc((size_t)cap_reads * sizeof(int32_t));
        WIN[w].total = malloc((size_t)cap_reads * sizeof(int32_t));
        if (!WIN[w].rec || !WIN[w].high || !WIN[w].total) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
    }
    B.offsets[0] = 0;
}

static inline float ratio_f32(int32_t high, int32_t total)
{
    volatile float r = total > 0 ? (float)high / (float)total : 0.0f; /* ref:1641 */
    return r;
}

static inline void rec_range(size_t n, int tid, int nt, size_t *a, size_t *b)
{
    *a = n * (size_t)tid / (size_t)nt;
    *b = n * (size_t)(tid + 1) / (size_t)nt;
}

/* sequence line (line 1) of record i */
static inline const char *seq_line(const struct findex *fi, int lines, size_t i, uint32_t *len)
{
    const size_t g = i * (size_t)lines + 1, a = line_start(fi, g);
    *len = (uint32_t)(line_end(fi, g) - a);
    return fi->base + a;
}

/* Phase 1: is_valid_sequence_pair / _single (ref:1404-1457) for every record: lengths first, then the alphabet.
 * Each thread stops at its first record with a bad base; records after the first such record of the window are
 * not processed at all (the reference exits there). */
static void size_records(int tid, int nt, void *arg)
{
    (void)arg;
    size_t a, b;
    rec_range(W->n_rec, tid, nt, &a, &b);
    size_t units = 0, bases = 0, seedable_f = 0, seedable_r = 0, unused = 0;
    P.t_bad[tid] = SIZE_MAX;
    P.t_seedable[0][tid] = P.t_seedable[1][tid] = P.t_unused[tid] = 0;
    for (size_t i = a; i < b; i++) {
        for (int l = 0; l < P.lines; l++) {
            const size_t g = i * (size_t)P.lines + (size_t)l;
            if (line_end(&W->F, g) - line_start(&W->F, g) > MAX_LINE) P.long_line = 1;
            if (P.paired && line_end(&W->R, g) - line_start(&W->R, g) > MAX_LINE) P.long_line = 1;
        }
        uint32_t lf, lr = 0;
        const char *sf = seq_line(&W->F, P.lines, i, &lf);
        const char *sr = P.paired ? seq_line(&W->R, P.lines, i, &lr) : NULL;
        const bool long_enough = lf >= P.K && (!P.paired || lr >= P.K);
        W->use[i] = long_enough;
        seedable_f += lf > P.K;
        seedable_r += P.paired && lr > P.K;
        if (!long_enough) { unused++; continue; } /* dropped without being counted anywhere, ref:1630-1631 */
        if (lf > MAX_LINE || lr > MAX_LINE) continue; /* reported after the phase */
        if (P.size_validates) {
            if (!valid_bases(sf, lf)) { P.t_bad[tid] = i; P.t_bad_rev[tid] = 0; break; }
            if (P.paired && !valid_bases(sr, lr)) { P.t_bad[tid] = i; P.t_bad_rev[tid] = 1; break; }
        }
        units++;
        bases += (size_t)lf + lr;
    }
    P.t_units[tid] = units;
    P.t_bases[tid] = bases;
    P.t_seedable[0][tid] = seedable_f;
    P.t_seedable[1][tid] = seedable_r;
    P.t_unused[tid] = unused;
}

/* Phase 2: sequence lines into the staging area, in record order; unless the sizing pass has done it, the alphabet check
 * of is_valid_sequence_pair / _single (ref:1404-1457) on the bytes just copied.  A thread stops at its first record with a
 * bad base (the reference exits there; what other threads stage beyond it is never submitted). */
static void stage_records(int tid, int nt, void *arg)
{
    (void)arg;
    size_t a, b;
    rec_range(W->n_rec, tid, nt, &a, &b);
    size_t u = 0, at = 0;
    for (int t = 0; t < tid; t++) { u += P.t_units[t]; at += P.t_bases[t]; }
    const size_t u_end = u + P.t_units[tid];
    const bool validate = !P.size_validates;
    if (validate) P.t_bad[tid] = SIZE_MAX;
    for (size_t i = a; i < b && u < u_end; i++) {
        if (!W->use[i]) continue;
        uint32_t lf, lr = 0;
        const char *sf = seq_line(&W->F, P.lines, i, &lf);
        W->rec[u] = (uint32_t)i;
        if (P.paired) {
            const char *sr = seq_line(&W->R, P.lines, i, &lr);
            memcpy(B.seq + at, sf, lf);
            B.offsets[2 * u + 1] = at + lf;
            memcpy(B.seq + at + lf, sr, lr);
            B.offsets[2 * u + 2] = at + lf + lr;
        } else {
            memcpy(B.seq + at, sf, lf);
            B.offsets[u + 1] = at + lf;
        }
        if (validate) {
            const char *copy = (const char *)B.seq + at;
            const int bad_read = !valid_bases(copy, lf) ? 0 : (P.paired && !valid_bases(copy + lf, lr)) ? 1 : -1;
            if (bad_read >= 0) { P.t_bad[tid] = i; P.t_bad_rev[tid] = bad_read; P.t_bad_unit[tid] = u; break; }
        }
        at += (size_t)lf + lr;
        u++;
    }
}

static inline char *put_masked(char *o, const char *s, size_t n)
{
    memcpy(o, s, n);
    if (memchr(s, 'N', n))
        for (size_t i = 0; i < n; i++)
            if (o[i] == 'N') o[i] = 'A';
    o[n] = '\n';
    return o + n + 1;
}

/* Kept record, ref:1649-1666: the stored lines with the sequence line N-masked, or the FASTQ->FASTA form
 * (ref:852-876: '>' header, "/1" or "/2" appended unless the header already ends with it). */
static char *put_record(char *o, const struct findex *fi, size_t i, int lines, bool to_fasta, bool is_forward)
{
    const size_t g0 = i * (size_t)lines;
    if (to_fasta) {
        const char *h = fi->base + line_start(fi, g0);
        const size_t hl = line_end(fi, g0) - line_start(fi, g0);
        const char *suffix = is_forward ? "/1" : "/2";
        *o++ = '>';
        if (hl > 1) { memcpy(o, h + 1, hl - 1); o += hl - 1; }
        if (hl < 2 || memcmp(h + hl - 2, suffix, 2) != 0) { memcpy(o, suffix, 2); o += 2; }
        *o++ = '\n';
        return put_masked(o, fi->base + line_start(fi, g0 + 1), line_end(fi, g0 + 1) - line_start(fi, g0 + 1));
    }
    /* the stored lines, each followed by a newline: in the input they lie back to back with their newlines, so the
     * record is one copy (plus the newline an unterminated last line of the file lacks), and the sequence line is
     * N-masked where it landed */
    const size_t a = line_start(fi, g0), e = line_end(fi, g0 + (size_t)lines - 1);
    const size_t sa = line_start(fi, g0 + 1) - a, sn = line_end(fi, g0 + 1) - line_start(fi, g0 + 1);
    memcpy(o, fi->base + a, e - a);
    if (memchr(o + sa, 'N', sn))
        for (size_t j = 0; j < sn; j++)
            if (o[sa + j] == 'N') o[sa + j] = 'A';
    o[e - a] = '\n';
    return o + (e - a) + 1;
}

static inline bool keep_unit(uint32_t u)
{
    if (P.paired) return nkml_keep_pair(W->high[2 * u], W->total[2 * u], W->high[2 * u + 1], W->total[2 * u + 1], cfg.coverage) != 0; /* ref:1646 */
    return nkml_keep_single(W->high[u], W->total[u], cfg.coverage) != 0;                                                             /* ref:1992 */
}

/* Phase 3: decisions and the bytes of the kept records, per thread into its own buffer (written out in thread order) */
static void format_records(int tid, int nt, void *arg)
{
    (void)arg;
    size_t a, b;
    rec_range(W->n_units, tid, nt, &a, &b);
    const int set = P.set;
    P.olen[set][0][tid] = P.olen[set][1][tid] = 0;
    P.t_printed[tid] = 0;
    if (a == b) return;
    /* ref:1995-1999: the single-end FASTQ->FASTA branch builds the record and never writes it */
    const bool write_f = P.paired || !P.to_fasta, write_r = P.paired;
    for (int f = 0; f < 2; f++) {
        if (!(f == 0 ? write_f : write_r)) continue;
        const struct findex *fi = f == 0 ? &W->F : &W->R;
        const size_t first = (size_t)W->rec[a] * (size_t)P.lines, last = (size_t)W->rec[b - 1] * (size_t)P.lines + (size_t)P.lines - 1;
        const size_t need = line_end(fi, last) - line_start(fi, first) + 4 * (b - a) + 64;
        if (need > P.ocap[set][f][tid]) {
            free(P.obuf[set][f][tid]);
            P.ocap[set][f][tid] = need + need / 4;
            P.obuf[set][f][tid] = malloc(P.ocap[set][f][tid]);
            if (!P.obuf[set][f][tid]) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
        }
    }
    char *of = P.obuf[set][0][tid], *orv = P.obuf[set][1][tid];
    size_t printed = 0;
    for (size_t u = a; u < b; u++) {
        if (!keep_unit((uint32_t)u)) continue;
        printed++;
        if (write_f) of = put_record(of, &W->F, W->rec[u], P.lines, P.to_fasta, true);
        if (write_r) orv = put_record(orv, &W->R, W->rec[u], P.lines, P.to_fasta, false);
    }
    P.olen[set][0][tid] = write_f ? (size_t)(of - P.obuf[set][0][tid]) : 0;
    P.olen[set][1][tid] = write_r ? (size_t)(orv - P.obuf[set][1][tid]) : 0;
    P.t_printed[tid] = printed;
}

/* Phase 4: the buffers go to the output files, in thread order (= input order), by one writer thread per file.  Buffered
 * writes to one file take the file's lock one after the other whatever the number of threads that issue them, so one
 * thread per file does as well as sixteen - and as a thread of its own it works while the record loop is already
 * indexing, sizing and staging the next window. */
struct writer {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int f;              /* 0 forward, 1 reverse */
    int job_set;        /* set of buffers waiting to be written, -1 none */
    int busy_set;       /* set being written, -1 none */
    off_t job_off;      /* file offset of the waiting job */
    int job_threads;
    bool quit, started, inited;
};
static struct writer WR[2];

static void *writer_main(void *arg)
{
    struct writer *w = arg;
    pthread_mutex_lock(&w->mu);
    for (;;) {
        while (w->job_set < 0 && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
        if (w->job_set < 0) break;
        const int set = w->job_set, nt = w->job_threads;
        off_t at = w->job_off;
        w->busy_set = set;
        w->job_set = -1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        for (int t = 0; t < nt; t++) {
            size_t n = P.olen[set][w->f][t];
            const char *p = P.obuf[set][w->f][t];
            while (n) {
                ssize_t k = pwrite(out_fd[w->f], p, n, at);
                if (k < 0) { perror("Error writing output"); exit(EXIT_FAILURE); }
                p += k; n -= (size_t)k; at += k;
            }
        }
        pthread_mutex_lock(&w->mu);
        w->busy_set = -1;
        pthread_cond_broadcast(&w->cv);
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

static void writers_start(void)
{
    for (int f = 0; f < 2; f++) {
        struct writer *w = &WR[f];
        if (out_fd[f] < 0 || w->started) continue;
        if (!w->inited) {
            pthread_mutex_init(&w->mu, NULL);
            pthread_cond_init(&w->cv, NULL);
            w->inited = true;
        }
        w->f = f;
        w->job_set = w->busy_set = -1;
        w->quit = false;
        if (pthread_create(&w->th, NULL, writer_main, w) != 0) { perror("pthread_create"); exit(EXIT_FAILURE); }
        w->started = true;
    }
}

/* until neither writer holds buffers of `set` any more (set < 0: any set) */
static void writers_wait(int set)
{
    for (int f = 0; f < 2; f++) {
        struct writer *w = &WR[f];
        if (!w->started) continue;
        pthread_mutex_lock(&w->mu);
        while ((w->job_set >= 0 && (set < 0 || w->job_set == set)) || (w->busy_set >= 0 && (set < 0 || w->busy_set == set)))
            pthread_cond_wait(&w->cv, &w->mu);
        pthread_mutex_unlock(&w->mu);
    }
}

static void writers_post(int set)
{
    for (int f = 0; f < 2; f++) {
        struct writer *w = &WR[f];
        size_t total = 0;
        for (int t = 0; t < n_threads; t++) total += P.olen[set][f][t];
        if (!w->started || total == 0) continue;
        pthread_mutex_lock(&w->mu);
        while (w->job_set >= 0) pthread_cond_wait(&w->cv, &w->mu); /* the mailbox holds one job */
        w->job_set = set;
        w->job_off = out_off[f];
        w->job_threads = n_threads;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        out_off[f] += (off_t)total;
    }
}

static void writers_stop(void)
{
    writers_wait(-1);
    for (int f = 0; f < 2; f++) {
        struct writer *w = &WR[f];
        if (!w->started) continue;
        pthread_mutex_lock(&w->mu);
        w->quit = true;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pthread_join(w->th, NULL);
        w->started = false;
    }
}

/* The staged window goes to the GPU; on the partitioned counting path the call returns at once. */
static void submit_window(void)
{
    if (W->n_units == 0) return;
    int rc = 0;
    TIMED(3, {
        rc = eng_submit(B.seq, B.offsets, W->n_reads);
        if (rc != NKML_OK) die_nkml("submit", rc);
    });
}

/* Wait for the window's counters, then decide and write in input order. */
static void finish_window(void)
{
    if (W->n_units == 0) return;
    int rc = 0;
    TIMED(3, {
        rc = eng_collect(W->high, W->total);
        if (rc != NKML_OK) die_nkml("collect", rc);
    });
}

static void write_window(void)
{
    if (W->n_units == 0) return;
    TIMED(5, writers_wait(P.set)); /* the window before the last one has left these buffers */
    TIMED(4, parallel_run(format_records, NULL));
    writers_post(P.set);
    P.set ^= 1;
    if (cfg.debug > 1) {
        for (uint32_t u = 0; u < W->n_units; u++) {
            const bool keep = keep_unit(u);
            if (P.paired) { /* ref:1677-1692 */
                int32_t hf = W->high[2 * u], tf = W->total[2 * u], hr = W->high[2 * u + 1], tr = W->total[2 * u + 1];
                printf("Thread 0 - Sequence pair %'zu %s: High (%d) count kmers: F:%d;R:%d, Total kmers: F:%d;R:%d High count ratio: F:%.2f;R:%.2f\n",
                       tot.processed + u + 1, keep ? "PRINTED" : "SKIPPED", cfg.depth, hf, hr, tf, tr, ratio_f32(hf, tf), ratio_f32(hr, tr));
            } else { /* ref:2017-2032 */
                int32_t hf = W->high[u], tf = W->total[u];
                printf("Thread 0 - Sequence pair %'zu %s: High (%d) count kmers: F:%d, Total kmers: F:%d High count ratio: F:%.2f\n",
                       tot.processed + u + 1, keep ? "PRINTED" : "SKIPPED", cfg.depth, hf, tf, ratio_f32(hf, tf));
            }
        }
    }
    size_t printed = 0;
    for (int t = 0; t < n_threads; t++) printed += P.t_printed[t];
    tot.processed += W->n_units;
    tot.printed += printed;
    tot.skipped += W->n_units - printed;
    W->n_reads = W->n_units = 0;
    W->n_bases = 0;
}

/* Index and size the next window of the file (pair) into *W.  Returns false when no record (pair) is left.
 * *bad / *bad_rev: the record at which the run has to stop with the reference's fatal message (SIZE_MAX: none). */
static bool size_window(const struct mapped *fwd, const struct mapped *rev, size_t *pos_f, size_t *pos_r, size_t *bytes_per_rec,
                        size_t *bad, int *bad_rev)
{
    const size_t cap_units = P.paired ? B.cap_reads / 2 : B.cap_reads;
    size_t want = cap_units * *bytes_per_rec;
    if (want > (size_t)3 << 30) want = (size_t)3 << 30; /* line offsets are 32-bit */
    size_t wf = fwd->size - *pos_f < want ? fwd->size - *pos_f : want;
    TIMED(0, build_index(&W->F, fwd->data + *pos_f, wf, *pos_f + wf == fwd->size));
    size_t n_rec = W->F.n_lines / (size_t)P.lines;
    if (P.paired) {
        size_t wr = rev->size - *pos_r < want ? rev->size - *pos_r : want;
        TIMED(0, build_index(&W->R, rev->data + *pos_r, wr, *pos_r + wr == rev->size));
        if (W->R.n_lines / (size_t)P.lines < n_rec) n_rec = W->R.n_lines / (size_t)P.lines;
    }
    if (n_rec > cap_units) n_rec = cap_units;
    if (n_rec == 0) return false; /* fewer lines left than one record (pair): the loop of ref:1605 ends */
    if (n_rec > W->cap_use) {
        free(W->use);
        W->cap_use = n_rec + n_rec / 8;
        W->use = malloc(W->cap_use);
        if (!W->use) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
    }
    for (;;) {
        W->n_rec = n_rec;
        P.long_line = 0;
        TIMED(1, parallel_run(size_records, NULL));
        if (P.long_line) {
            fprintf(stderr, "FATAL: input line longer than %d bytes\n", MAX_LINE);
            exit(EXIT_FAILURE);
        }
        *bad = SIZE_MAX;
        for (int t = 0; t < n_threads; t++)
            if (P.t_bad[t] < *bad) { *bad = P.t_bad[t]; *bad_rev = P.t_bad_rev[t]; }
        size_t units = 0, bases = 0;
        for (int t = 0; t < n_threads; t++) {
            size_t a, b;
            rec_range(n_rec, t, n_threads, &a, &b);
            if (a > *bad) P.t_units[t] = P.t_bases[t] = 0; /* beyond the record the run stops at */
            units += P.t_units[t];
            bases += P.t_bases[t];
        }
        if (bases <= B.cap_bases) { W->n_units = (uint32_t)units; W->n_bases = bases; break; }
        n_rec = (size_t)((double)n_rec * (double)B.cap_bases / (double)bases * 0.98); /* too many bases for one batch */
        if (n_rec == 0) n_rec = 1;
    }
    W->n_reads = P.paired ? 2 * W->n_units : W->n_units;
    /* on to the byte after the last record taken */
    const size_t last = n_rec * (size_t)P.lines - 1;
    const size_t used_f = line_end(&W->F, last) + (last < W->F.n_nl ? 1 : 0);
    *pos_f += used_f;
    *bytes_per_rec = used_f / n_rec + 64;
    if (P.paired) {
        const size_t used_r = line_end(&W->R, last) + (last < W->R.n_nl ? 1 : 0);
        *pos_r += used_r;
        if (used_r / n_rec + 64 > *bytes_per_rec) *bytes_per_rec = used_r / n_rec + 64;
    }
    return true;
}

static inline bool record_terminated(const struct findex *fi, int lines, size_t i)
{
    return i * (size_t)lines + (size_t)lines - 1 < fi->n_nl; /* the seeding pass only sees newline-terminated lines (ref:1322-1373) */
}

/* after sizing: does this window hold a read that the seeding pass would have inserted and the loop will not count? */
static bool lazy_window_needs_seeding(void)
{
    size_t unused = 0;
    for (int t = 0; t < n_threads; t++) unused += P.t_unused[t];
    const bool open = lazy_seen[0] < lazy_limit || (P.paired && lazy_seen[1] < lazy_limit);
    if (unused && open) return true;
    for (int t = 0; t < n_threads; t++) { lazy_seen[0] += P.t_seedable[0][t]; lazy_seen[1] += P.t_seedable[1][t]; }
    return false;
}

/* Seeds those reads (sequence_to_hash_zero, ref:1501-1537): records in order, each file's first lazy_limit reads longer
 * than k count towards its limit, the ones whose record the loop drops are inserted.  Uses the staging area, which
 * is free at this point. */
static void lazy_seed_window(void)
{
    uint32_t n = 0;
    uint64_t at = 0;
    for (size_t i = 0; i < W->n_rec; i++) {
        for (int f = 0; f < (P.paired ? 2 : 1); f++) {
            const struct findex *fi = f == 0 ? &W->F : &W->R;
            uint32_t len;
            const char *sq = seq_line(fi, P.lines, i, &len);
            if (len <= P.K || len > MAX_LINE || !record_terminated(fi, P.lines, i)) continue;
            if (lazy_seen[f]++ >= lazy_limit || W->use[i]) continue;
            if (!valid_bases(sq, len)) { /* the reference's seeding pass dies here: let it */
                writers_stop();
                unlink_outputs_on_seed_fatal = true;
                W = &WIN[2];
                seed_file(cfg.fwd[0], (int)lazy_limit);
                if (cfg.n_rev > 0) seed_file(cfg.rev[0], (int)lazy_limit);
                fprintf(stderr, "internal error: the seeding pass did not stop at the invalid read\n");
                exit(EXIT_FAILURE);
            }
            if (n == B.cap_reads || at + len > B.cap_bases) {
                int rc = eng_seed(B.seq, B.offsets, n);
                if (rc != NKML_OK) die_nkml("seed", rc);
                n = 0;
                at = 0;
            }
            memcpy(B.seq + at, sq, len);
            at += len;
            B.offsets[++n] = at;
        }
    }
    if (n) {
        int rc = eng_seed(B.seq, B.offsets, n);
        if (rc != NKML_OK) die_nkml("seed", rc);
    }
}

/* The record loop of process_thread_chunk_paired / _single (ref:1568-1770, 1921-2111) for one file (pair):
 * window after window, each a whole number of records that fits one GPU batch.  While the GPU counts window i the
 * host indexes and sizes window i+1, and once window i+1 has been handed over, formats and writes window i. */
static void process_files(const struct mapped *fwd, const struct mapped *rev)
{
    P.lines = cfg.fastq_in ? 4 : 2;
    P.paired = rev != NULL;
    P.to_fasta = cfg.fastq_in && !cfg.fastq_out;
    P.K = (uint32_t)cfg.ksize;
    P.size_validates = false;
    size_t pos_f = 0, pos_r = 0;
    size_t bytes_per_rec = 512; /* window size guess; follows the data */
    struct window *prev = NULL;
    W = &WIN[0];
    for (;;) {
        struct window *cur = W;
        size_t bad = SIZE_MAX;
        int bad_rev = 0;
        const bool more = size_window(fwd, rev, &pos_f, &pos_r, &bytes_per_rec, &bad, &bad_rev);
        bool prev_done = false;
        if (more) {
            /* the staging area is the library's: the previous window's copy to the device has to be over */
            if (prev) { int rc = eng_wait_staged(); if (rc != NKML_OK) die_nkml("wait", rc); }
            if (lazy_seed && lazy_window_needs_seeding()) {
                if (prev) { W = prev; finish_window(); W = cur; prev_done = true; }
                TIMED(6, lazy_seed_window());
            }
            TIMED(2, parallel_run(stage_records, NULL));
            /* a record with a bad base: the window ends before it (everything before it is counted and written, see below) */
            for (int t = 0; t < n_threads; t++)
                if (P.t_bad[t] < bad) {
                    bad = P.t_bad[t];
                    bad_rev = P.t_bad_rev[t];
                    W->n_units = (uint32_t)P.t_bad_unit[t];
                }
            if (bad != SIZE_MAX) {
                W->n_reads = P.paired ? 2 * W->n_units : W->n_units;
                W->n_bases = B.offsets[W->n_reads];
            }
        }
        if (prev && !prev_done) { W = prev; finish_window(); W = cur; }
        if (more) submit_window();
        if (prev) { W = prev; write_window(); W = cur; }
        if (!more) break;
        if (bad != SIZE_MAX) { /* everything before the bad record is counted and written, then the reference's exit */
            finish_window();
            if (lazy_seed) {
                /* would the reference's seeding pass have met it first?  Run that pass now: it ends the run like the reference
                 * (no output files yet) if so, and comes back if the record lies beyond the seeded prefix */
                struct window *keep = W;
                const struct plan keep_plan = P;
                writers_stop();
                unlink_outputs_on_seed_fatal = true;
                W = &WIN[2];
                seed_file(cfg.fwd[0], (int)lazy_limit);
                if (cfg.n_rev > 0) seed_file(cfg.rev[0], (int)lazy_limit);
                unlink_outputs_on_seed_fatal = false;
                W = keep;
                P = keep_plan;
                writers_start();
            }
            write_window();
            writers_stop();
            uint32_t len;
            const char *sq = seq_line(bad_rev ? &W->R : &W->F, P.lines, bad, &len);
            fatal_sequence(bad_rev ? "REV" : "FWD", sq, len);
            exit(EXIT_FAILURE);
        }
        prev = cur;
        W = cur == &WIN[0] ? &WIN[1] : &WIN[0];
    }
    W = &WIN[0];
}

/* seed_kmer_hash, ref:1322-1373: the first `limit` records whose sequence line is longer than k, of one file,
 * enter the table with count 0.  Only newline-terminated lines count, as in the reference's byte loop. */
static void seed_file(const char *name, int limit)
{
    if (cfg.verbose) printf("Seeding hash table with up to %'d records from file %s\n", limit, name);
    struct mapped m = map_file(name);
    if (!m.data) return;
    P.lines = cfg.fastq_in ? 4 : 2;
    P.paired = false;
    P.K = (uint32_t)cfg.ksize + 1; /* strictly longer than k, ref:1352 */
    P.size_validates = true;
    size_t size = m.size;
    /* the reference only sees complete, newline-terminated lines: stop at the last newline */
    while (size > 0 && m.data[size - 1] != '\n') size--;
    size_t pos = 0, done = 0, bytes_per_rec = 512;
    while (done < (size_t)limit) {
        size_t want = (size_t)B.cap_reads * bytes_per_rec;
        if (want > (size_t)3 << 30) want = (size_t)3 << 30;
        const size_t w = size - pos < want ? size - pos : want;
        build_index(&W->F, m.data + pos, w, false);
        size_t n_rec = W->F.n_lines / (size_t)P.lines;
        if (n_rec > B.cap_reads) n_rec = B.cap_reads;
        if (n_rec == 0) break;
        if (n_rec > W->cap_use) {
            free(W->use);
            W->cap_use = n_rec + n_rec / 8;
            W->use = malloc(W->cap_use);
            if (!W->use) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
        }
        size_t units, bases;
        for (;;) {
            W->n_rec = n_rec;
            P.long_line = 0;
            parallel_run(size_records, NULL);
            if (P.long_line) {
                fprintf(stderr, "FATAL: input line longer than %d bytes\n", MAX_LINE);
                exit(EXIT_FAILURE);
            }
            size_t bad = SIZE_MAX;
            for (int t = 0; t < n_threads; t++)
                if (P.t_bad[t] < bad) bad = P.t_bad[t];
            units = bases = 0;
            size_t cut = SIZE_MAX; /* record at which the limit is reached, if it is reached in this window */
            for (int t = 0; t < n_threads && cut == SIZE_MAX; t++) {
                size_t a, e;
                rec_range(n_rec, t, n_threads, &a, &e);
                if (a > bad) break;
                if (units + P.t_units[t] > (size_t)limit - done) {
                    size_t have = units;
                    for (size_t i = a; i < e; i++)
                        if (W->use[i] && ++have == (size_t)limit - done) { cut = i; break; }
                }
                units += P.t_units[t];
                bases += P.t_bases[t];
            }
            if (cut != SIZE_MAX && cut + 1 < n_rec) { n_rec = cut + 1; continue; } /* only the records up to the limit */
            if (bad != SIZE_MAX) { /* a record the reference would have stopped at before reaching the limit */
                uint32_t len;
                const char *sq = seq_line(&W->F, P.lines, bad, &len);
                if (unlink_outputs_on_seed_fatal) /* the reference dies here before it has created its output files */
                    for (int f = 0; f < 2; f++)
                        if (out_name[f]) unlink(out_name[f]);
                fatal_sequence("FWD", sq, len);
                exit(EXIT_FAILURE);
            }
            if (bases <= B.cap_bases) break;
            n_rec = (size_t)((double)n_rec * (double)B.cap_bases / (double)bases * 0.98);
            if (n_rec == 0) n_rec = 1;
        }
        W->n_units = W->n_reads = (uint32_t)units;
        W->n_bases = bases;
        if (units) {
            parallel_run(stage_records, NULL);
            int rc = eng_seed(B.seq, B.offsets, W->n_reads);
            if (rc != NKML_OK) die_nkml("seed", rc);
        }
        done += units;
        W->n_reads = W->n_units = 0;
        W->n_bases = 0;
        const size_t last = n_rec * (size_t)P.lines - 1;
        const size_t used = line_end(&W->F, last) + 1;
        pos += used;
        bytes_per_rec = used / n_rec + 64;
    }
    if (cfg.verbose) {
        nkml_stats st;
        eng_stats(&st);
        printf("%'zu records with %'zu kmers seeded\n", done, (size_t)st.unique_kmers);
    }
    unmap_file(&m);
}

static char *output_name(const char *base, int k, int depth, int thread, const char *suffix)
{
    size_t n = strlen(base) + 64;
    char *s = malloc(n);
    if (!s) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
    if (thread >= 0) snprintf(s, n, "%s.k%d_norm%d_thread%d.%s", base, k, depth, thread, suffix); /* ref:845 */
    else snprintf(s, n, "%s.k%d_norm%d.%s", base, k, depth, suffix);
    return s;
}

/* print_kmer_table, ref:354-385: kmer<TAB>count for every stored k-mer, unsorted */
static void print_table(const char *identifier, int thread)
{
    uint64_t n = 0;
    int rc = eng_dump(NULL, NULL, 0, &n);
    if (rc != NKML_OK) die_nkml("dump", rc);
    uint64_t *codes = malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t *counts = malloc((n ? n : 1) * sizeof(uint32_t));
    if (!codes || !counts) { fprintf(stderr, "Memory allocation failed\n"); exit(EXIT_FAILURE); }
    uint64_t got = 0;
    rc = eng_dump(codes, counts, n, &got);
    if (rc != NKML_OK) die_nkml("dump", rc);
    if (got < n) n = got;
    char base[32];
    snprintf(base, sizeof base, "output_kmer%s", identifier);
    char *name = output_name(base, cfg.ksize, cfg.depth, thread, "tsv");
    FILE *f = fopen(name, "w");
    if (f) {
        char kmer[40];
        for (uint64_t i = 0; i < n; i++) {
            uint64_t e = codes[i];
            for (int j = cfg.ksize - 1; j >= 0; j--) { kmer[j] = "ACGT"[e & 3]; e >>= 2; } /* decode_kmer_plain, ref:1128-1136 */
            kmer[cfg.ksize] = 0;
            fprintf(f, "%s\t%u\n", kmer, counts[i]);
        }
        fclose(f);
    }
    free(name); free(codes); free(counts);
}

int main(int argc, char **argv)
{
    setlocale(LC_ALL, "");
    if (!parse_arguments(argc, argv)) { usage(); return 1; }

    nkml_config nc;
    memset(&nc, 0, sizeof nc);
    nc.ksize = cfg.ksize;
    nc.depth = cfg.depth;
    nc.canonical = cfg.canonical;
    nc.device = getenv("NKML_DEVICE") ? atoi(getenv("NKML_DEVICE")) : 0;
    nc.memory_gb = cfg.memory;
    nc.max_batch_reads = 1u << 22;      /* 2 M pairs of 150 bp per batch: where the k=15 path is at its best */
    nc.max_batch_bases = 640ull << 20;
    if (getenv("NKML_BATCH_READS")) nc.max_batch_reads = (uint32_t)strtoul(getenv("NKML_BATCH_READS"), NULL, 10);
    if (getenv("NKML_BATCH_BASES")) nc.max_batch_bases = strtoull(getenv("NKML_BATCH_BASES"), NULL, 10);
    if (cfg.doprint) nc.flags |= NKML_F_EXACT_COUNTS;
    {
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        n_threads = cpus > 16 ? 16 : cpus < 1 ? 1 : (int)cpus; /* host threads of the record loop (not the reference's --cpu) */
        if (getenv("NKML_THREADS")) n_threads = atoi(getenv("NKML_THREADS"));
        if (n_threads < 1) n_threads = 1;
        if (n_threads > MAX_THREADS) n_threads = MAX_THREADS;
    }
    timing_on = getenv("NKML_TIMING") != NULL;
    /* NKML_DEVICES: a number n (devices 0 .. n-1) or a comma-separated list of HIP ordinals, one per table shard (an
     * ordinal may repeat: shards then share that GPU).  More than one entry: one k-mer table sharded over them. */
    int32_t devices[64];
    uint32_t n_devices = 0;
    if (getenv("NKML_DEVICES")) {
        const char *p = getenv("NKML_DEVICES");
        if (!strchr(p, ',')) {
            const long n = strtol(p, NULL, 10);
            for (long i = 0; i < n && i < 64; i++) devices[n_devices++] = (int32_t)i;
        } else {
            while (*p && n_devices < 64) {
                char *end;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                devices[n_devices++] = (int32_t)d;
                p = *end == ',' ? end + 1 : end;
            }
        }
    }
    int rc = 0;
    TIMED(7, {
        if (n_devices > 1) {
            rc = nkml_multi_create(&nc, devices, n_devices, &mctx);
            if (rc != NKML_OK) die_nkml("cannot set up the k-mer table over the GPUs of NKML_DEVICES", rc);
        } else {
            if (n_devices == 1) nc.device = devices[0];
            rc = nkml_create(&nc, &ctx);
            if (rc != NKML_OK) die_nkml("cannot set up the GPU k-mer table", rc);
        }
        batch_alloc(nc.max_batch_reads, nc.max_batch_bases);
    });

    /* seeding pass over every input file, before any counting: ref:2242-2250 */
    int seed_n = 1 + (int)(SEED_RECORDS / cfg.n_fwd);
    if (!getenv("NKML_NO_SEED")) {
        /* the first file (pair) is seeded from within the record loop (see lazy_seed) unless the seeding pass itself is
         * on show: --verbose reports it, --print dumps the seeded table, NKML_EAGER_SEED=1 asks for it */
        lazy_seed = !cfg.verbose && !cfg.doprint && !getenv("NKML_EAGER_SEED");
        lazy_limit = (size_t)seed_n;
        TIMED(6, {
            for (int i = lazy_seed ? 1 : 0; i < cfg.n_fwd; i++) {
                seed_file(cfg.fwd[i], seed_n);
                if (i < cfg.n_rev) seed_file(cfg.rev[i], seed_n);
            }
        });
    }
    if (cfg.doprint) print_table("_seeds", -1);

    char *name_f = output_name("output_forward", cfg.ksize, cfg.depth, 0, "fastq"); /* ref:2286: always .fastq */
    out_name[0] = name_f;
    out_fd[0] = open(name_f, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (out_fd[0] < 0) { fprintf(stderr, "Error opening file to write: %s\n", name_f); return EXIT_FAILURE; }
    char *name_r = NULL;
    if (cfg.n_rev != 0) { /* ref:2294 */
        name_r = output_name("output_reverse", cfg.ksize, cfg.depth, 0, "fastq");
        out_name[1] = name_r;
        out_fd[1] = open(name_r, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (out_fd[1] < 0) { fprintf(stderr, "Error opening file to write: %s\n", name_r); return EXIT_FAILURE; }
    }

    writers_start();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t max_unique = 0;
    for (int i = 0; i < cfg.n_fwd; i++) {
        struct mapped mf = map_file(cfg.fwd[i]), mr = {NULL, 0};
        const bool paired = i < cfg.n_rev;
        if (paired) {
            printf("Processing file pair %d of %d: %s and %s\n", i + 1, cfg.n_fwd, cfg.fwd[i], cfg.rev[i]);
            mr = map_file(cfg.rev[i]);
            if (!mf.data || !mr.data) { fprintf(stderr, "Error memory mapping input files\n"); break; }
        } else {
            printf("Processing single-ended file %d of %d: %s\n", i + 1, cfg.n_fwd, cfg.fwd[i]);
            if (!mf.data) { fprintf(stderr, "Error memory mapping input files\n"); break; }
        }
        const char first = cfg.fastq_in ? '@' : '>'; /* ref:2337-2361 */
        if (mf.size == 0 || mf.data[0] != first) {
            fprintf(stderr, "Input %s file %s starts with %c which is not expected\n", cfg.fastq_in ? "FASTQ" : "FASTA", cfg.fwd[i], mf.size ? mf.data[0] : ' ');
            exit(EXIT_FAILURE);
        }
        if (paired && (mr.size == 0 || mr.data[0] != first)) {
            fprintf(stderr, "Input %s file %s starts with %c which is not expected\n", cfg.fastq_in ? "FASTQ" : "FASTA", cfg.rev[i], mr.size ? mr.data[0] : ' ');
            exit(EXIT_FAILURE);
        }
        struct timespec f0, f1;
        clock_gettime(CLOCK_MONOTONIC, &f0);
        size_t before = tot.processed;
        process_files(&mf, paired ? &mr : NULL);
        lazy_seed = false; /* only the first file (pair) was left to the loop */
        clock_gettime(CLOCK_MONOTONIC, &f1);
        nkml_stats st;
        if ((rc = eng_stats(&st)) != NKML_OK) die_nkml("stats", rc);
        if ((size_t)st.unique_kmers > max_unique) max_unique = (size_t)st.unique_kmers;
        double secs = (double)(f1.tv_sec - f0.tv_sec) + 1e-9 * (double)(f1.tv_nsec - f0.tv_nsec);
        /* ref:1752-1759 (rate measured with a finer clock than the reference's 1 s) and ref:1911 */
        printf("Thread 0 - Processing rate: %'.0f sequences/s, processed %'zu pairs, printed: %'zu, skipped: %'zu, Unique kmers (all sequences; this thread): %'zu\n",
               secs > 0 ? (double)(tot.processed - before) / secs : 0.0, tot.processed, tot.printed, tot.skipped, (size_t)st.unique_kmers);
        printf("Cumulative file statistics: Processed %'zu, Printed %'zu, Skipped %'zu, Cumulative Max Unique Kmers in a thread: %'zu\n",
               tot.processed, tot.printed, tot.skipped, (size_t)st.unique_kmers);
        unmap_file(&mf);
        if (paired) unmap_file(&mr);
    }
    TIMED(5, writers_stop());
    close(out_fd[0]);
    if (out_fd[1] >= 0) close(out_fd[1]);
    if (cfg.doprint) print_table("", 0);

    /* ref:2415-2419, 2438-2453 */
    printf("\n--- Final Report ---\n");
    printf("Processed Records: %'zu\n", tot.processed);
    printf("Printed Records: %'zu\n", tot.printed);
    printf("Skipped Records: %'zu\n", tot.skipped);
    printf("Cumulative Max unique kmers in any thread: %'zu\n", max_unique);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double runtime = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("Total runtime: %.2f seconds\n", runtime);
    if (tot.processed > 0) {
        if (cfg.n_rev != 0) printf("Overall processing rate: %'.0f sequence pairs per second\n", (double)tot.processed / runtime);
        else printf("Overall processing rate: %'.0f sequences per second\n", (double)tot.processed / runtime);
    } else {
        printf("No data processed\n");
    }
    if (timing_on) {
        fprintf(stderr, "host phases (s):");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.3f", t_name[i], t_phase[i]);
        fprintf(stderr, " (threads %d)\n", n_threads);
    }
    if (mctx) nkml_multi_destroy(mctx);
    else nkml_destroy(ctx);
    free(name_f);
    free(name_r);
    return 0;
}
