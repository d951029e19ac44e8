/*
 * kssd_oracle.c -- TEST INFRASTRUCTURE ONLY (see kssd_oracle.h).
 *
 * CPU restatement of the kssd sketch + distance hot path.  Own code, written from the behaviour of
 * the reference (/root/reference, KSSD v1.2.21); each function names the reference lines it follows.
 * The algorithmic structure of the reference is kept on purpose (one 2^21-slot double-hashing table
 * per genome, posting-list traversal) so that it can also serve as the "port" CPU baseline.
 */
#define _GNU_SOURCE
#include "kssd_oracle.h"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef uint32_t u32;

#define KO_COMPONENT_SZ 7        /* reference Makefile:4  -DCOMPONENT_SZ=7                          */
#define KO_CTX_SPC_USE_L 8       /* global_basic.h:45-47                                           */
#define KO_MIN_SUBCTX_DIM 4096   /* command_shuffle.h:29 MIN_SUBCTX_DIM_SMP_SZ                     */
#define KO_LD_FCTR 0.6           /* global_basic.h:49                                              */
#define KO_HIBIT 0x8000000000000000ULL

/* table sizes for the per-genome hash: largest primes below powers of two (global_basic.c:74-81) */
static const u32 ko_primes[25] = {
    251u, 509u, 1021u, 2039u, 4093u, 8191u, 16381u, 32749u, 65521u, 131071u, 262139u, 524287u,
    1048573u, 2097143u, 4194301u, 8388593u, 16777213u, 33554393u, 67108859u, 134217689u,
    268435399u, 536870909u, 1073741789u, 2147483647u, 4294967291u};

struct ko_ctx {
    ko_params p;
    const int32_t *table;
    u64 *co; /* hashsize slots */
};

/* A/C/G/T in either case -> 0..3, anything else -1 (global_basic.c:64-71; bytes >= 128 index the
 * reference's table out of range, we treat them as invalid) */
static inline int base_code(unsigned char ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* seq2co_global_var_initial (iseq2comem.c:54-77) + get_hashsz (command_dist.c:217-236) */
int ko_params_init(ko_params *p, int shuf_id, int k, int subk, int drlevel)
{
    memset(p, 0, sizeof *p);
    if (k < subk || subk >= 8 || subk < 1 || drlevel < 0 || k > 15) /* command_shuffle.c:163-168 */
        return KO_ERR_PARAM;
    int pidx = 4 * (k - drlevel) - KO_CTX_SPC_USE_L - 7;
    if (pidx < 0 || pidx > 24)
        return KO_ERR_PARAM;
    p->shuf_id = shuf_id;
    p->k = k;
    p->subk = subk;
    p->drlevel = drlevel;
    p->TL = 2 * k;
    p->out = k - subk;
    p->hashsize = ko_primes[pidx];
    p->hashlimit = (u32)(p->hashsize * KO_LD_FCTR);
    int extra = k - drlevel - KO_COMPONENT_SZ;
    p->comp_bits = extra > 0 ? 4 * extra : 0;
    p->comp_num = extra > 0 ? (int)(1u << p->comp_bits) : 1;
    p->rc_shift = 4 * k - 2;
    p->tupmask = ~0ULL >> (64 - 4 * k);
    p->domask = ((1ULL << (4 * subk)) - 1) << (2 * p->out);
    p->undomask = ((1ULL << (2 * p->out)) - 1) << (2 * (k + subk));
    int64_t sub = 1LL << (4 * (subk - drlevel > 0 ? subk - drlevel : 0));
    if (subk - drlevel < 0)
        sub = 0;
    p->dim_end = sub > KO_MIN_SUBCTX_DIM ? sub : KO_MIN_SUBCTX_DIM;
    return 0;
}

ko_ctx *ko_open(const int32_t *table, int shuf_id, int k, int subk, int drlevel)
{
    ko_ctx *c = calloc(1, sizeof *c);
    if (!c)
        return NULL;
    if (ko_params_init(&c->p, shuf_id, k, subk, drlevel) != 0) {
        free(c);
        return NULL;
    }
    c->table = table;
    c->co = malloc((size_t)c->p.hashsize * sizeof(u64));
    if (!c->co) {
        free(c);
        return NULL;
    }
    return c;
}

void ko_close(ko_ctx *c)
{
    if (c) {
        free(c->co);
        free(c);
    }
}

const ko_params *ko_get_params(const ko_ctx *c) { return &c->p; }
void ko_free(void *p) { free(p); }

/* read_dim_shuffle_file (command_shuffle.c:192-207): 16-byte header + int32[16^subk] */
int ko_shuf_read(const char *path, int hdr[4], int32_t **table)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return KO_ERR_IO;
    if (fread(hdr, sizeof(int), 4, f) != 4 || hdr[2] < 1 || hdr[2] >= 8) {
        fclose(f);
        return KO_ERR_IO;
    }
    size_t n = (size_t)1 << (4 * hdr[2]);
    int32_t *t = malloc(n * sizeof(int32_t));
    if (!t || fread(t, sizeof(int32_t), n, f) != n) {
        free(t);
        fclose(f);
        return KO_ERR_IO;
    }
    fclose(f);
    *table = t;
    return 0;
}

/* canonical strand, .shuf filter and reduced-tuple encoding (iseq2comem.c:245-253).
 * returns 1 and sets *dr when the k-mer survives the filter */
static inline int reduce_kmer(const ko_ctx *c, u64 fwd, u64 rev, u64 *dr)
{
    const ko_params *p = &c->p;
    u64 u = fwd < rev ? fwd : rev;
    u64 dim = (u & p->domask) >> (2 * p->out);
    int64_t pf = c->table[dim];
    if (pf < 0 || pf >= p->dim_end)
        return 0;
    u64 low = u & ((1ULL << (2 * p->out)) - 1);
    *dr = (((u & p->undomask) + (low << (2 * p->TL - 4 * p->out))) >> (4 * p->drlevel)) + (u64)pf;
    return 1;
}

/* double hashing probe sequence (global_basic.h:228-230) */
static inline u32 probe_slot(u64 key, u64 i, u64 S) { return (u32)((key % S + i * (1 + key % (S - 1))) % S); }

/* slot-order dump shared by the three writers */
static long dump_plain(const ko_ctx *c, u32 *ids, uint8_t *comps, size_t cap)
{
    /* wrt_co2cmpn_use_inn_subctx (iseq2comem.c:525-551): non-empty and top bit clear */
    const ko_params *p = &c->p;
    size_t w = 0;
    for (u32 s = 0; s < p->hashsize; s++) {
        u64 v = c->co[s];
        if (v != 0 && v < KO_HIBIT) {
            if (w >= cap)
                return KO_ERR_BUFSZ;
            ids[w] = (u32)(v >> p->comp_bits);
            if (comps)
                comps[w] = (uint8_t)(v % (u64)p->comp_num);
            w++;
        }
    }
    return (long)w;
}

long ko_fasta2co(ko_ctx *c, const unsigned char *text, size_t n, int uniq, u32 *ids, uint8_t *comps, size_t cap)
{
    const ko_params *p = &c->p;
    if (n == 0)
        return KO_ERR_EMPTY; /* iseq2comem.c:201-202 */
    memset(c->co, 0, (size_t)p->hashsize * sizeof(u64)); /* :192 */
    u64 fwd = 0, rev = 0, run = 1; /* run = "base" counter of the reference, starts at 1 (:203) */
    u32 keycount = 0;
    const u64 S = p->hashsize;
    for (size_t i = 0; i < n; i++) {
        unsigned char ch = text[i];
        int b = base_code(ch);
        if (b >= 0) { /* :215-220 */
            fwd = ((fwd << 2) | (u64)b) & p->tupmask;
            rev = (rev >> 2) + (((u64)b ^ 3ULL) << p->rc_shift);
            run++;
        } else if (ch == '\n' || ch == '\r') { /* :221 line breaks are transparent */
            continue;
        } else if (ch == '>') { /* :223-238 skip the header line wherever '>' shows up */
            while (i < n && text[i] != '\n')
                i++;
            if (i >= n)
                return KO_ERR_HEADER; /* :233 */
            run = 1;
            continue;
        } else { /* :222,:239-242 letters other than ACGT and every other byte break the run */
            run = 1;
            continue;
        }
        if (run <= (u64)p->TL) /* :243 */
            continue;
        u64 dr;
        if (!reduce_kmer(c, fwd, rev, &dr))
            continue;
        for (u64 t = 0; t < S; t++) { /* :255-268 / :683-698 */
            u32 s = probe_slot(dr, t, S);
            if (c->co[s] == 0) {
                c->co[s] = dr; /* dr==0 leaves the slot empty but still counts (:258-261) */
                if (++keycount > p->hashlimit)
                    return KO_ERR_CAPACITY;
                break;
            }
            if (!uniq) {
                if (c->co[s] == dr)
                    break;
            } else if ((c->co[s] | KO_HIBIT) == (dr | KO_HIBIT)) {
                c->co[s] |= KO_HIBIT; /* seen twice: flagged, dropped at dump time */
                break;
            }
        }
    }
    return dump_plain(c, ids, comps, cap);
}

/* ---- sketch by read (dist --byread) --------------------------------------------------------- */
/* reads2mco (iseq2comem.c:78-186): the FASTA scanner of fasta2co without the per-genome table -- every k-mer that
 * passes the .shuf filter is written, in stream order and as often as it occurs; a '>' anywhere starts the next read
 * (:126-150).  ids/comps = the stream of (drtuple >> comp_bits, drtuple % comp_num) (:173-175);
 * read_of[i] = the read entry i belongs to (0 = before the first '>'), from which the caller derives the per-component
 * index files: cumulative counts over reads 0..n_reads, WITHOUT a leading zero (:177-183).
 * *n_reads = number of '>' met.  Returns the number of entries or a negative KO_ERR_*. */
long ko_reads2mco(ko_ctx *c, const unsigned char *text, size_t n, u32 *ids, uint8_t *comps, u32 *read_of, size_t cap,
                  uint64_t *n_reads)
{
    const ko_params *p = &c->p;
    if (n == 0)
        return KO_ERR_EMPTY; /* :104 */
    u64 fwd = 0, rev = 0, run = 1, readn = 0;
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char ch = text[i];
        int b = base_code(ch);
        if (b >= 0) { /* :117-122 */
            fwd = ((fwd << 2) | (u64)b) & p->tupmask;
            rev = (rev >> 2) + (((u64)b ^ 3ULL) << p->rc_shift);
            run++;
        } else if (ch == '\n' || ch == '\r') { /* :123 */
            continue;
        } else if (ch == '>') { /* :125-151: next read, skip the header line */
            readn++;
            while (i < n && text[i] != '\n')
                i++;
            if (i >= n)
                return KO_ERR_HEADER; /* :146 */
            run = 1;
            continue;
        } else { /* :124,:152-155 */
            run = 1;
            continue;
        }
        if (run <= (u64)p->TL) /* :156 */
            continue;
        u64 dr;
        if (!reduce_kmer(c, fwd, rev, &dr)) /* :158-170 */
            continue;
        if (w >= cap)
            return KO_ERR_BUFSZ;
        ids[w] = (u32)(dr >> p->comp_bits); /* :173 */
        if (comps)
            comps[w] = (uint8_t)(dr % (u64)p->comp_num);
        read_of[w] = (u32)readn; /* :172 */
        w++;
    }
    if (n_reads)
        *n_reads = readn;
    return (long)w;
}

/* ---- FASTQ ------------------------------------------------------------------------------- */
#define KO_FQ_LEN 20000 /* iseq2comem.c:274 */

typedef struct {
    const unsigned char *p;
    size_t n, pos;
    int eof;
} mstream;

/* fgets() over a memory stream, including its end-of-file indicator semantics */
static char *m_fgets(char *buf, int size, mstream *s)
{
    int i = 0;
    while (i < size - 1) {
        if (s->pos >= s->n) {
            s->eof = 1;
            break;
        }
        char ch = (char)s->p[s->pos++];
        buf[i++] = ch;
        if (ch == '\n')
            break;
    }
    if (i == 0)
        return NULL;
    buf[i] = 0;
    return buf;
}

long ko_fastq2co(ko_ctx *c, const unsigned char *text, size_t n, int Q, int M, u32 *ids, uint8_t *comps, size_t cap)
{
    const ko_params *p = &c->p;
    if (M >= 15 || M < 1)
        return KO_ERR_PARAM; /* :279 */
    memset(c->co, 0, (size_t)p->hashsize * sizeof(u64));
    char *seq = calloc(1, KO_FQ_LEN + 10), *qual = calloc(1, KO_FQ_LEN + 10);
    mstream ms = {text, n, 0, 0};
    /* a record = 4 fgets() calls; only line 2 and line 4 survive in seq / qual (:291-292) */
    m_fgets(seq, KO_FQ_LEN, &ms);
    m_fgets(seq, KO_FQ_LEN, &ms);
    m_fgets(qual, KO_FQ_LEN, &ms);
    m_fgets(qual, KO_FQ_LEN, &ms);
    u64 fwd = 0, rev = 0, run = 1;
    const u64 S = p->hashsize;
    int sl = (int)strlen(seq);
    for (int pos = 0; pos < sl; pos++) {
        if (seq[pos] == '\n') { /* :297-308 next record; stop once the stream has hit EOF */
            m_fgets(seq, KO_FQ_LEN, &ms);
            m_fgets(seq, KO_FQ_LEN, &ms);
            m_fgets(qual, KO_FQ_LEN, &ms);
            m_fgets(qual, KO_FQ_LEN, &ms);
            sl = (int)strlen(seq);
            if (ms.eof)
                break;
            run = 1;
            pos = -1;
            continue;
        }
        int b = base_code((unsigned char)seq[pos]);
        if (b >= 0 && qual[pos] >= Q) { /* :312 raw ASCII compare on a (signed) char */
            fwd = ((fwd << 2) | (u64)b) & p->tupmask;
            rev = (rev >> 2) + (((u64)b ^ 3ULL) << p->rc_shift);
            run++;
        } else {
            run = 1;
            continue;
        }
        if (run <= (u64)p->TL)
            continue;
        u64 dr;
        if (!reduce_kmer(c, fwd, rev, &dr))
            continue;
        for (u64 t = 0; t < S; t++) { /* :333-348 4-bit saturating occurrence counter */
            u32 s = probe_slot(dr, t, S);
            u64 v = c->co[s];
            if (v == 0) {
                c->co[s] = (M == 1) ? ((dr << 4) | 0xFULL) : ((dr << 4) + 1ULL);
                break;
            }
            if ((v >> 4) == dr) {
                if ((v & 0xF) != 0xF) {
                    v += 1;
                    if (!((v & 0xF) < (u64)M))
                        v |= 0xF;
                    c->co[s] = v;
                }
                break;
            }
        }
    }
    free(seq);
    free(qual);
    /* write_fqco2file (:499-524): keep slots whose counter nibble saturated */
    size_t w = 0;
    for (u32 s = 0; s < p->hashsize; s++) {
        u64 v = c->co[s];
        if ((v & 0xF) == 0xF) {
            if (w >= cap)
                return KO_ERR_BUFSZ;
            ids[w] = (u32)(v >> (p->comp_bits + 4));
            if (comps)
                comps[w] = (uint8_t)((v >> 4) % (u64)p->comp_num);
            w++;
        }
    }
    return (long)w;
}

/* whole (possibly gzip'ed) file into memory; the reference streams `zcat -fc file` (iseq2comem.c:187) */
/* ---- FASTQ, abundance sketches (dist -A) -------------------------------------------------- */
#define KO_KOC_LEN 4096    /* FQ_LEN, iseq2comem.c:553 */
#define KO_OCCRC_BIT 16    /* iseq2comem.c:357 */
#define KO_OCCRC_MAX 0xffffULL

/* mt_shortreads2koc with one thread + write_fqkoc2files (iseq2comem.c:554-615, 435-471): every sampled k-mer of the
 * second line of every four-line record counted in a 16-bit saturating counter beside its key; no quality rule, no -n;
 * crowding is fatal.  Dump in hash-slot order: ids, component, occurrences. */
long ko_fastq2koc(ko_ctx *c, const unsigned char *text, size_t n, u32 *ids, uint8_t *comps, uint16_t *counts, size_t cap)
{
    const ko_params *p = &c->p;
    memset(c->co, 0, (size_t)p->hashsize * sizeof(u64));
    char *seq = calloc(1, KO_KOC_LEN + 10), *tmp = calloc(1, KO_KOC_LEN + 10);
    mstream ms = {text, n, 0, 0};
    const u64 S = p->hashsize;
    u32 keycount = 0;
    long rc = 0;
    while (!rc && m_fgets(tmp, KO_KOC_LEN, &ms) && m_fgets(seq, KO_KOC_LEN, &ms) && m_fgets(tmp, KO_KOC_LEN, &ms) &&
           m_fgets(tmp, KO_KOC_LEN, &ms)) { /* :567 */
        u64 fwd = 0, rev = 0, run = 1;
        for (int pos = 0; seq[pos] && seq[pos] != '\n' && !rc; pos++) { /* :574 */
            int b = base_code((unsigned char)seq[pos]);
            if (b < 0) { /* :580 */
                run = 1;
                continue;
            }
            fwd = ((fwd << 2) | (u64)b) & p->tupmask;
            rev = (rev >> 2) + (((u64)b ^ 3ULL) << p->rc_shift);
            run++;
            if (run <= (u64)p->TL)
                continue;
            u64 dr;
            if (!reduce_kmer(c, fwd, rev, &dr))
                continue;
            for (u64 t = 0; t < S; t++) { /* :591-607 */
                u32 s = probe_slot(dr, t, S);
                u64 v = c->co[s];
                if (v == 0) {
                    c->co[s] = (dr << KO_OCCRC_BIT) + 1ULL;
                    if (++keycount > p->hashlimit)
                        rc = KO_ERR_CAPACITY;
                    break;
                }
                if ((v >> KO_OCCRC_BIT) == dr) {
                    if ((v & KO_OCCRC_MAX) < KO_OCCRC_MAX)
                        c->co[s] = v + 1ULL;
                    break;
                }
            }
        }
    }
    free(seq);
    free(tmp);
    if (rc)
        return rc;
    size_t w = 0;
    for (u32 s = 0; s < p->hashsize; s++) { /* :452-463 */
        u64 v = c->co[s];
        if (v > 0) {
            if (w >= cap)
                return KO_ERR_BUFSZ;
            ids[w] = (u32)(v >> (p->comp_bits + KO_OCCRC_BIT));
            if (comps)
                comps[w] = (uint8_t)((v >> KO_OCCRC_BIT) % (u64)p->comp_num);
            counts[w] = (uint16_t)(v & KO_OCCRC_MAX);
            w++;
        }
    }
    return (long)w;
}

static unsigned char *slurp_gz(const char *path, size_t *len)
{
    gzFile g = gzopen(path, "rb");
    if (!g)
        return NULL;
    gzbuffer(g, 1 << 20);
    size_t cap = 1 << 22, n = 0;
    unsigned char *buf = malloc(cap);
    for (;;) {
        if (cap - n < (1 << 20)) {
            cap *= 2;
            buf = realloc(buf, cap);
        }
        int r = gzread(g, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (r <= 0)
            break;
        n += (size_t)r;
    }
    gzclose(g);
    *len = n;
    return buf;
}

long ko_sketch_file(ko_ctx *c, const char *path, int is_fastq, int uniq, int Q, int M, u32 *ids, uint8_t *comps, size_t cap)
{
    size_t n = 0;
    unsigned char *txt = slurp_gz(path, &n);
    if (!txt)
        return KO_ERR_IO;
    long r = is_fastq ? ko_fastq2co(c, txt, n, Q, M, ids, comps, cap) : ko_fasta2co(c, txt, n, uniq, ids, comps, cap);
    free(txt);
    return r;
}

/* run_stageI (command_dist.c:258-380): one private table per thread, files in parallel */
static long sketch_many(const int32_t *table, int shuf_id, int k, int subk, int drlevel, const char *const *paths,
                        const unsigned char *const *texts, const size_t *lens, int nitems, int threads, u64 *off,
                        u32 *ids, size_t cap)
{
    ko_params pp;
    if (ko_params_init(&pp, shuf_id, k, subk, drlevel) != 0 || pp.comp_num != 1)
        return KO_ERR_PARAM;
    if (threads < 1)
        threads = 1;
    u32 **tmp = calloc((size_t)nitems, sizeof(u32 *));
    long *cnt = calloc((size_t)nitems, sizeof(long));
    long err = 0;
#pragma omp parallel num_threads(threads)
    {
        ko_ctx *c = ko_open(table, shuf_id, k, subk, drlevel);
        u32 *scratch = malloc((size_t)pp.hashlimit * sizeof(u32) + 64);
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < nitems; i++) {
            long r = paths ? ko_sketch_file(c, paths[i], 0, 0, 0, 1, scratch, NULL, pp.hashlimit + 1)
                           : ko_fasta2co(c, texts[i], lens[i], 0, scratch, NULL, pp.hashlimit + 1);
            cnt[i] = r;
            if (r >= 0) {
                tmp[i] = malloc((size_t)(r ? r : 1) * sizeof(u32));
                memcpy(tmp[i], scratch, (size_t)r * sizeof(u32));
            } else {
#pragma omp critical
                err = r;
            }
        }
        free(scratch);
        ko_close(c);
    }
    long total = 0;
    if (!err) {
        off[0] = 0;
        for (int i = 0; i < nitems; i++) {
            if ((size_t)(total + cnt[i]) > cap) {
                err = KO_ERR_BUFSZ;
                break;
            }
            memcpy(ids + total, tmp[i], (size_t)cnt[i] * sizeof(u32));
            total += cnt[i];
            off[i + 1] = (u64)total;
        }
    }
    for (int i = 0; i < nitems; i++)
        free(tmp[i]);
    free(tmp);
    free(cnt);
    return err ? err : total;
}

long ko_sketch_files(const int32_t *table, int shuf_id, int k, int subk, int drlevel, const char *const *paths,
                     int nfiles, int threads, u64 *off, u32 *ids, size_t cap)
{
    return sketch_many(table, shuf_id, k, subk, drlevel, paths, NULL, NULL, nfiles, threads, off, ids, cap);
}

long ko_sketch_texts(const int32_t *table, int shuf_id, int k, int subk, int drlevel, const unsigned char *const *texts,
                     const size_t *lens, int ntexts, int threads, u64 *off, u32 *ids, size_t cap)
{
    return sketch_many(table, shuf_id, k, subk, drlevel, NULL, texts, lens, ntexts, threads, off, ids, cap);
}

/* ---- inverted index + intersection --------------------------------------------------------- */
static void radix_sort_u64(u64 *a, u64 *tmp, size_t n, int bits)
{
    for (int sh = 0; sh < bits; sh += 16) {
        size_t *hist = calloc(65537, sizeof(size_t));
        for (size_t i = 0; i < n; i++)
            hist[((a[i] >> sh) & 0xFFFF) + 1]++;
        for (int b = 0; b < 65536; b++)
            hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; i++)
            tmp[hist[(a[i] >> sh) & 0xFFFF]++] = a[i];
        memcpy(a, tmp, n * sizeof(u64));
        free(hist);
    }
}

/* combco2mco (co2mco.c:25-77): for each k-mer id the ascending list of genome indices that hold it.
 * The reference stores offsets for all 16^7 ids (2 GiB); here only ids that occur are kept. */
long ko_build_index(const u64 *roff, const u32 *rids, int R, u32 *uid, u64 *upos, u32 *post)
{
    size_t n = (size_t)roff[R];
    u64 *pairs = malloc((n ? n : 1) * sizeof(u64)), *tmp = malloc((n ? n : 1) * sizeof(u64));
    for (int g = 0; g < R; g++)
        for (u64 i = roff[g]; i < roff[g + 1]; i++)
            pairs[i] = ((u64)rids[i] << 32) | (u32)g;
    radix_sort_u64(pairs, tmp, n, 64);
    long U = 0;
    for (size_t i = 0; i < n; i++) {
        u32 id = (u32)(pairs[i] >> 32);
        if (U == 0 || uid[U - 1] != id) {
            uid[U] = id;
            upos[U] = i;
            U++;
        }
        post[i] = (u32)pairs[i];
    }
    upos[U] = n;
    free(pairs);
    free(tmp);
    return U;
}

/* mco_cbdco_nobin_dist hot loop (command_dist.c:774-785): one output row per query */
int ko_shared_counts(const u64 *roff, const u32 *rids, int R, const u64 *qoff, const u32 *qids, int Q, u32 *shared,
                     int threads)
{
    size_t n = (size_t)roff[R];
    u32 *uid = malloc((n ? n : 1) * sizeof(u32)), *post = malloc((n ? n : 1) * sizeof(u32));
    u64 *upos = malloc((n + 1) * sizeof(u64));
    long U = ko_build_index(roff, rids, R, uid, upos, post);
    if (threads < 1)
        threads = 1;
    memset(shared, 0, (size_t)Q * (size_t)R * sizeof(u32));
#pragma omp parallel for num_threads(threads) schedule(guided)
    for (int q = 0; q < Q; q++) {
        u32 *row = shared + (size_t)q * (size_t)R;
        for (u64 i = qoff[q]; i < qoff[q + 1]; i++) {
            u32 id = qids[i];
            long lo = 0, hi = U;
            while (lo < hi) {
                long mid = (lo + hi) >> 1;
                if (uid[mid] < id)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo < U && uid[lo] == id)
                for (u64 g = upos[lo]; g < upos[lo + 1]; g++)
                    row[post[g]]++;
        }
    }
    free(uid);
    free(post);
    free(upos);
    return 0;
}

/* ---- metrics and report ---------------------------------------------------------------------- */
static inline double dist_arg(int metric_sel, double m)
{
    /* GET_MATRIC (command_dist.c:1251): Jaccard -> 1/(2J)+0.5 (Mash), containment -> 1/C (Aaf) */
    return metric_sel == 0 ? 1 / (2 * m) + 0.5 : 1 / m;
}

void ko_output_ctrl(u32 X, u32 Y, u32 s, int kmerlen, int dim_rd_len, int metric_sel, int correction,
                    double dthreshold, u64 cmprsn_num, ko_metric *o)
{
    double rs = 0;
    if (correction) { /* command_dist.c:1254-1261 expected number of chance co-occurrences */
        u32 xo = X - s, yo = Y - s;
        double miss = 1 - 1 / pow((double)4, (double)(kmerlen - dim_rd_len));
        double px = 1 - pow(miss, (double)xo);
        double py = 1 - pow(miss, (double)yo);
        rs = px * py * (u32)(xo + yo) / (px + py - 2 * px * py);
    }
    u32 den = metric_sel == 0 ? X + Y - s : (X < Y ? X : Y); /* :1262-1263 */
    double m = ((double)s - rs) / den;
    double d = log(dist_arg(metric_sel, m)) / kmerlen; /* :1265 */
    if (d > 1)
        d = 1;
    memset(o, 0, sizeof *o);
    o->metric = m;
    o->dist = d;
    o->rs_u = (u32)rs;
    o->skipped = d > dthreshold; /* :1267 */
    double sd = pow(m * (1 - m) / den, 0.5); /* :1272 */
    o->pv = 0.5 * erfc(m / sd * pow(0.5, 0.5));
    o->fdr = o->pv * cmprsn_num;
    o->ci_m1 = m - 1.96 * sd; /* :1277-1280 */
    o->ci_m2 = m + 1.96 * sd;
    o->ci_d1 = log(dist_arg(metric_sel, o->ci_m2)) / kmerlen;
    o->ci_d2 = log(dist_arg(metric_sel, o->ci_m1)) / kmerlen;
}

/* the four metric values of many (X, Y, s) triples, rs = 0: the same expressions as above (command_dist.c:1262-1266),
 * evaluated with the host's libm; for whole-matrix comparisons */
void ko_metrics_batch(const u32 *X, const u32 *Y, const u32 *S, size_t n, int kmerlen, double *J, double *MD, double *Cc, double *AD)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const u32 x = X[i], y = Y[i], s = S[i];
        const double j = (double)s / (u32)(x + y - s), c = (double)s / (x < y ? x : y);
        double d = log(dist_arg(0, j)) / kmerlen;
        J[i] = j;
        MD[i] = d > 1 ? 1 : d;
        d = log(dist_arg(1, c)) / kmerlen;
        Cc[i] = c;
        AD[i] = d > 1 ? 1 : d;
    }
}

int ko_format_line(char *buf, size_t cap, const char *qname, const char *rname, u32 X, u32 Y, u32 s, int kmerlen,
                   int dim_rd_len, int metric_sel, int pfield, int correction, double dthreshold, u64 cmprsn_num)
{
    ko_metric o;
    ko_output_ctrl(X, Y, s, kmerlen, dim_rd_len, metric_sel, correction, dthreshold, cmprsn_num, &o);
    if (o.skipped)
        return 0;
    /* field formats of command_dist.c:1269-1285 */
    int len = snprintf(buf, cap, "%s\t%s\t%u-%u|%u|%u\t%.6lf\t%.6lf", qname, rname, s, o.rs_u, X, Y, o.metric, o.dist);
    if (pfield > 0)
        len += snprintf(buf + len, cap - (size_t)len, "\t%E\t%E", o.pv, o.fdr);
    if (pfield > 1)
        len += snprintf(buf + len, cap - (size_t)len, "\t[%.6lf,%.6lf]\t[%.6lf,%.6lf]", o.ci_m1, o.ci_m2, o.ci_d1,
                        o.ci_d2);
    len += snprintf(buf + len, cap - (size_t)len, "\n");
    return len;
}

int ko_dist_print(const char *path, const u32 *shared, int R, int Q, const u32 *ref_sz, const u32 *qry_sz,
                  const char *refnames, const char *qrynames, int kmerlen, int dim_rd_len, int metric_sel, int pfield,
                  int correction, double dthreshold, int n_max)
{
    static const char *cols[2][3] = {{"Jaccard\tMashD", "P-value(J)\tFDR(J)", "Jaccard_CI\tMashD_CI"},
                                     {"ContainmentM\tAafD", "P-value(C)\tFDR(C)", "ContainmentM_CI\tAafD_CI"}};
    if (n_max > 1024 || n_max > R)
        return KO_ERR_PARAM; /* command_dist.c:1198 */
    FILE *f = fopen(path, "w");
    if (!f)
        return KO_ERR_IO;
    fprintf(f, "Qry\tRef\tShared_k|Ref_s|Qry_s");
    for (int i = 0; i <= pfield; i++)
        fprintf(f, "\t%s", cols[metric_sel][i]);
    fprintf(f, "\n");
    u64 cmp = (u32)((u32)R * (u32)Q); /* 32-bit product, command_dist.c:1186 */
    char line[1024];
    double *bm = malloc(((size_t)n_max + 2) * sizeof(double));
    int *bi = malloc(((size_t)n_max + 2) * sizeof(int));
    for (int q = 0; q < Q; q++) {
        const char *qn = qrynames + (size_t)q * 256;
        const u32 *row = shared + (size_t)q * (size_t)R;
        u32 Y = qry_sz[q];
        if (n_max) { /* :1212-1227 keep the n_max largest raw metrics, ties keep the earlier ref */
            for (int i = 0; i < n_max; i++) {
                bm[i] = 0;
                bi[i] = -1;
            }
            for (int r = 0; r < R; r++) {
                u32 X = ref_sz[r], s = row[r];
                double m = metric_sel == 1 ? (double)s / (X < Y ? X : Y) : (double)s / (X + Y - s);
                for (int i = n_max - 1; i >= 0; i--) {
                    if (m > bm[i]) {
                        bm[i + 1] = bm[i];
                        bi[i + 1] = bi[i];
                        bm[i] = m;
                        bi[i] = r;
                    } else
                        break;
                }
            }
            for (int i = 0; i < n_max; i++) {
                if (bi[i] < 0)
                    continue;
                int len = ko_format_line(line, sizeof line, qn, refnames + (size_t)bi[i] * 256, ref_sz[bi[i]], Y,
                                         row[bi[i]], kmerlen, dim_rd_len, metric_sel, pfield, correction, dthreshold, cmp);
                if (len > 1)
                    fwrite(line, 1, (size_t)len, f);
            }
        } else {
            for (int r = 0; r < R; r++) {
                int len = ko_format_line(line, sizeof line, qn, refnames + (size_t)r * 256, ref_sz[r], Y, row[r],
                                         kmerlen, dim_rd_len, metric_sel, pfield, correction, dthreshold, cmp);
                if (len > 1)
                    fwrite(line, 1, (size_t)len, f);
            }
        }
    }
    free(bm);
    free(bi);
    fclose(f);
    return 0;
}
