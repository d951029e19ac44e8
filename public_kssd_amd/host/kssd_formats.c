/*
 * kssd_formats.c -- the reference's on-disk formats and its distance report.
 * Everything here is host C on purpose: these files are the interoperability contract with the reference
 * binary, and the report uses the host libm so that its text is the reference's text.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "kssd_host.h"

#define COMPONENT_SZ 7      /* reference Makefile:4 */
#define CTX_SPC_USE_L 8     /* global_basic.h:45-47 */
#define LD_FCTR 0.6         /* global_basic.h:49 */

/* hash-table sizes of the reference (global_basic.c:74-81): they fix the capacity rule and the dump order */
static const uint32_t primes[25] = {251u, 509u, 1021u, 2039u, 4093u, 8191u, 16381u, 32749u, 65521u, 131071u,
                                    262139u, 524287u, 1048573u, 2097143u, 4194301u, 8388593u, 16777213u,
                                    33554393u, 67108859u, 134217689u, 268435399u, 536870909u, 1073741789u,
                                    2147483647u, 4294967291u};

int kssd_derive(kssd_derived *d, int k, int subk, int drlevel)
{
    memset(d, 0, sizeof *d);
    if (k < subk || subk >= 8 || subk < 1 || drlevel < 0 || k > 15) return KSSD_HOST_ERR_PARAM;
    int pidx = 4 * (k - drlevel) - CTX_SPC_USE_L - 7;
    if (pidx < 0 || pidx > 24) return KSSD_HOST_ERR_PARAM;
    d->k = k;
    d->subk = subk;
    d->drlevel = drlevel;
    d->kmerlen = 2 * k;
    d->dim_rd_len = 2 * drlevel;
    int extra = k - drlevel - COMPONENT_SZ;
    d->comp_bits = extra > 0 ? 4 * extra : 0;
    d->comp_num = extra > 0 ? 1 << d->comp_bits : 1;
    d->hashsize = primes[pidx];
    d->hashlimit = (uint32_t)(d->hashsize * LD_FCTR);
    return KSSD_HOST_OK;
}

void kssd_sketchset_release(kssd_sketchset *s)
{
    if (!s) return;
    free(s->off);
    free(s->ids);
    free(s->names);
    free(s->counts);
    memset(s, 0, sizeof *s);
}

/* ---- slot order ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t slot, id;
} slot_id;

static int cmp_slot(const void *a, const void *b)
{
    uint32_t x = ((const slot_id *)a)->slot, y = ((const slot_id *)b)->slot;
    return x < y ? -1 : x > y;
}

typedef struct {
    uint32_t pos, id;
} pos_id;

static int cmp_pos(const void *a, const void *b)
{
    uint32_t x = ((const pos_id *)a)->pos, y = ((const pos_id *)b)->pos;
    return x < y ? -1 : x > y;
}

/* replay the reference's insertions, in the order given, into a sparse image of its table; ids come back in
 * ascending slot order */
static uint64_t slot_order_replay_keep(uint32_t *ids, const uint8_t *keep, uint64_t n, uint32_t hashsize);
static int slot_order_replay(uint32_t *ids, uint64_t n, uint32_t hashsize)
{
    return slot_order_replay_keep(ids, NULL, n, hashsize) == UINT64_MAX ? KSSD_HOST_ERR_NOMEM : 0;
}

/* keep != NULL: every id takes its slot, only the ids with keep[i] != 0 come back (their number is returned; UINT64_MAX:
 * out of memory, ids untouched) */
static uint64_t slot_order_replay_keep(uint32_t *ids, const uint8_t *keep, uint64_t n, uint32_t hashsize)
{
    uint64_t cap = 16;
    while (cap < 4 * n) cap <<= 1;
    uint32_t *occ = malloc(cap * sizeof(uint32_t)); /* open-addressing set of occupied slots, 0xFFFFFFFF = free */
    slot_id *sl = malloc((n ? n : 1) * sizeof(slot_id));
    if (!occ || !sl) { free(occ); free(sl); return UINT64_MAX; }
    uint64_t m = 0;
    memset(occ, 0xFF, cap * sizeof(uint32_t));
    const uint64_t S = hashsize;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t key = ids[i];
        const uint64_t h1 = key % S, h2 = 1 + key % (S - 1);
        for (uint64_t t = 0;; t++) {
            uint32_t slot = (uint32_t)((h1 + t * h2) % S);
            uint64_t p = ((uint64_t)slot * 0x9E3779B97F4A7C15ull) >> 32 & (cap - 1);
            int taken = 0;
            while (occ[p] != 0xFFFFFFFFu) {
                if (occ[p] == slot) { taken = 1; break; }
                p = (p + 1) & (cap - 1);
            }
            if (!taken) {
                occ[p] = slot;
                if (!keep || keep[i]) {
                    sl[m].slot = slot;
                    sl[m].id = ids[i];
                    m++;
                }
                break;
            }
        }
    }
    qsort(sl, m, sizeof(slot_id), cmp_slot);
    for (uint64_t i = 0; i < m; i++) ids[i] = sl[i].id;
    free(occ);
    free(sl);
    return m;
}

typedef struct {
    uint32_t pos;
    uint64_t key;
} pos_key;
typedef struct {
    uint32_t slot;
    uint64_t key;
} slot_key;
static int cmp_pos_key(const void *a, const void *b)
{
    uint32_t x = ((const pos_key *)a)->pos, y = ((const pos_key *)b)->pos;
    return x < y ? -1 : x > y;
}
static int cmp_slot_key(const void *a, const void *b)
{
    uint32_t x = ((const slot_key *)a)->slot, y = ((const slot_key *)b)->slot;
    return x < y ? -1 : x > y;
}

/* tuples beyond 32 bits: the same replay (insertions in sequence order into a sparse image of the reference's table) */
int kssd_slot_order_pos64(uint64_t *tuples, const uint32_t *first_pos, uint64_t n, uint32_t hashsize)
{
    if (n < 2) return 0;
    pos_key *pk = malloc(n * sizeof *pk);
    slot_key *sl = malloc(n * sizeof *sl);
    uint64_t cap = 16;
    while (cap < 4 * n) cap <<= 1;
    uint32_t *occ = malloc(cap * sizeof(uint32_t));
    if (!pk || !sl || !occ) { free(pk); free(sl); free(occ); return KSSD_HOST_ERR_NOMEM; }
    for (uint64_t i = 0; i < n; i++) { pk[i].pos = first_pos[i]; pk[i].key = tuples[i]; }
    qsort(pk, n, sizeof *pk, cmp_pos_key);
    memset(occ, 0xFF, cap * sizeof(uint32_t));
    const uint64_t S = hashsize;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t key = pk[i].key, h1 = key % S, h2 = 1 + key % (S - 1); /* global_basic.h:228-230 */
        for (uint64_t t = 0;; t++) {
            const uint32_t slot = (uint32_t)((h1 + t * h2) % S);
            uint64_t p = ((uint64_t)slot * 0x9E3779B97F4A7C15ull) >> 32 & (cap - 1);
            int taken = 0;
            while (occ[p] != 0xFFFFFFFFu) {
                if (occ[p] == slot) { taken = 1; break; }
                p = (p + 1) & (cap - 1);
            }
            if (!taken) {
                occ[p] = slot;
                sl[i].slot = slot;
                sl[i].key = key;
                break;
            }
        }
    }
    qsort(sl, n, sizeof *sl, cmp_slot_key);
    for (uint64_t i = 0; i < n; i++) tuples[i] = sl[i].key;
    free(pk);
    free(sl);
    free(occ);
    return 0;
}

/* the keep variant for tuples beyond 32 bits: every tuple takes its slot (insertions in sequence order), the ones with keep[i] != 0
 * come back at the front in file order; their number is returned (UINT64_MAX: out of memory) */
typedef struct {
    uint32_t pos;
    uint8_t keep;
    uint64_t key;
} pos_keep_key;
static int cmp_pos_keep_key(const void *a, const void *b)
{
    uint32_t x = ((const pos_keep_key *)a)->pos, y = ((const pos_keep_key *)b)->pos;
    return x < y ? -1 : x > y;
}
uint64_t kssd_slot_order_pos64_keep(uint64_t *tuples, const uint32_t *first_pos, const uint8_t *keep, uint64_t n, uint32_t hashsize)
{
    if (n == 0) return 0;
    pos_keep_key *pk = malloc(n * sizeof *pk);
    slot_key *sl = malloc(n * sizeof *sl);
    uint64_t cap = 16;
    while (cap < 4 * n) cap <<= 1;
    uint32_t *occ = malloc(cap * sizeof(uint32_t));
    if (!pk || !sl || !occ) { free(pk); free(sl); free(occ); return UINT64_MAX; }
    for (uint64_t i = 0; i < n; i++) { pk[i].pos = first_pos[i]; pk[i].keep = keep[i]; pk[i].key = tuples[i]; }
    qsort(pk, n, sizeof *pk, cmp_pos_keep_key); /* first positions are distinct: one k-mer per position */
    memset(occ, 0xFF, cap * sizeof(uint32_t));
    const uint64_t S = hashsize;
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t key = pk[i].key, h1 = key % S, h2 = 1 + key % (S - 1); /* global_basic.h:228-230 */
        for (uint64_t t = 0;; t++) {
            const uint32_t slot = (uint32_t)((h1 + t * h2) % S);
            uint64_t p = ((uint64_t)slot * 0x9E3779B97F4A7C15ull) >> 32 & (cap - 1);
            int taken = 0;
            while (occ[p] != 0xFFFFFFFFu) {
                if (occ[p] == slot) { taken = 1; break; }
                p = (p + 1) & (cap - 1);
            }
            if (!taken) {
                occ[p] = slot;
                if (pk[i].keep) {
                    sl[m].slot = slot;
                    sl[m].key = key;
                    m++;
                }
                break;
            }
        }
    }
    qsort(sl, m, sizeof *sl, cmp_slot_key);
    for (uint64_t i = 0; i < m; i++) tuples[i] = sl[i].key;
    free(pk);
    free(sl);
    free(occ);
    return m;
}

int kssd_slot_order(uint32_t *ids, uint64_t n, uint32_t hashsize)
{
    if (n < 2) return 0;
    return slot_order_replay(ids, n, hashsize); /* insertions in ascending id order */
}

int kssd_slot_order_pos(uint32_t *ids, const uint32_t *first_pos, uint64_t n, uint32_t hashsize)
{
    if (n < 2) return 0;
    pos_id *pi = malloc(n * sizeof(pos_id));
    if (!pi) return KSSD_HOST_ERR_NOMEM;
    for (uint64_t i = 0; i < n; i++) { pi[i].pos = first_pos[i]; pi[i].id = ids[i]; }
    qsort(pi, n, sizeof(pos_id), cmp_pos); /* first positions are distinct: one k-mer per position */
    for (uint64_t i = 0; i < n; i++) ids[i] = pi[i].id;
    free(pi);
    return slot_order_replay(ids, n, hashsize); /* insertions in sequence order, as fasta2co makes them */
}

/* all distinct ids of a genome with their first positions, and which of them the dump keeps: the dropped ones (fastq
 * -n: fewer than n occurrences, iseq2comem.c:336-345,516-518; -u: seen twice, :694-696,540-541) sit in the reference's
 * table all the same and shift the probes of later ids.  ids come back as the kept ones in the reference's file order. */
uint64_t kssd_slot_order_pos_keep(uint32_t *ids, const uint32_t *first_pos, const uint8_t *keep, uint64_t n, uint32_t hashsize)
{
    if (n == 0) return 0;
    pos_id *pi = malloc(n * sizeof(pos_id));
    uint8_t *kk = malloc(n);
    if (!pi || !kk) { free(pi); free(kk); return UINT64_MAX; }
    /* sort (position, index) so that the flags follow the ids */
    for (uint64_t i = 0; i < n; i++) { pi[i].pos = first_pos[i]; pi[i].id = (uint32_t)i; }
    qsort(pi, n, sizeof(pos_id), cmp_pos);
    uint32_t *tmp = malloc(n * sizeof(uint32_t));
    if (!tmp) { free(pi); free(kk); return UINT64_MAX; }
    for (uint64_t i = 0; i < n; i++) { tmp[i] = ids[pi[i].id]; kk[i] = keep[pi[i].id]; }
    memcpy(ids, tmp, n * sizeof(uint32_t));
    free(tmp);
    free(pi);
    const uint64_t m = slot_order_replay_keep(ids, kk, n, hashsize);
    free(kk);
    return m;
}

typedef struct {
    uint32_t id;
    uint16_t cnt;
} id_cnt;

static int cmp_id_cnt(const void *a, const void *b)
{
    uint32_t x = ((const id_cnt *)a)->id, y = ((const id_cnt *)b)->id;
    return x < y ? -1 : x > y;
}

int kssd_counts_follow(const uint32_t *ids_before, const uint16_t *counts_before, const uint32_t *ids_after,
                       uint16_t *counts_after, uint64_t n)
{
    if (!n) return KSSD_HOST_OK;
    id_cnt *t = malloc(n * sizeof(id_cnt));
    if (!t) return KSSD_HOST_ERR_NOMEM;
    int sorted = 1;
    for (uint64_t i = 0; i < n; i++) {
        t[i].id = ids_before[i];
        t[i].cnt = counts_before[i];
        if (i && ids_before[i] < ids_before[i - 1]) sorted = 0;
    }
    if (!sorted) qsort(t, n, sizeof(id_cnt), cmp_id_cnt);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if (t[mid].id < ids_after[i]) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= n || t[lo].id != ids_after[i]) { free(t); return KSSD_HOST_ERR_PARAM; }
        counts_after[i] = t[lo].cnt;
    }
    free(t);
    return KSSD_HOST_OK;
}

/* ---- stat files ---------------------------------------------------------------------------------------- */
static int write_stat(const kssd_sketchset *s, const char *dir, int mco)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, mco ? "mcofiles.stat" : "cofiles.stat");
    FILE *f = fopen(path, "wb");
    if (!f) return KSSD_HOST_ERR_IO;
    uint64_t total = s->off[s->n];
    if (mco) { /* mco_dstat_t, command_dist.h:57-64 */
        int32_t h[5] = {(int32_t)s->shuf_id, s->kmerlen, s->dim_rd_len, s->comp_num, (int32_t)s->n};
        fwrite(h, sizeof h, 1, f);
    } else { /* co_dstat_t, global_basic.h:94-103: 32 bytes with the bool padded to 4 */
        unsigned char h[32];
        memset(h, 0, sizeof h);
        memcpy(h + 0, &s->shuf_id, 4);
        h[4] = (unsigned char)(s->koc != 0);
        int32_t v[4] = {s->kmerlen, s->dim_rd_len, s->comp_num, (int32_t)s->n};
        memcpy(h + 8, v, 16);
        memcpy(h + 24, &total, 8);
        fwrite(h, sizeof h, 1, f);
    }
    for (uint32_t g = 0; g < s->n; g++) {
        uint32_t sz = (uint32_t)(s->off[g + 1] - s->off[g]);
        fwrite(&sz, 4, 1, f);
    }
    fwrite(s->names, KSSD_PATHLEN, s->n, f);
    return fclose(f) == 0 ? KSSD_HOST_OK : KSSD_HOST_ERR_IO;
}

static int read_stat(kssd_sketchset *s, const char *dir, int mco, uint32_t **sizes)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, mco ? "mcofiles.stat" : "cofiles.stat");
    FILE *f = fopen(path, "rb");
    if (!f) return KSSD_HOST_ERR_IO;
    memset(s, 0, sizeof *s);
    int ok = 1;
    if (mco) {
        int32_t h[5];
        ok = fread(h, sizeof h, 1, f) == 1;
        s->shuf_id = (uint32_t)h[0]; s->kmerlen = h[1]; s->dim_rd_len = h[2]; s->comp_num = h[3]; s->n = (uint32_t)h[4];
    } else {
        unsigned char h[32];
        ok = fread(h, sizeof h, 1, f) == 1;
        int32_t v[4];
        memcpy(&s->shuf_id, h, 4);
        s->koc = h[4];
        memcpy(v, h + 8, 16);
        s->kmerlen = v[0]; s->dim_rd_len = v[1]; s->comp_num = v[2]; s->n = (uint32_t)v[3];
    }
    if (!ok || s->comp_num < 1 || s->comp_num > 65536) { fclose(f); return KSSD_HOST_ERR_FORMAT; }
    if (s->comp_num > 16) { fclose(f); return KSSD_HOST_ERR_WIDE; } /* stored id (28 bits) + component: more than the 32 bits the sets hold */
    {   /* n comes from the file: it must agree with the file's size before anything is sized by it */
        struct stat st;
        const size_t hdr = mco ? 20 : 32;
        if (fstat(fileno(f), &st) != 0 || (uint64_t)st.st_size < hdr + (uint64_t)s->n * (4 + KSSD_PATHLEN)) { fclose(f); return KSSD_HOST_ERR_FORMAT; }
    }
    *sizes = malloc(((size_t)s->n + 1) * 4);
    s->names = malloc(((size_t)s->n + 1) * KSSD_PATHLEN);
    if (!*sizes || !s->names) { fclose(f); return KSSD_HOST_ERR_NOMEM; }
    ok = fread(*sizes, 4, s->n, f) == s->n && fread(s->names, KSSD_PATHLEN, s->n, f) == s->n;
    fclose(f);
    return ok ? KSSD_HOST_OK : KSSD_HOST_ERR_FORMAT;
}

int kssd_probe_dir(const char *dir)
{
    char p[4096];
    struct stat st;
    int r = 0;
    if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return 0;
    snprintf(p, sizeof p, "%s/cofiles.stat", dir);
    if (access(p, R_OK) == 0) r |= 1;
    snprintf(p, sizeof p, "%s/mcofiles.stat", dir);
    if (access(p, R_OK) == 0) r |= 2;
    return r;
}

/* ---- sketch container ---------------------------------------------------------------------------------- */
static int comp_bits_of(int comp_num)
{
    int b = 0;
    while ((1 << b) < comp_num) b++;
    return b;
}

int kssd_sketchset_write(const kssd_sketchset *s, const char *dir, uint32_t hashsize, int slot_order)
{
    mkdir(dir, 0777);
    if (s->sub && (slot_order || s->comp_num != 256)) return KSSD_HOST_ERR_PARAM; /* 36-bit tuples: in file order already */
    if (slot_order)
        for (uint32_t g = 0; g < s->n; g++) {
            const uint64_t n = s->off[g + 1] - s->off[g];
            uint32_t *before = NULL;
            uint16_t *cb4 = NULL;
            if (s->koc && s->counts && n) { /* the abundances travel with their ids */
                before = malloc(n * 4);
                cb4 = malloc(n * 2);
                if (!before || !cb4) { free(before); free(cb4); return KSSD_HOST_ERR_NOMEM; }
                memcpy(before, s->ids + s->off[g], n * 4);
                memcpy(cb4, s->counts + s->off[g], n * 2);
            }
            if (kssd_slot_order(s->ids + s->off[g], n, hashsize)) { free(before); free(cb4); return KSSD_HOST_ERR_NOMEM; }
            if (before) kssd_counts_follow(before, cb4, s->ids + s->off[g], s->counts + s->off[g], n);
            free(before);
            free(cb4);
        }
    const int cb = comp_bits_of(s->comp_num);
    const uint32_t cmask = (uint32_t)s->comp_num - 1u;
    char path[4096];
    uint64_t *idx = malloc(((size_t)s->n + 1) * sizeof(uint64_t));
    if (!idx) return KSSD_HOST_ERR_NOMEM;
    for (int c = 0; c < s->comp_num; c++) { /* command_dist.c:314-357 */
        snprintf(path, sizeof path, "%s/combco.%d", dir, c);
        FILE *f = fopen(path, "wb");
        if (!f) { free(idx); return KSSD_HOST_ERR_IO; }
        FILE *fa = NULL; /* abundances beside the ids, same order (command_dist.c:323-351) */
        if (s->koc && s->counts) {
            snprintf(path, sizeof path, "%s/combco.%d.a", dir, c);
            if (!(fa = fopen(path, "wb"))) { fclose(f); free(idx); return KSSD_HOST_ERR_IO; }
        }
        uint64_t run = 0;
        idx[0] = 0;
        for (uint32_t g = 0; g < s->n; g++) {
            if (s->comp_num == 1) {
                uint64_t cnt = s->off[g + 1] - s->off[g];
                fwrite(s->ids + s->off[g], 4, cnt, f);
                if (fa) fwrite(s->counts + s->off[g], 2, cnt, fa);
                run += cnt;
            } else {
                for (uint64_t i = s->off[g]; s->sub && i < s->off[g + 1]; i++) /* the tuple is ids[i] << 4 | sub[i] */
                    if (((((s->ids[i] & 15u) << 4) | s->sub[i]) & cmask) == (uint32_t)c) {
                        uint32_t id = s->ids[i] >> (cb - 4);
                        fwrite(&id, 4, 1, f);
                        if (fa) fwrite(s->counts + i, 2, 1, fa); /* write_fqkoc2files, iseq2comem.c:435-469 */
                        run++;
                    }
                for (uint64_t i = s->off[g]; !s->sub && i < s->off[g + 1]; i++)
                    if ((s->ids[i] & cmask) == (uint32_t)c) { /* drtuple % component_num, iseq2comem.c:543 */
                        uint32_t id = s->ids[i] >> cb;
                        fwrite(&id, 4, 1, f);
                        if (fa) fwrite(s->counts + i, 2, 1, fa);
                        run++;
                    }
            }
            idx[g + 1] = run;
        }
        if (fa && fclose(fa) != 0) { fclose(f); free(idx); return KSSD_HOST_ERR_IO; }
        if (fclose(f) != 0) { free(idx); return KSSD_HOST_ERR_IO; }
        snprintf(path, sizeof path, "%s/combco.index.%d", dir, c);
        f = fopen(path, "wb");
        if (!f) { free(idx); return KSSD_HOST_ERR_IO; }
        fwrite(idx, sizeof(uint64_t), (size_t)s->n + 1, f);
        if (fclose(f) != 0) { free(idx); return KSSD_HOST_ERR_IO; }
    }
    free(idx);
    return write_stat(s, dir, 0);
}

int kssd_byread_write(const char *dir, uint32_t shuf_id, int k, int drlevel, const char (*names)[KSSD_PATHLEN], uint32_t n_files,
                      const uint32_t *ids, const uint32_t *pos, uint64_t n, const uint64_t *read_start, uint64_t n_reads)
{
    mkdir(dir, 0777);
    const int extra = k - drlevel - 7; /* COMPONENT_SZ, iseq2comem.c:63-64,80 */
    const int cb = extra > 0 ? 4 * extra : 0;
    const int comp_num = extra > 0 ? 1 << cb : 1;
    const uint32_t cmask = (uint32_t)comp_num - 1u;
    /* read of every entry: number of cut points <= its position (both sequences ascend) */
    int64_t *cum = malloc(((size_t)n_reads + 1) * sizeof(int64_t));
    if (!cum) return KSSD_HOST_ERR_NOMEM;
    char path[4096];
    for (int c = 0; c < comp_num; c++) {
        snprintf(path, sizeof path, "%s/combco.%d", dir, c);
        FILE *f = fopen(path, "wb");
        if (!f) { free(cum); return KSSD_HOST_ERR_IO; }
        memset(cum, 0, ((size_t)n_reads + 1) * sizeof(int64_t));
        uint64_t r = 0;
        for (uint64_t i = 0; i < n; i++) {
            while (r < n_reads && read_start[r] <= pos[i]) r++;
            if ((ids[i] & cmask) != (uint32_t)c) continue; /* drtuple % component_num, iseq2comem.c:172-175 */
            const uint32_t id = ids[i] >> cb;
            fwrite(&id, 4, 1, f);
            cum[r]++;
        }
        if (fclose(f) != 0) { free(cum); return KSSD_HOST_ERR_IO; }
        for (uint64_t j = 1; j <= n_reads; j++) cum[j] += cum[j - 1]; /* iseq2comem.c:177-183: no leading zero */
        snprintf(path, sizeof path, "%s/combco.index.%d", dir, c);
        f = fopen(path, "wb");
        if (!f) { free(cum); return KSSD_HOST_ERR_IO; }
        fwrite(cum, sizeof(int64_t), (size_t)n_reads + 1, f);
        if (fclose(f) != 0) { free(cum); return KSSD_HOST_ERR_IO; }
    }
    free(cum);
    /* cofiles.stat as run_stageI writes it behind reads2mco: every input file listed, nothing counted
     * (command_dist.c:360-378) */
    kssd_sketchset s;
    memset(&s, 0, sizeof s);
    uint64_t *off = calloc((size_t)n_files + 1, sizeof(uint64_t));
    if (!off) return KSSD_HOST_ERR_NOMEM;
    s.shuf_id = shuf_id; s.kmerlen = 2 * k; s.dim_rd_len = 2 * drlevel; s.comp_num = comp_num; s.n = n_files;
    s.off = off; s.names = (char (*)[KSSD_PATHLEN])names;
    const int rc = write_stat(&s, dir, 0);
    free(off);
    return rc;
}

static void *slurp_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc((size_t)(n > 0 ? n : 1));
    if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *len = (size_t)(n > 0 ? n : 0);
    return p;
}

int kssd_sketchset_read(kssd_sketchset *s, const char *dir)
{
    uint32_t *sizes = NULL;
    int rc = read_stat(s, dir, 0, &sizes);
    if (rc) { free(sizes); kssd_sketchset_release(s); return rc; }
    s->off = malloc(((size_t)s->n + 1) * sizeof(uint64_t));
    if (!s->off) { free(sizes); kssd_sketchset_release(s); return KSSD_HOST_ERR_NOMEM; }
    s->off[0] = 0;
    for (uint32_t g = 0; g < s->n; g++) s->off[g + 1] = s->off[g] + sizes[g];
    free(sizes);
    s->ids = malloc((size_t)(s->off[s->n] ? s->off[s->n] : 1) * 4);
    uint64_t *fill = calloc((size_t)s->n + 1, sizeof(uint64_t));
    if (!s->ids || !fill) { free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_NOMEM; }
    if (s->koc) { /* abundances, if their files are around (the searches never need them, command_dist.c:880) */
        s->counts = calloc((size_t)(s->off[s->n] ? s->off[s->n] : 1), 2);
        if (!s->counts) { free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_NOMEM; }
    }
    const int cb = comp_bits_of(s->comp_num);
    char path[4096];
    for (int c = 0; c < s->comp_num; c++) {
        size_t li = 0, lc = 0;
        snprintf(path, sizeof path, "%s/combco.index.%d", dir, c);
        uint64_t *idx = slurp_file(path, &li);
        snprintf(path, sizeof path, "%s/combco.%d", dir, c);
        uint32_t *co = slurp_file(path, &lc);
        if (!idx || !co || li != ((size_t)s->n + 1) * 8 || lc != (size_t)idx[s->n] * 4) {
            free(idx); free(co); free(fill); kssd_sketchset_release(s);
            return KSSD_HOST_ERR_FORMAT;
        }
        uint16_t *ab = NULL;
        if (s->koc) {
            size_t la = 0;
            snprintf(path, sizeof path, "%s/combco.%d.a", dir, c);
            ab = slurp_file(path, &la);
            if (ab && la != (size_t)idx[s->n] * 2) { free(ab); ab = NULL; }
            if (!ab) { free(s->counts); s->counts = NULL; }
        }
        for (uint32_t g = 0; g < s->n; g++)
            for (uint64_t i = idx[g]; i < idx[g + 1]; i++) {
                uint64_t w = s->off[g] + fill[g]++;
                if (w >= s->off[g + 1]) { free(idx); free(co); free(ab); free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_FORMAT; }
                s->ids[w] = (co[i] << cb) | (uint32_t)c;
                if (ab && s->counts) s->counts[w] = ab[i];
            }
        free(idx);
        free(co);
        free(ab);
    }
    free(fill);
    return KSSD_HOST_OK;
}

/* ---- inverted index files ---------------------------------------------------------------------------- */
typedef struct {
    uint32_t id, gid;
} id_gid;

static int cmp_id_gid(const void *a, const void *b)
{
    const id_gid *x = a, *y = b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return x->gid < y->gid ? -1 : x->gid > y->gid;
}

int kssd_index_write(const kssd_sketchset *s, const char *dir)
{
    mkdir(dir, 0777);
    const int cb = comp_bits_of(s->comp_num);
    const uint32_t cmask = (uint32_t)s->comp_num - 1u;
    const uint64_t comp_sz = 1ull << (4 * COMPONENT_SZ);
    const uint64_t total = s->off[s->n];
    id_gid *pairs = malloc((size_t)(total ? total : 1) * sizeof(id_gid));
    const size_t BLK = 1u << 20;
    uint64_t *blk = malloc(BLK * sizeof(uint64_t));
    if (!pairs || !blk) { free(pairs); free(blk); return KSSD_HOST_ERR_NOMEM; }
    char path[4096];
    int rc = KSSD_HOST_OK;
    for (int c = 0; c < s->comp_num && rc == KSSD_HOST_OK; c++) {
        uint64_t np = 0;
        for (uint32_t g = 0; g < s->n; g++)
            for (uint64_t i = s->off[g]; i < s->off[g + 1]; i++)
                if ((s->ids[i] & cmask) == (uint32_t)c) {
                    pairs[np].id = s->ids[i] >> cb;
                    pairs[np].gid = g;
                    np++;
                }
        qsort(pairs, np, sizeof(id_gid), cmp_id_gid);
        /* mco.index.<c>: inclusive prefix sums of the posting lengths over the whole 16^7 id space (co2mco.c:57-62) */
        snprintf(path, sizeof path, "%s/mco.index.%d", dir, c);
        FILE *f = fopen(path, "wb");
        if (!f) { rc = KSSD_HOST_ERR_IO; break; }
        uint64_t p = 0;
        for (uint64_t base = 0; base < comp_sz; base += BLK) {
            for (size_t j = 0; j < BLK; j++) {
                const uint64_t id = base + j;
                while (p < np && pairs[p].id <= id) p++;
                blk[j] = p;
            }
            if (fwrite(blk, sizeof(uint64_t), BLK, f) != BLK) { rc = KSSD_HOST_ERR_IO; break; }
        }
        if (fclose(f) != 0) rc = KSSD_HOST_ERR_IO;
        if (rc) break;
        /* mco.<c>: the postings, ascending genome index inside each (co2mco.c:63-72) */
        snprintf(path, sizeof path, "%s/mco.%d", dir, c);
        f = fopen(path, "wb");
        if (!f) { rc = KSSD_HOST_ERR_IO; break; }
        for (uint64_t i = 0; i < np; i++) fwrite(&pairs[i].gid, 4, 1, f);
        if (fclose(f) != 0) rc = KSSD_HOST_ERR_IO;
    }
    free(pairs);
    free(blk);
    if (rc) return rc;
    return write_stat(s, dir, 1);
}

int kssd_index_read(kssd_sketchset *s, const char *dir)
{
    uint32_t *sizes = NULL;
    int rc = read_stat(s, dir, 1, &sizes);
    if (rc) { free(sizes); kssd_sketchset_release(s); return rc; }
    s->off = malloc(((size_t)s->n + 1) * sizeof(uint64_t));
    if (!s->off) { free(sizes); kssd_sketchset_release(s); return KSSD_HOST_ERR_NOMEM; }
    s->off[0] = 0;
    for (uint32_t g = 0; g < s->n; g++) s->off[g + 1] = s->off[g] + sizes[g];
    free(sizes);
    s->ids = malloc((size_t)(s->off[s->n] ? s->off[s->n] : 1) * 4);
    uint64_t *fill = calloc((size_t)s->n + 1, sizeof(uint64_t));
    if (!s->ids || !fill) { free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_NOMEM; }
    const int cb = comp_bits_of(s->comp_num);
    const uint64_t comp_sz = 1ull << (4 * COMPONENT_SZ);
    char path[4096];
    for (int c = 0; c < s->comp_num; c++) {
        snprintf(path, sizeof path, "%s/mco.index.%d", dir, c);
        int fd = open(path, O_RDONLY);
        if (fd < 0) { free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_IO; }
        struct stat st;
        fstat(fd, &st);
        if ((uint64_t)st.st_size != comp_sz * 8) { close(fd); free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_FORMAT; }
        const uint64_t *idx = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        size_t lp = 0;
        snprintf(path, sizeof path, "%s/mco.%d", dir, c);
        uint32_t *post = slurp_file(path, &lp);
        if (idx == MAP_FAILED || !post) { free(post); free(fill); kssd_sketchset_release(s); return KSSD_HOST_ERR_IO; }
        uint64_t prev = 0;
        for (uint64_t id = 0; id < comp_sz; id++) {
            const uint64_t end = idx[id];
            for (uint64_t p = prev; p < end && p * 4 < lp; p++) {
                const uint32_t g = post[p];
                if (g < s->n && s->off[g] + fill[g] < s->off[g + 1]) s->ids[s->off[g] + fill[g]++] = ((uint32_t)id << cb) | (uint32_t)c;
            }
            prev = end;
        }
        munmap((void *)idx, (size_t)st.st_size);
        free(post);
    }
    free(fill);
    return KSSD_HOST_OK;
}

/* ---- distance report ------------------------------------------------------------------------------------ */
static inline double dist_arg(int metric, double m) { return metric == 0 ? 1 / (2 * m) + 0.5 : 1 / m; }

/*
 * Number formatting of the report.  The reference prints every number of a line with snprintf ("%.6lf", "%E",
 * command_dist.c:1269-1286); the C library formats a double through arbitrary-precision arithmetic, and with eight
 * numbers per line that was most of the report's time.  The two formats are produced here exactly as the library
 * produces them -- the digits of the double's exact binary value, rounded to nearest, ties to even -- by integer
 * arithmetic: "%.6lf" exactly (128-bit product of the mantissa and 10^6), "%E" through an 80-bit scaling whose error
 * is bounded and a hand-over to snprintf whenever the scaled value is within 2^-20 of a rounding boundary (one value
 * in a million).  Anything unusual (nan, inf, subnormals, |x| >= 1e12 for "%.6lf") goes to snprintf as well.  The
 * arithmetic that produces the numbers (log, pow, erfc) stays the host libm's.
 */
static inline char *put_uint(char *p, uint64_t v)
{
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
}
static inline char *put_digits(char *p, uint32_t v, int n) /* exactly n digits, zero padded */
{
    for (int i = n - 1; i >= 0; i--) { p[i] = (char)('0' + v % 10); v /= 10; }
    return p + n;
}

char *kssd_fmt_f6(char *p, double x) /* "%.6lf" */
{
    uint64_t b;
    memcpy(&b, &x, 8);
    const int ef = (int)((b >> 52) & 0x7FF);
    if (ef == 0x7FF || ef >= 1023 + 39) return p + sprintf(p, "%.6lf", x); /* nan, inf, |x| >= 2^39 */
    if (b >> 63) *p++ = '-';
    uint64_t M = b & 0x000FFFFFFFFFFFFFull;
    int E;
    if (ef) { M |= 1ull << 52; E = ef - 1075; } else E = -1074;
    /* |x| * 10^6 = M * 10^6 * 2^E, E < 0: integer part and remainder of a 128-bit product */
    const unsigned __int128 P = (unsigned __int128)M * 1000000u;
    const int sh = -E;
    uint64_t N = 0;
    if (sh < 128) {
        N = (uint64_t)(P >> sh);
        const unsigned __int128 rem = P & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
        if (rem > half || (rem == half && (N & 1))) N++;
    }
    p = put_uint(p, N / 1000000u);
    *p++ = '.';
    return put_digits(p, (uint32_t)(N % 1000000u), 6);
}

static long double pow10_tab[700]; /* 10^(i - 340) */
static int pow10_ready;
static void pow10_init(void)
{
#pragma omp critical(kssd_pow10)
    {
        if (!pow10_ready) {
            for (int i = 0; i < 700; i++) pow10_tab[i] = powl(10.0L, (long double)(i - 340));
            __atomic_store_n(&pow10_ready, 1, __ATOMIC_RELEASE);
        }
    }
}

char *kssd_fmt_e6(char *p, double x) /* "%E" */
{
    uint64_t b;
    memcpy(&b, &x, 8);
    const int ef = (int)((b >> 52) & 0x7FF);
    if (ef == 0x7FF || (ef == 0 && (b << 1))) return p + sprintf(p, "%E", x); /* nan, inf, subnormal */
    if (b >> 63) *p++ = '-';
    if (ef == 0) { memcpy(p, "0.000000E+00", 12); return p + 12; }
    if (!__atomic_load_n(&pow10_ready, __ATOMIC_ACQUIRE)) pow10_init();
    const long double ax = (long double)fabs(x);
    const int e2 = ef - 1023;                                      /* |x| = m * 2^e2, 1 <= m < 2 */
    int d = (int)floor((double)e2 * 0.30102999566398120);          /* floor(log10 |x|) or one less */
    long double s = ax * pow10_tab[6 - d + 340];                   /* in [10^6, 10^8): 7 or 8 digits in front of the point */
    int unsure = fabsl(s - 9999999.5L) < 0x1p-20L;                 /* "9.999999" or "1.000000" of the next decade? */
    if (!unsure && s > 9999999.5L) {
        d++;
        s = ax * pow10_tab[6 - d + 340];                           /* in [10^6 - 0.05, 10^7) */
    }
    const long double fl = floorl(s), fr = s - fl;
    if (unsure || fabsl(fr - 0.5L) < 0x1p-20L || s < 999999.9L) { /* too close to a tie for the 80-bit product: the library decides */
        if (b >> 63) p--;
        return p + sprintf(p, "%E", x);
    }
    uint32_t N = (uint32_t)fl + (fr > 0.5L ? 1u : 0u);
    if (N >= 10000000u) { N = 1000000u; d++; }
    *p++ = (char)('0' + N / 1000000u);
    *p++ = '.';
    p = put_digits(p, N % 1000000u, 6);
    *p++ = 'E';
    *p++ = d < 0 ? '-' : '+';
    const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
    return ad >= 100 ? put_digits(p, ad, 3) : put_digits(p, ad, 2);
}

/* what a line without shared k-mers looks like behind its names and sizes: the same bytes for every such pair of one
 * report (metric 0, distance 1, "-NAN" p-values, "[inf,inf]"), formatted once by the library */
typedef struct {
    int len;
    char txt[160];
    double sqrt_half; /* pow(0.5, 0.5) of this libm, computed once per report */
} zero_tail;

/* one line of distance.out exactly as output_ctrl writes it (command_dist.c:1251-1287): the library's bounded formatting.
 * Used for lines whose names leave no room for the fast path and, once per report, for the tail of the lines without
 * shared k-mers.  force: skip the -D test (the tail is wanted whatever the threshold) */
static int format_line_lib(char *buf, size_t cap, const char *qn, const char *rn, uint32_t X, uint32_t Y, uint32_t s, int kmerlen,
                           int dim_rd_len, const kssd_print_opt *o, uint64_t cmprsn, int force)
{
    double rs = 0;
    if (o->correction) {
        uint32_t xo = X - s, yo = Y - s;
        double miss = 1 - 1 / pow((double)4, (double)(kmerlen - dim_rd_len));
        double px = 1 - pow(miss, (double)xo), py = 1 - pow(miss, (double)yo);
        rs = px * py * (uint32_t)(xo + yo) / (px + py - 2 * px * py);
    }
    uint32_t den = o->metric == 0 ? X + Y - s : (X < Y ? X : Y);
    double m = ((double)s - rs) / den;
    double d = log(dist_arg(o->metric, m)) / kmerlen;
    if (d > 1) d = 1;
    if (!force && d > o->dthreshold) return 0;
    int len = snprintf(buf, cap, "%s\t%s\t%u-%u|%u|%u\t%.6lf\t%.6lf", qn, rn, s, (uint32_t)rs, X, Y, m, d);
    if ((size_t)len >= cap) len = (int)cap - 1;
    if (o->pfield > 0) {
        double sd = pow(m * (1 - m) / den, 0.5);
        double pv = 0.5 * erfc(m / sd * pow(0.5, 0.5));
        len += snprintf(buf + len, cap - (size_t)len, "\t%E\t%E", pv, pv * cmprsn);
        if ((size_t)len >= cap) len = (int)cap - 1; /* truncated (names of 255 bytes each leave room; guard anyway) */
        if (o->pfield > 1) {
            double m1 = m - 1.96 * sd, m2 = m + 1.96 * sd;
            double d1 = log(dist_arg(o->metric, m2)) / kmerlen, d2 = log(dist_arg(o->metric, m1)) / kmerlen;
            len += snprintf(buf + len, cap - (size_t)len, "\t[%.6lf,%.6lf]\t[%.6lf,%.6lf]", m1, m2, d1, d2);
            if ((size_t)len >= cap) len = (int)cap - 1;
        }
    }
    len += snprintf(buf + len, cap - (size_t)len, "\n");
    if ((size_t)len >= cap) len = (int)cap - 1;
    return len;
}

/* one line of distance.out, the arithmetic and the formats of output_ctrl (command_dist.c:1251-1287) */
static int format_line(char *buf, size_t cap, const char *qn, size_t qlen, const char *rn, size_t rlen, uint32_t X, uint32_t Y, uint32_t s,
                       int kmerlen, int dim_rd_len, const kssd_print_opt *o, uint64_t cmprsn, const zero_tail *zt)
{
    double rs = 0;
    if (o->correction) {
        uint32_t xo = X - s, yo = Y - s;
        double miss = 1 - 1 / pow((double)4, (double)(kmerlen - dim_rd_len));
        double px = 1 - pow(miss, (double)xo), py = 1 - pow(miss, (double)yo);
        rs = px * py * (uint32_t)(xo + yo) / (px + py - 2 * px * py);
    }
    uint32_t den = o->metric == 0 ? X + Y - s : (X < Y ? X : Y);
    char *p = buf;
    const int roomy = qlen + rlen + 220 <= cap;
    if (roomy && zt && s == 0 && den != 0 && !o->correction) { /* nothing shared: the constant tail */
        if (1.0 > o->dthreshold) return 0;
        memcpy(p, qn, qlen); p += qlen; *p++ = '\t';
        memcpy(p, rn, rlen); p += rlen; *p++ = '\t';
        memcpy(p, "0-0|", 4); p += 4;
        p = put_uint(p, X); *p++ = '|';
        p = put_uint(p, Y);
        memcpy(p, zt->txt, (size_t)zt->len);
        return (int)(p - buf) + zt->len;
    }
    double m = ((double)s - rs) / den;
    double d = log(dist_arg(o->metric, m)) / kmerlen;
    if (d > 1) d = 1;
    if (d > o->dthreshold) return 0;
    if (!roomy) return format_line_lib(buf, cap, qn, rn, X, Y, s, kmerlen, dim_rd_len, o, cmprsn, 0);
    memcpy(p, qn, qlen); p += qlen; *p++ = '\t';
    memcpy(p, rn, rlen); p += rlen; *p++ = '\t';
    p = put_uint(p, s); *p++ = '-';
    p = put_uint(p, (uint32_t)rs); *p++ = '|';
    p = put_uint(p, X); *p++ = '|';
    p = put_uint(p, Y); *p++ = '\t';
    p = kssd_fmt_f6(p, m); *p++ = '\t';
    p = kssd_fmt_f6(p, d);
    if (o->pfield > 0) {
        const double sqrt_half = zt ? zt->sqrt_half : pow(0.5, 0.5); /* command_dist.c:1273 */
        double sd = pow(m * (1 - m) / den, 0.5);
        double pv = 0.5 * erfc(m / sd * sqrt_half);
        *p++ = '\t';
        p = kssd_fmt_e6(p, pv); *p++ = '\t';
        p = kssd_fmt_e6(p, pv * cmprsn);
        if (o->pfield > 1) {
            double m1 = m - 1.96 * sd, m2 = m + 1.96 * sd;
            double d1 = log(dist_arg(o->metric, m2)) / kmerlen, d2 = log(dist_arg(o->metric, m1)) / kmerlen;
            *p++ = '\t'; *p++ = '[';
            p = kssd_fmt_f6(p, m1); *p++ = ',';
            p = kssd_fmt_f6(p, m2); *p++ = ']'; *p++ = '\t'; *p++ = '[';
            p = kssd_fmt_f6(p, d1); *p++ = ',';
            p = kssd_fmt_f6(p, d2); *p++ = ']';
        }
    }
    *p++ = '\n';
    return (int)(p - buf);
}

typedef struct {
    char *p;
    size_t n, cap;
    int failed; /* out of memory while growing */
} strbuf;

static void sb_add(strbuf *b, const char *s, size_t n)
{
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1 << 16;
        while (nc < b->n + n) nc *= 2;
        char *q = realloc(b->p, nc);
        if (!q) { b->failed = 1; return; }
        b->p = q;
        b->cap = nc;
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
}

/* rows either dense (shared != NULL: Q x R counts) or as the pairs a selection left (poff/pr/ps: CSR over the queries,
 * references ascending inside a row) */
static int print_rows(const char *path, const uint32_t *shared, const uint64_t *poff, const uint32_t *pr, const uint32_t *ps,
                      const kssd_sketchset *ref, const kssd_sketchset *qry, const kssd_print_opt *o);

int kssd_distance_print(const char *path, const uint32_t *shared, const kssd_sketchset *ref, const kssd_sketchset *qry,
                        const kssd_print_opt *o)
{
    return print_rows(path, shared, NULL, NULL, NULL, ref, qry, o);
}

int kssd_distance_print_pairs(const char *path, const uint64_t *pair_off, const uint32_t *pair_ref, const uint32_t *pair_shared,
                              const kssd_sketchset *ref, const kssd_sketchset *qry, const kssd_print_opt *o)
{
    if (!pair_off) return KSSD_HOST_ERR_PARAM;
    return print_rows(path, NULL, pair_off, pair_ref, pair_shared, ref, qry, o);
}

static int print_rows(const char *path, const uint32_t *shared, const uint64_t *poff, const uint32_t *pr, const uint32_t *ps,
                      const kssd_sketchset *ref, const kssd_sketchset *qry, const kssd_print_opt *o)
{
    static const char *cols[2][3] = {{"Jaccard\tMashD", "P-value(J)\tFDR(J)", "Jaccard_CI\tMashD_CI"},
                                     {"ContainmentM\tAafD", "P-value(C)\tFDR(C)", "ContainmentM_CI\tAafD_CI"}};
    const uint32_t R = ref->n, Q = qry->n;
    if (o->n_max > 1024 || (uint32_t)o->n_max > R) return KSSD_HOST_ERR_PARAM; /* command_dist.c:1198 */
    FILE *f = fopen(path, "w");
    if (!f) return KSSD_HOST_ERR_IO;
    fprintf(f, "Qry\tRef\tShared_k|Ref_s|Qry_s");
    for (int i = 0; i <= o->pfield; i++) fprintf(f, "\t%s", cols[o->metric][i]);
    fprintf(f, "\n");
    const uint64_t cmprsn = (uint32_t)(R * Q); /* 32-bit product like command_dist.c:1186 */
    const int kmerlen = qry->kmerlen, drl = qry->dim_rd_len;
    int threads = o->threads > 0 ? o->threads : 1;
    /* the tail of every line without shared k-mers, from the library's own formatting of one such pair */
    zero_tail zt;
    {
        volatile double half = 0.5;
        zt.sqrt_half = pow(half, half);
        char tmp[512];
        const int n = format_line_lib(tmp, sizeof tmp, "", "", 1, 1, 0, kmerlen, drl, o, cmprsn, 1);
        static const char pre[] = "\t\t0-0|1|1";
        zt.len = 0;
        if (n > (int)sizeof pre - 1 && n - (int)(sizeof pre - 1) < (int)sizeof zt.txt && memcmp(tmp, pre, sizeof pre - 1) == 0) {
            zt.len = n - (int)(sizeof pre - 1);
            memcpy(zt.txt, tmp + sizeof pre - 1, (size_t)zt.len);
        }
    }
    uint32_t *rlen = malloc(((size_t)R + 1) * sizeof *rlen);
    const uint32_t QB = 64; /* queries formatted per parallel batch, written back in order */
    strbuf *sb = calloc(2 * (size_t)QB, sizeof(strbuf)); /* two batches: one is written while the next one is formatted */
    int nomem = 0, ioerr = 0;
    if (!sb || !rlen) { free(sb); free(rlen); fclose(f); return KSSD_HOST_ERR_NOMEM; }
    for (uint32_t r = 0; r < R; r++) rlen[r] = (uint32_t)strlen(ref->names[r]);
    fflush(f);
#pragma omp parallel num_threads(threads)
    {
        int par = 0;
        for (uint32_t q0 = 0; q0 < Q; q0 += QB, par ^= 1) {
            const uint32_t q1 = q0 + QB < Q ? q0 + QB : Q;
            strbuf *bat = sb + (size_t)par * QB;
#pragma omp for schedule(dynamic, 1) nowait
            for (uint32_t q = q0; q < q1; q++) {
                strbuf *b = &bat[q - q0];
                b->n = 0;
                char line[1024];
                const char *qn = qry->names[q];
                const size_t qlen = strlen(qn);
                const uint32_t *row = shared ? shared + (size_t)q * R : NULL;
                const uint64_t nrow = row ? R : poff[q + 1] - poff[q]; /* entries of this row */
                const uint32_t *rr = row ? NULL : pr + poff[q], *rs = row ? NULL : ps + poff[q];
#define ROW_REF(i) (row ? (uint32_t)(i) : rr[i])
#define ROW_SHARED(i) (row ? row[i] : rs[i])
                const uint32_t Y = (uint32_t)(qry->off[q + 1] - qry->off[q]);
                if (o->n_max) { /* -N: the n_max largest raw metrics, earlier reference wins ties (:1212-1227) */
                    double bm[1026];
                    int bi[1026];
                    for (int i = 0; i < o->n_max; i++) { bm[i] = 0; bi[i] = -1; }
                    uint32_t bs[1026];
                    for (uint64_t e = 0; e < nrow; e++) {
                        const uint32_t r = ROW_REF(e), s = ROW_SHARED(e);
                        const uint32_t X = (uint32_t)(ref->off[r + 1] - ref->off[r]);
                        const double m = o->metric == 1 ? (double)s / (X < Y ? X : Y) : (double)s / (X + Y - s);
                        for (int i = o->n_max - 1; i >= 0; i--) {
                            if (m > bm[i]) { bm[i + 1] = bm[i]; bi[i + 1] = bi[i]; bs[i + 1] = bs[i]; bm[i] = m; bi[i] = (int)r; bs[i] = s; }
                            else break;
                        }
                    }
                    for (int i = 0; i < o->n_max; i++) {
                        if (bi[i] < 0) continue;
                        const uint32_t r = (uint32_t)bi[i];
                        int len = format_line(line, sizeof line, qn, qlen, ref->names[r], rlen[r],
                                              (uint32_t)(ref->off[r + 1] - ref->off[r]), Y, bs[i], kmerlen, drl, o, cmprsn, &zt);
                        if (len > 1) sb_add(b, line, (size_t)len);
                    }
                } else {
                    for (uint64_t e = 0; e < nrow; e++) {
                        const uint32_t r = ROW_REF(e);
                        int len = format_line(line, sizeof line, qn, qlen, ref->names[r], rlen[r],
                                              (uint32_t)(ref->off[r + 1] - ref->off[r]), Y, ROW_SHARED(e), kmerlen, drl, o, cmprsn, &zt);
                        if (len > 1) sb_add(b, line, (size_t)len);
                    }
                }
#undef ROW_REF
#undef ROW_SHARED
            }
            /* everybody has formatted its rows of this batch -- and the batch before it has been written: the thread that
             * wrote it cannot have arrived here before it was done */
#pragma omp barrier
#pragma omp single nowait
            {   /* one thread writes the batch, the others go on with the next one (the other set of buffers) */
                for (uint32_t q = q0; q < q1; q++) {
                    if (bat[q - q0].failed) nomem = 1;
                    if (bat[q - q0].n && fwrite(bat[q - q0].p, 1, bat[q - q0].n, f) != bat[q - q0].n) ioerr = 1;
                }
            }
        }
    }
    for (uint32_t i = 0; i < 2 * QB; i++) free(sb[i].p);
    free(sb);
    free(rlen);
    if (nomem) { fclose(f); return KSSD_HOST_ERR_NOMEM; }
    if (fclose(f) != 0 || ioerr) return KSSD_HOST_ERR_IO;
    return KSSD_HOST_OK;
}

/* ---- kssd reverse ------------------------------------------------------------------------------------- */
/* An id back to its canonical k-mer -- the inverse of the sketch's reduction (kssd_core.h kssd_s2_tuple; the reference:
 * core_reverse2unituple, command_reverse.c:311-321).  Forwards, a canonical 2k-mer  L | M | R  (L, R: the k - subk bases on
 * either side, M: the 2 subk bases of the sub-context) became   id = (((L | R) << 4 subk) >> 4 drlevel) + rank(M).
 * Backwards: the rank is the id modulo 4096 (MIN_SUBCTX_DIM_SMP_SZ -- the reference's assumption, exact whenever
 * 4 (subk - drlevel) <= 12; `kssd reverse` refuses other shuffles), M = accepted[rank], L | R = what stands above the
 * 4 (subk - drlevel) rank bits, and the three pieces go back to their places. */
uint64_t kssd_reverse_id(uint32_t full_id, int k, int subk, int drlevel, const uint32_t *accepted)
{
    const int side = 2 * (k - subk);      /* bits of L, and of R */
    const int mid = 4 * subk;             /* bits of M */
    const uint64_t M = accepted[full_id % 4096u];
    const uint64_t LR = (uint64_t)full_id >> (4 * (subk - drlevel));
    const uint64_t L = LR >> side, R = LR & ((1ull << side) - 1ull);
    return (L << (mid + side)) + (M << side) + R;
}

int kssd_shuf_accepted(const kssd_shuf *s, uint32_t *accepted)
{
    const uint64_t n = 1ull << (4 * s->subk);
    uint32_t count = 0;
    for (uint64_t i = 0; i < n; i++)
        if (s->table[i] >= 0 && s->table[i] < 4096) { accepted[s->table[i]] = (uint32_t)i; count++; }
    return count == 4096 ? KSSD_HOST_OK : KSSD_HOST_ERR_FORMAT; /* "count %d not match MIN_SUBCTX_DIM_SMP_SZ", :232 */
}

int kssd_reverse_dir(const kssd_shuf *sh, const char *sketch_dir, const char *outdir)
{
    uint32_t accepted[4096];
    int rc = kssd_shuf_accepted(sh, accepted);
    if (rc) return rc;
    kssd_sketchset s;
    memset(&s, 0, sizeof s);
    rc = kssd_sketchset_read(&s, sketch_dir); /* ids come back component after component, file order inside each */
    if (rc) return rc;
    const int TL = 2 * sh->k;
    char path[4096], kstring[64];
    kstring[TL] = '\0';
    for (uint32_t g = 0; g < s.n && rc == KSSD_HOST_OK; g++) {
        if (s.off[g + 1] == s.off[g]) continue; /* the reference writes nothing for an empty sketch (:290) */
        const char *nm = strrchr(s.names[g], '/');
        nm = nm ? nm + 1 : s.names[g];
        snprintf(path, sizeof path, "%s/%s", outdir, nm);
        FILE *f = fopen(path, "w");
        if (!f) { rc = KSSD_HOST_ERR_IO; break; }
        for (uint64_t i = s.off[g]; i < s.off[g + 1]; i++) {
            uint64_t u = kssd_reverse_id(s.ids[i], sh->k, sh->subk, sh->drlevel, accepted);
            for (int b = 0; b < TL; b++) { kstring[TL - b - 1] = "ACGT"[u & 3u]; u >>= 2; }
            fprintf(f, "%s\n", kstring);
        }
        if (fclose(f) != 0) rc = KSSD_HOST_ERR_IO;
    }
    kssd_sketchset_release(&s);
    return rc;
}

/* co_rvs2kmer_byreads (command_reverse.c:147-218): the reads of a --byread sketch directory as text on `out`.
 * ">read n" for n = 1..readn (readn = entries of combco.index.0 minus one), under each the k-mers of component 0, 1, ...
 * The reference takes the COUNT of read n from index[n] - index[n-1] but reads the ids sequentially from the start of
 * combco.<c>: k-mers in front of the first header (index[0] of them) shift every read's list by that many entries.
 * Restated as it behaves. */
int kssd_reverse_byreads(const kssd_shuf *sh, const char *sketch_dir, FILE *out)
{
    uint32_t accepted[4096];
    int rc = kssd_shuf_accepted(sh, accepted);
    if (rc) return rc;
    const int extra = sh->k - sh->drlevel - 7;
    const int cb = extra > 0 ? 4 * extra : 0;
    char path[4096];
    snprintf(path, sizeof path, "%s/cofiles.stat", sketch_dir);
    size_t hl = 0;
    unsigned char *hdr = slurp_file(path, &hl);
    if (!hdr || hl < 32) { free(hdr); return KSSD_HOST_ERR_IO; }
    int32_t comp_num;
    memcpy(&comp_num, hdr + 16, 4);
    free(hdr);
    if (comp_num < 1 || comp_num > 65536) return KSSD_HOST_ERR_FORMAT;
    uint64_t **idx = calloc((size_t)comp_num, sizeof *idx);
    uint32_t **ids = calloc((size_t)comp_num, sizeof *ids);
    size_t *n_ids = calloc((size_t)comp_num, sizeof *n_ids), *cursor = calloc((size_t)comp_num, sizeof *cursor);
    uint64_t readn = 0;
    rc = KSSD_HOST_OK;
    for (int j = 0; j < comp_num && rc == KSSD_HOST_OK; j++) {
        size_t il = 0, xl = 0;
        snprintf(path, sizeof path, "%s/combco.index.%d", sketch_dir, j);
        idx[j] = slurp_file(path, &xl);
        snprintf(path, sizeof path, "%s/combco.%d", sketch_dir, j);
        ids[j] = slurp_file(path, &il);
        if (!idx[j] || !ids[j] || xl < 8) { rc = KSSD_HOST_ERR_IO; break; }
        if (j == 0) readn = xl / 8 - 1; /* :184-185 */
        else if (xl / 8 - 1 < readn) rc = KSSD_HOST_ERR_FORMAT;
        n_ids[j] = il / 4;
    }
    const int TL = 2 * sh->k;
    char kstring[64];
    kstring[TL] = '\0';
    for (uint64_t n = 0; n < readn && rc == KSSD_HOST_OK; n++) {
        fprintf(out, ">read %llu\n", (unsigned long long)(n + 1));
        for (int j = 0; j < comp_num; j++) {
            const uint64_t cnt = idx[j][n + 1] - idx[j][n];
            for (uint64_t t = 0; t < cnt; t++) {
                if (cursor[j] >= n_ids[j]) break; /* the reference's fread fails silently there and repeats the last id */
                const uint32_t full = (ids[j][cursor[j]++] << cb) | (uint32_t)j;
                uint64_t u = kssd_reverse_id(full, sh->k, sh->subk, sh->drlevel, accepted);
                for (int b = 0; b < TL; b++) { kstring[TL - b - 1] = "ACGT"[u & 3u]; u >>= 2; }
                fprintf(out, "%s\n", kstring);
            }
        }
    }
    for (int j = 0; j < comp_num; j++) { free(idx[j]); free(ids[j]); }
    free(idx); free(ids); free(n_ids); free(cursor);
    return rc;
}
