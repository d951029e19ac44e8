/*
 * kssd_host.c -- .shuf generation / io and FASTA / FASTQ tokenisation into packed batches.
 * See kssd_host.h.  Reference behaviour is cited per function; the code is new.
 */
#define _GNU_SOURCE
#include "kssd_host.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#define CHUNK_BASES 4096u
#define CHUNK_WORDS 256u
#define CHUNK_MASKW 128u
#define SLACK_WORDS 8u

void kssd_host_free(void *p) { free(p); }

const char *kssd_host_strerror(int code)
{
    switch (code) {
    case KSSD_HOST_OK: return "ok";
    case KSSD_HOST_ERR_IO: return "i/o error";
    case KSSD_HOST_ERR_PARAM: return "bad parameter (half-context / half-subcontext / level)";
    case KSSD_HOST_ERR_HEADER: return "can not find seqences head start from '>'";
    case KSSD_HOST_ERR_EMPTY: return "eof or fread error";
    case KSSD_HOST_ERR_NOMEM: return "out of memory";
    case KSSD_HOST_ERR_FORMAT: return "malformed file";
    case KSSD_HOST_ERR_WIDE:
        return "sketches of 36-bit tuples (k - drlevel = 9, 256 components) can be written but not indexed or searched: the reference's own "
               "index builder does not survive them either (tests/golden/make_golden_k12.py)";
    default: return "unknown kssd_host error";
    }
}

/* ---------------------------------------------------------------------------------------------------
 * .shuf
 * ------------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct { uint64_t s[4]; } xoshiro;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t xo_next(xoshiro *g)
{
    uint64_t *s = g->s;
    const uint64_t r = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return r;
}

/* unbiased draw from [0, n) */
static inline uint32_t xo_below(xoshiro *g, uint32_t n)
{
    uint64_t m = (uint64_t)(uint32_t)(xo_next(g) >> 32) * n;
    uint32_t l = (uint32_t)m;
    if (l < n) {
        uint32_t t = (0u - n) % n;
        while (l < t) {
            m = (uint64_t)(uint32_t)(xo_next(g) >> 32) * n;
            l = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

int kssd_shuf_generate(kssd_shuf *s, int k, int subk, int drlevel, uint64_t seed)
{
    if (!s) return KSSD_HOST_ERR_PARAM;
    memset(s, 0, sizeof *s);
    /* same refusals as write_dim_shuffle_file (command_shuffle.c:163-168) */
    if (k < subk || subk >= 8 || subk < 1 || drlevel < 0 || drlevel > subk) return KSSD_HOST_ERR_PARAM;
    if (seed == 0) seed = (uint64_t)time(NULL); /* command_shuffle.c:180 */
    xoshiro g;
    uint64_t sm = seed;
    for (int i = 0; i < 4; i++) g.s[i] = splitmix64(&sm);
    const uint32_t n = 1u << (4 * subk);
    int32_t *t = malloc((size_t)n * sizeof(int32_t));
    if (!t) return KSSD_HOST_ERR_NOMEM;
    for (uint32_t i = 0; i < n; i++) t[i] = (int32_t)i;
    for (uint32_t i = n - 1; i > 0; i--) { /* Fisher-Yates, command_shuffle.c:131-144 */
        uint32_t j = xo_below(&g, i + 1);
        int32_t tmp = t[i];
        t[i] = t[j];
        t[j] = tmp;
    }
    s->id = (int32_t)(xo_next(&g) >> 33); /* a positive int, like rand() (command_shuffle.c:181) */
    s->k = k;
    s->subk = subk;
    s->drlevel = drlevel;
    s->table = t;
    return KSSD_HOST_OK;
}

int kssd_shuf_write(const kssd_shuf *s, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return KSSD_HOST_ERR_IO;
    int32_t hdr[4] = {s->id, s->k, s->subk, s->drlevel}; /* command_shuffle.c:184-185 */
    size_t n = (size_t)1 << (4 * s->subk);
    int ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(s->table, sizeof(int32_t), n, f) == n;
    if (fclose(f) != 0) ok = 0;
    return ok ? KSSD_HOST_OK : KSSD_HOST_ERR_IO;
}

int kssd_shuf_read(kssd_shuf *s, const char *path)
{
    memset(s, 0, sizeof *s);
    size_t L = strlen(path);
    if (L < 5 || strcmp(path + L - 5, ".shuf") != 0) return KSSD_HOST_ERR_FORMAT; /* command_shuffle.c:194-196 */
    FILE *f = fopen(path, "rb");
    if (!f) return KSSD_HOST_ERR_IO;
    int32_t hdr[4];
    if (fread(hdr, sizeof hdr, 1, f) != 1 || hdr[2] < 1 || hdr[2] >= 8) {
        fclose(f);
        return KSSD_HOST_ERR_FORMAT;
    }
    size_t n = (size_t)1 << (4 * hdr[2]);
    int32_t *t = malloc(n * sizeof(int32_t));
    if (!t) {
        fclose(f);
        return KSSD_HOST_ERR_NOMEM;
    }
    if (fread(t, sizeof(int32_t), n, f) != n) {
        free(t);
        fclose(f);
        return KSSD_HOST_ERR_FORMAT;
    }
    fclose(f);
    s->id = hdr[0]; s->k = hdr[1]; s->subk = hdr[2]; s->drlevel = hdr[3];
    s->table = t;
    return KSSD_HOST_OK;
}

/* The accepted sub-contexts of a .shuf: accepted[r] = the sub-context the permutation sends to rank r, r < dim_end =
 * max(16^(subk - drlevel), 4096) (iseq2comem.c:74-76, 247-249).  They are all of the file the sketch path ever looks at --
 * 16 KiB of a 64 MiB (L3K10) or 1 GiB (-s 7) table -- so they are kept beside it as "<file>.core":
 *   u32 magic, u32 version, i32 id, k, subk, drlevel, u64 size of the .shuf, i64 its mtime (ns), u32 n, u32 pad, u32 accepted[n]
 * A core whose header does not describe the .shuf as it is now is ignored and rewritten; a directory that cannot be
 * written to just means the table is scanned every time (mapped, scanned by the OpenMP team, never copied). */
#define SHUF_CORE_MAGIC 0x6b737363u /* "kssc" */
typedef struct {
    uint32_t magic, version;
    int32_t id, k, subk, drlevel;
    uint64_t shuf_size;
    int64_t shuf_mtime_ns;
    uint32_t n, sum; /* sum: FNV-1a over accepted[] */
} shuf_core_hdr;

static uint32_t core_sum(const uint32_t *acc, uint32_t n)
{
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; i++)
        for (int b = 0; b < 4; b++) h = (h ^ ((acc[i] >> (8 * b)) & 0xFFu)) * 16777619u;
    return h;
}

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* a cached core is believed only if it can be a core of THIS table: every value a sub-context, no value twice, and a sample
 * of 64 ranks read back from the .shuf itself (a table rewritten inside one mtime tick at the same size is caught by the
 * sample with all but negligible probability: two permutations agree on a given rank with probability dim_end / 16^subk) */
static int core_plausible(int shuf_fd, const uint32_t *acc, uint32_t n, size_t n_tab)
{
    uint32_t *sorted = malloc((size_t)n * 4);
    if (!sorted) return 0;
    memcpy(sorted, acc, (size_t)n * 4);
    qsort(sorted, n, 4, cmp_u32);
    int ok = sorted[n - 1] < n_tab;
    for (uint32_t i = 1; ok && i < n; i++) ok = sorted[i] != sorted[i - 1];
    free(sorted);
    for (uint32_t j = 0; ok && j < 64; j++) {
        const uint32_t r = (uint32_t)(((uint64_t)j * n) / 64);
        int32_t v;
        ok = pread(shuf_fd, &v, 4, (off_t)(16 + (uint64_t)acc[r] * 4)) == 4 && v == (int32_t)r;
    }
    return ok;
}

int kssd_shuf_read_core(const char *path, kssd_shuf *hdr_out, uint32_t **accepted_out, uint32_t *n_out, int *from_cache)
{
    memset(hdr_out, 0, sizeof *hdr_out);
    *accepted_out = NULL;
    *n_out = 0;
    if (from_cache) *from_cache = 0;
    size_t L = strlen(path);
    if (L < 5 || strcmp(path + L - 5, ".shuf") != 0) return KSSD_HOST_ERR_FORMAT; /* command_shuffle.c:194-196 */
    int fd = open(path, O_RDONLY);
    if (fd < 0) return KSSD_HOST_ERR_IO;
    struct stat st;
    int32_t hdr[4];
    if (fstat(fd, &st) != 0 || pread(fd, hdr, sizeof hdr, 0) != (ssize_t)sizeof hdr || hdr[2] < 1 || hdr[2] >= 8) {
        close(fd);
        return KSSD_HOST_ERR_FORMAT;
    }
    const size_t n_tab = (size_t)1 << (4 * hdr[2]);
    if ((uint64_t)st.st_size < 16 + n_tab * 4) { close(fd); return KSSD_HOST_ERR_FORMAT; }
    hdr_out->id = hdr[0]; hdr_out->k = hdr[1]; hdr_out->subk = hdr[2]; hdr_out->drlevel = hdr[3];
    if (hdr[3] < 0 || hdr[3] > hdr[2]) { close(fd); return KSSD_HOST_ERR_FORMAT; }
    uint64_t sub = 1ull << (4 * (hdr[2] - hdr[3]));
    const uint32_t dim_end = (uint32_t)(sub > 4096 ? sub : 4096); /* MIN_SUBCTX_DIM_SMP_SZ, command_shuffle.h:29 */
    if (dim_end > n_tab) { close(fd); return KSSD_HOST_ERR_FORMAT; }
    uint32_t *acc = malloc((size_t)dim_end * 4);
    if (!acc) { close(fd); return KSSD_HOST_ERR_NOMEM; }
    const int64_t mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    char cpath[KSSD_PATHLEN + 16];
    snprintf(cpath, sizeof cpath, "%s.core", path);
    if (!getenv("KSSD_NO_SHUF_CORE")) {
        int cfd = open(cpath, O_RDONLY);
        if (cfd >= 0) {
            shuf_core_hdr ch;
            if (read(cfd, &ch, sizeof ch) == (ssize_t)sizeof ch && ch.magic == SHUF_CORE_MAGIC && ch.version == 2 && ch.id == hdr[0] && ch.k == hdr[1] &&
                ch.subk == hdr[2] && ch.drlevel == hdr[3] && ch.shuf_size == (uint64_t)st.st_size && ch.shuf_mtime_ns == mtime_ns &&
                ch.n == dim_end && read(cfd, acc, (size_t)dim_end * 4) == (ssize_t)((size_t)dim_end * 4) && ch.sum == core_sum(acc, dim_end) &&
                core_plausible(fd, acc, dim_end, n_tab)) {
                close(cfd);
                close(fd);
                *accepted_out = acc;
                *n_out = dim_end;
                if (from_cache) *from_cache = 1;
                return KSSD_HOST_OK;
            }
            close(cfd); /* anything off: the table itself is scanned (and the core rewritten) */
        }
    }
    const int32_t *tab = mmap(NULL, 16 + n_tab * 4, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (tab == MAP_FAILED) { free(acc); return KSSD_HOST_ERR_IO; }
    tab += 4;
    memset(acc, 0xFF, (size_t)dim_end * 4);
    uint64_t found = 0;
    int dup = 0;
#pragma omp parallel for schedule(static) reduction(+ : found) reduction(| : dup)
    for (size_t x = 0; x < n_tab; x++) {
        const int32_t r = tab[x];
        if (r >= 0 && (uint32_t)r < dim_end) {
            uint32_t expect = 0xFFFFFFFFu; /* (two sub-contexts with one rank: not a permutation) */
            if (!__atomic_compare_exchange_n(&acc[r], &expect, (uint32_t)x, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) dup = 1;
            found++;
        }
    }
    munmap((void *)(tab - 4), 16 + n_tab * 4);
    if (dup || found != dim_end) { free(acc); return KSSD_HOST_ERR_FORMAT; }
    if (!getenv("KSSD_NO_SHUF_CORE")) { /* leave the core for the next run: temporary name, then rename (a reader never sees half a file) */
        char tmp[KSSD_PATHLEN + 48];
        snprintf(tmp, sizeof tmp, "%s.tmp%ld", cpath, (long)getpid());
        int wfd = open(tmp, O_WRONLY | O_CREAT | O_EXCL, 0644);
        if (wfd >= 0) {
            shuf_core_hdr ch = {SHUF_CORE_MAGIC, 2, hdr[0], hdr[1], hdr[2], hdr[3], (uint64_t)st.st_size, mtime_ns, dim_end, core_sum(acc, dim_end)};
            const int ok = write(wfd, &ch, sizeof ch) == (ssize_t)sizeof ch && write(wfd, acc, (size_t)dim_end * 4) == (ssize_t)((size_t)dim_end * 4);
            close(wfd);
            if (!ok || rename(tmp, cpath) != 0) unlink(tmp);
        }
    }
    *accepted_out = acc;
    *n_out = dim_end;
    return KSSD_HOST_OK;
}

int kssd_shard_plan(const int *devices, int n_devices, uint32_t n_files, uint32_t *first)
{
    if (!devices || !first || n_devices < 1) return KSSD_HOST_ERR_PARAM;
    for (int d = 0; d < n_devices; d++) {
        if (devices[d] < 0) return KSSD_HOST_ERR_PARAM;
        for (int e = 0; e < d; e++)
            if (devices[e] == devices[d]) return KSSD_HOST_ERR_PARAM;
    }
    const uint64_t per = ((uint64_t)n_files + (uint64_t)n_devices - 1) / (uint64_t)n_devices;
    for (int d = 0; d <= n_devices; d++) {
        const uint64_t at = per * (uint64_t)d;
        first[d] = (uint32_t)(at < n_files ? at : n_files);
    }
    return KSSD_HOST_OK;
}

void kssd_shuf_release(kssd_shuf *s)
{
    if (s) {
        free(s->table);
        s->table = NULL;
    }
}

/* ---------------------------------------------------------------------------------------------------
 * packed batches
 * ------------------------------------------------------------------------------------------------- */
struct kssd_batch {
    uint32_t *packed, *mask;
    uint64_t n_chunks, cap_chunks;
    uint64_t *chunk_off; /* n_genomes + 1 */
    uint64_t *n_pos;     /* per genome */
    uint32_t n_genomes, cap_genomes;
    void *(*alloc)(size_t);  /* where packed / mask live: NULL = malloc, else e.g. page-locked memory of the GPU runtime */
    void (*release)(void *);
};

kssd_batch *kssd_batch_create(void) { return kssd_batch_create_ex(NULL, NULL); }

kssd_batch *kssd_batch_create_ex(void *(*alloc)(size_t), void (*release)(void *))
{
    kssd_batch *b = calloc(1, sizeof *b);
    if (!b) return NULL;
    if (alloc && release) {
        b->alloc = alloc;
        b->release = release;
    }
    b->cap_genomes = 16;
    b->chunk_off = calloc(b->cap_genomes + 1, sizeof(uint64_t));
    b->n_pos = calloc(b->cap_genomes, sizeof(uint64_t));
    if (!b->chunk_off || !b->n_pos) {
        kssd_batch_destroy(b);
        return NULL;
    }
    return b;
}

void kssd_batch_destroy(kssd_batch *b)
{
    if (!b) return;
    if (b->release) {
        if (b->packed) b->release(b->packed);
        if (b->mask) b->release(b->mask);
    } else {
        free(b->packed);
        free(b->mask);
    }
    free(b->chunk_off);
    free(b->n_pos);
    free(b);
}

void kssd_batch_clear(kssd_batch *b)
{
    /* everything past n_chunks is zero already (growth zeroes new memory, every writer stays inside its genome) */
    uint64_t used = b->n_chunks + 1 < b->cap_chunks ? b->n_chunks + 1 : b->cap_chunks;
    if (b->packed) memset(b->packed, 0, (size_t)used * CHUNK_WORDS * 4);
    if (b->mask) memset(b->mask, 0, (size_t)used * CHUNK_MASKW * 4);
    b->n_chunks = 0;
    b->n_genomes = 0;
    b->chunk_off[0] = 0;
}

static int batch_reserve_chunks(kssd_batch *b, uint64_t need)
{
    if (need <= b->cap_chunks) return KSSD_HOST_OK;
    uint64_t nc = b->cap_chunks ? b->cap_chunks : 64;
    while (nc < need) nc += nc / 2 + 64;
    size_t pw_old = b->packed ? (size_t)b->cap_chunks * CHUNK_WORDS + SLACK_WORDS : 0;
    size_t mw_old = b->mask ? (size_t)b->cap_chunks * CHUNK_MASKW + SLACK_WORDS : 0;
    size_t pw = (size_t)nc * CHUNK_WORDS + SLACK_WORDS, mw = (size_t)nc * CHUNK_MASKW + SLACK_WORDS;
    uint32_t *p, *m;
    if (b->alloc) { /* no realloc for foreign memory: new block, copy what is in use, release the old one */
        p = b->alloc(pw * 4);
        if (!p) return KSSD_HOST_ERR_NOMEM;
        if (b->packed) {
            memcpy(p, b->packed, pw_old * 4);
            b->release(b->packed);
        }
        b->packed = p;
        memset(p + pw_old, 0, (pw - pw_old) * 4);
        m = b->alloc(mw * 4);
        if (!m) return KSSD_HOST_ERR_NOMEM;
        if (b->mask) {
            memcpy(m, b->mask, mw_old * 4);
            b->release(b->mask);
        }
        b->mask = m;
        memset(m + mw_old, 0, (mw - mw_old) * 4);
        b->cap_chunks = nc;
        return KSSD_HOST_OK;
    }
    p = realloc(b->packed, pw * 4);
    if (!p) return KSSD_HOST_ERR_NOMEM;
    b->packed = p;
    memset(p + pw_old, 0, (pw - pw_old) * 4);
    m = realloc(b->mask, mw * 4);
    if (!m) return KSSD_HOST_ERR_NOMEM;
    b->mask = m;
    memset(m + mw_old, 0, (mw - mw_old) * 4);
    b->cap_chunks = nc;
    return KSSD_HOST_OK;
}

static int batch_begin(kssd_batch *b)
{
    if (b->n_genomes == b->cap_genomes) {
        uint32_t nc = b->cap_genomes * 2;
        uint64_t *o = realloc(b->chunk_off, ((size_t)nc + 1) * sizeof(uint64_t));
        if (!o) return KSSD_HOST_ERR_NOMEM;
        b->chunk_off = o;
        uint64_t *np = realloc(b->n_pos, (size_t)nc * sizeof(uint64_t));
        if (!np) return KSSD_HOST_ERR_NOMEM;
        b->n_pos = np;
        b->cap_genomes = nc;
    }
    return KSSD_HOST_OK;
}

/* writer state of the genome being appended */
typedef struct {
    kssd_batch *b;
    uint64_t base_pos; /* first position of this genome in the batch */
    uint64_t p;        /* positions written so far */
    int pending_break; /* a run-breaking byte was seen since the last base */
    uint32_t pw, mw;   /* the packed / mask word being put together (stored when full, gw_put) */
    uint64_t limit;    /* fixed mode (kssd_batch_fill_text): the genome's reserved positions (0: an empty reservation) */
    int fixed;         /* 1 = fixed mode, 0 = growing mode */
    uint32_t genome;   /* fixed mode: which genome */
} gwriter;

static inline int gw_room(gwriter *w)
{
    /* make sure position base_pos + p is inside the buffers */
    if (w->fixed) return w->p < w->limit ? KSSD_HOST_OK : KSSD_HOST_ERR_PARAM; /* reserved and zeroed up front */
    uint64_t chunk = (w->base_pos + w->p) / CHUNK_BASES;
    if (chunk >= w->b->cap_chunks) return batch_reserve_chunks(w->b, chunk + 1);
    return KSSD_HOST_OK;
}

/* one position: the words are put together in registers and stored whole, 16 bases / 32 mask bits at a time (the batch
 * may live in page-locked memory, where a read-modify-write per base costs a bus round trip); a genome starts on a
 * chunk boundary and has one writer, so no word is shared */
static inline int gw_put(gwriter *w, uint32_t code, uint32_t valid)
{
    if ((w->p & (CHUNK_BASES - 1)) == 0) { /* entering a chunk: room for it */
        int rc = gw_room(w);
        if (rc) return rc;
    }
    const unsigned q = (unsigned)(w->p & 31);
    w->pw |= code << (30 - 2 * (q & 15));
    w->mw |= valid << q;
    w->p++;
    if ((q & 15) == 15) {
        const uint64_t g = w->base_pos + w->p - 1;
        w->b->packed[g >> 4] = w->pw;
        w->pw = 0;
        if (q == 31) {
            w->b->mask[g >> 5] = w->mw;
            w->mw = 0;
        }
    }
    return KSSD_HOST_OK;
}

/* store what the last, incomplete words hold */
static inline void gw_flush(gwriter *w)
{
    if (w->p & 15) w->b->packed[(w->base_pos + w->p) >> 4] = w->pw;
    if (w->p & 31) w->b->mask[(w->base_pos + w->p) >> 5] = w->mw;
    w->pw = w->mw = 0;
}

static inline int gw_base(gwriter *w, unsigned code)
{
    int rc;
    if (w->pending_break) {
        /* one invalid position stands for any stretch of bytes that reset the run counter */
        if (w->p && (rc = gw_put(w, 0, 0)) != 0) return rc;
        w->pending_break = 0;
    }
    return gw_put(w, code, 1);
}

static void gw_start(gwriter *w, kssd_batch *b)
{
    w->b = b;
    w->base_pos = b->n_chunks * CHUNK_BASES;
    w->p = 0;
    w->pending_break = 0;
    w->pw = w->mw = 0;
    w->limit = 0;
    w->fixed = 0;
    w->genome = 0;
}

static void gw_start_fixed(gwriter *w, kssd_batch *b, uint32_t g)
{
    w->b = b;
    w->base_pos = b->chunk_off[g] * CHUNK_BASES;
    w->p = 0;
    w->pending_break = 0;
    w->pw = w->mw = 0;
    w->limit = (b->chunk_off[g + 1] - b->chunk_off[g]) * CHUNK_BASES; /* 0 for an empty reservation: nothing may be written */
    w->fixed = 1;
    w->genome = g;
}

static void gw_finish(gwriter *w)
{
    kssd_batch *b = w->b;
    gw_flush(w);
    if (w->fixed) { /* the layout was fixed by kssd_batch_reserve */
        b->n_pos[w->genome] = w->p;
        return;
    }
    uint64_t chunks = (w->p + CHUNK_BASES - 1) / CHUNK_BASES;
    b->n_pos[b->n_genomes] = w->p;
    b->n_chunks += chunks;
    b->n_genomes++;
    b->chunk_off[b->n_genomes] = b->n_chunks;
}

/* roll back a genome whose input turned out to be malformed */
static void gw_abort(gwriter *w)
{
    kssd_batch *b = w->b;
    if (w->fixed) {
        const uint64_t c0 = b->chunk_off[w->genome], c1 = b->chunk_off[w->genome + 1];
        memset(b->packed + c0 * CHUNK_WORDS, 0, (size_t)(c1 - c0) * CHUNK_WORDS * 4);
        memset(b->mask + c0 * CHUNK_MASKW, 0, (size_t)(c1 - c0) * CHUNK_MASKW * 4);
        b->n_pos[w->genome] = 0;
        return;
    }
    uint64_t c0 = b->n_chunks, c1 = (w->base_pos + w->p + CHUNK_BASES - 1) / CHUNK_BASES;
    if (c1 > b->cap_chunks) c1 = b->cap_chunks;
    if (c1 > c0) {
        memset(b->packed + c0 * CHUNK_WORDS, 0, (size_t)(c1 - c0) * CHUNK_WORDS * 4);
        memset(b->mask + c0 * CHUNK_MASKW, 0, (size_t)(c1 - c0) * CHUNK_MASKW * 4);
    }
}

/* byte classes of the FASTA scanner (iseq2comem.c:213-242, Basemap global_basic.c:64-71) */
enum { C_A = 0, C_C = 1, C_G = 2, C_T = 3, C_SKIP = 4, C_HEADER = 5, C_BREAK = 6 };
/* constant table (every byte not named is C_BREAK = 6): no lazy initialisation, the tokenisers run on many threads */
#define B6 C_BREAK
#define B6x8 B6, B6, B6, B6, B6, B6, B6, B6
#define B6x16 B6x8, B6x8
static const unsigned char fa_class[256] = {
    /* 0x00 */ B6x8, B6, B6, C_SKIP /* \n */, B6, B6, C_SKIP /* \r */, B6, B6,
    /* 0x10 */ B6x16,
    /* 0x20 */ B6x16,
    /* 0x30 */ B6x8, B6, B6, B6, B6, B6, B6, C_HEADER /* > */, B6,
    /* 0x40 */ B6, C_A, B6, C_C, B6, B6, B6, C_G, B6x8,
    /* 0x50 */ B6, B6, B6, B6, C_T, B6, B6, B6, B6x8,
    /* 0x60 */ B6, C_A, B6, C_C, B6, B6, B6, C_G, B6x8,
    /* 0x70 */ B6, B6, B6, B6, C_T, B6, B6, B6, B6x8,
    /* 0x80 */ B6x16, B6x16, B6x16, B6x16, B6x16, B6x16, B6x16, B6x16};
#undef B6x16
#undef B6x8
#undef B6

static int tok_fasta(gwriter *w, const unsigned char *text, size_t n)
{
    int rc;
    if (n == 0) return KSSD_HOST_ERR_EMPTY;
    for (size_t i = 0; i < n; i++) {
        unsigned cls = fa_class[text[i]];
        if (cls < 4) {
            if ((rc = gw_base(w, cls)) != 0) { gw_abort(w); return rc; }
        } else if (cls == C_SKIP) {
            /* line ends are transparent: k-mers run across them */
        } else if (cls == C_HEADER) {
            const unsigned char *nl = memchr(text + i, '\n', n - i);
            if (!nl) { gw_abort(w); return KSSD_HOST_ERR_HEADER; }
            i = (size_t)(nl - text);
            w->pending_break = 1;
        } else {
            w->pending_break = 1;
        }
    }
    gw_finish(w);
    return KSSD_HOST_OK;
}

int kssd_batch_add_fasta(kssd_batch *b, const unsigned char *text, size_t n)
{
    if (n == 0) return KSSD_HOST_ERR_EMPTY;
    int rc = batch_begin(b);
    if (rc) return rc;
    gwriter w;
    gw_start(&w, b);
    return tok_fasta(&w, text, n);
}

/* The same scanner for dist --byread (reads2mco, iseq2comem.c:110-156): one genome per file, and for every '>' the
 * position (inside the genome) the next base will be written at.  A k-mer belongs to read r (0 = before the first
 * '>') iff starts[r-1] <= its position < starts[r]: every k-mer met after a header lies entirely behind the break
 * position the header leaves, every earlier one entirely before it. */
int kssd_batch_add_fasta_reads(kssd_batch *b, const unsigned char *text, size_t n, uint64_t **read_start, uint64_t *n_reads)
{
    if (!read_start || !n_reads) return KSSD_HOST_ERR_PARAM;
    *read_start = NULL;
    *n_reads = 0;
    if (n == 0) return KSSD_HOST_ERR_EMPTY;
    int rc = batch_begin(b);
    if (rc) return rc;
    size_t cap = 1024, nr = 0;
    uint64_t *st = malloc(cap * sizeof *st);
    if (!st) return KSSD_HOST_ERR_NOMEM;
    gwriter w;
    gw_start(&w, b);
    for (size_t i = 0; i < n; i++) {
        unsigned cls = fa_class[text[i]];
        if (cls < 4) {
            if ((rc = gw_base(&w, cls)) != 0) { gw_abort(&w); free(st); return rc; }
        } else if (cls == C_SKIP) {
        } else if (cls == C_HEADER) {
            const unsigned char *nl = memchr(text + i, '\n', n - i);
            if (!nl) { gw_abort(&w); free(st); return KSSD_HOST_ERR_HEADER; }
            i = (size_t)(nl - text);
            w.pending_break = 1;
            if (nr == cap) {
                cap *= 2;
                uint64_t *q = realloc(st, cap * sizeof *st);
                if (!q) { gw_abort(&w); free(st); return KSSD_HOST_ERR_NOMEM; }
                st = q;
            }
            st[nr++] = w.p + (w.p ? 1 : 0); /* gw_base puts one invalid position in front of the next base */
        } else {
            w.pending_break = 1;
        }
    }
    gw_finish(&w);
    *read_start = st;
    *n_reads = nr;
    return KSSD_HOST_OK;
}

/* fgets() on a memory stream, end-of-file indicator included */
typedef struct {
    const unsigned char *p;
    size_t n, pos;
    int eof;
} mstream;

static char *ms_gets(char *buf, int size, mstream *s)
{
    int i = 0;
    while (i < size - 1) {
        if (s->pos >= s->n) { s->eof = 1; break; }
        char ch = (char)s->p[s->pos++];
        buf[i++] = ch;
        if (ch == '\n') break;
    }
    if (i == 0) return NULL;
    buf[i] = 0;
    return buf;
}

#define FQ_LINE 20000 /* LEN, iseq2comem.c:274 */

static int tok_fastq(gwriter *wp, const unsigned char *text, size_t n, int Q, uint64_t *n_lines);

int kssd_batch_add_fastq(kssd_batch *b, const unsigned char *text, size_t n, int Q, uint64_t *n_lines)
{
    int rc = batch_begin(b);
    if (rc) return rc;
    gwriter w;
    gw_start(&w, b);
    return tok_fastq(&w, text, n, Q, n_lines);
}

static int tok_fastq(gwriter *wp, const unsigned char *text, size_t n, int Q, uint64_t *n_lines)
{
    int rc;
    char *seq = calloc(1, FQ_LINE + 10), *qual = calloc(1, FQ_LINE + 10);
    if (!seq || !qual) { free(seq); free(qual); gw_abort(wp); return KSSD_HOST_ERR_NOMEM; }
#define w (*wp)
    mstream ms = {text, n, 0, 0};
    uint64_t lines = 0;
    /* record = name, bases, '+', qualities; the second and fourth line are what the scanner keeps */
    ms_gets(seq, FQ_LINE, &ms); ms_gets(seq, FQ_LINE, &ms);
    ms_gets(qual, FQ_LINE, &ms); ms_gets(qual, FQ_LINE, &ms);
    int sl = (int)strlen(seq);
    for (int pos = 0; pos < sl; pos++) {
        if (seq[pos] == '\n') {
            ms_gets(seq, FQ_LINE, &ms); ms_gets(seq, FQ_LINE, &ms);
            ms_gets(qual, FQ_LINE, &ms); ms_gets(qual, FQ_LINE, &ms);
            sl = (int)strlen(seq);
            lines += 4;
            if (ms.eof) break; /* a final record without trailing newline is not scanned (iseq2comem.c:302-307) */
            w.pending_break = 1;
            pos = -1;
            continue;
        }
        unsigned cls = fa_class[(unsigned char)seq[pos]];
        if (cls < 4 && qual[pos] >= Q) {
            if ((rc = gw_base(&w, cls)) != 0) { gw_abort(&w); free(seq); free(qual); return rc; }
        } else {
            w.pending_break = 1;
        }
    }
    free(seq);
    free(qual);
    gw_finish(&w);
    if (n_lines) *n_lines = lines;
    return KSSD_HOST_OK;
#undef w
}

#define KOC_LINE 4096 /* FQ_LEN of the abundance scanner, iseq2comem.c:553 */

/* the reads as dist -A scans them (mt_shortreads2koc, iseq2comem.c:554-615): records of four fgets() lines of at
 * most KOC_LINE-1 bytes, the second one scanned up to its newline, no quality filter; a read restarts the k-mer and
 * so does every byte that is not ACGT/acgt.  (A line without newline inside KOC_LINE-1 bytes makes the reference
 * run off its buffer; here the scan stops at the end of what fgets() returned.) */
static int tok_reads(gwriter *wp, const unsigned char *text, size_t n, uint64_t *n_reads);

int kssd_batch_add_reads(kssd_batch *b, const unsigned char *text, size_t n, uint64_t *n_reads)
{
    int rc = batch_begin(b);
    if (rc) return rc;
    gwriter w;
    gw_start(&w, b);
    return tok_reads(&w, text, n, n_reads);
}

static int tok_reads(gwriter *wp, const unsigned char *text, size_t n, uint64_t *n_reads)
{
    int rc;
    char *seq = calloc(1, KOC_LINE + 10), *tmp = calloc(1, KOC_LINE + 10);
    if (!seq || !tmp) { free(seq); free(tmp); gw_abort(wp); return KSSD_HOST_ERR_NOMEM; }
#define w (*wp)
    mstream ms = {text, n, 0, 0};
    uint64_t reads = 0;
    while (ms_gets(tmp, KOC_LINE, &ms) && ms_gets(seq, KOC_LINE, &ms) && ms_gets(tmp, KOC_LINE, &ms) && ms_gets(tmp, KOC_LINE, &ms)) {
        w.pending_break = 1;
        for (int pos = 0; seq[pos] && seq[pos] != '\n'; pos++) {
            unsigned cls = fa_class[(unsigned char)seq[pos]];
            if (cls < 4) {
                if ((rc = gw_base(&w, cls)) != 0) { gw_abort(&w); free(seq); free(tmp); return rc; }
            } else {
                w.pending_break = 1;
            }
        }
        reads++;
    }
    free(seq);
    free(tmp);
    gw_finish(&w);
    if (n_reads) *n_reads = reads;
    return KSSD_HOST_OK;
#undef w
}

/* ---- parallel fill: the layout first (serial), then one tokeniser per genome on as many threads as the caller has ---- */
int kssd_batch_reserve(kssd_batch *b, uint32_t n, const uint64_t *max_pos, uint32_t *first)
{
    if (!b || (!max_pos && n)) return KSSD_HOST_ERR_PARAM;
    if (first) *first = b->n_genomes;
    uint64_t need = b->n_chunks;
    for (uint32_t i = 0; i < n; i++) need += (max_pos[i] + CHUNK_BASES - 1) / CHUNK_BASES;
    int rc = batch_reserve_chunks(b, need + 1);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
        if ((rc = batch_begin(b)) != 0) return rc;
        b->n_pos[b->n_genomes] = 0;
        b->n_chunks += (max_pos[i] + CHUNK_BASES - 1) / CHUNK_BASES;
        b->n_genomes++;
        b->chunk_off[b->n_genomes] = b->n_chunks;
    }
    return KSSD_HOST_OK;
}

int kssd_batch_fill_text(kssd_batch *b, uint32_t genome, int kind, const unsigned char *text, size_t n, int Q, uint64_t *n_lines)
{
    if (!b || genome >= b->n_genomes) return KSSD_HOST_ERR_PARAM;
    gwriter w;
    gw_start_fixed(&w, b, genome);
    if (kind == 2) return tok_reads(&w, text, n, n_lines);
    if (kind == 1) return tok_fastq(&w, text, n, Q, n_lines);
    return tok_fasta(&w, text, n);
}

int kssd_slurp(const char *path, unsigned char **buf, size_t *len)
{
    size_t cap = 0;
    *buf = NULL;
    const int rc = kssd_slurp_reuse(path, buf, &cap, len);
    if (rc != KSSD_HOST_OK) {
        free(*buf);
        *buf = NULL;
    } else if (!*buf) {
        *buf = malloc(1); /* (an empty file: the callers free what they get) */
    }
    return rc;
}

/* an open file's bytes from where it stands to its end (st_size is a hint), in a buffer the THREAD keeps from file to file: a
 * compressed genome is 1 - 2 MB, which malloc() serves by mmap -- a thousand files are a thousand mappings, half a million page
 * faults and as many trips through the process's one address-space lock, taken by sixteen threads at once.  `slot` 0 / 1: the two
 * files of a pair.  The buffers live as long as the thread. */
static __thread unsigned char *t_zbuf[2];
static __thread size_t t_zcap[2];
/* (given back when the thread ends: a key whose destructor frees them, armed by the thread's first read) */
static pthread_key_t t_zkey;
static pthread_once_t t_zkey_once = PTHREAD_ONCE_INIT;
static void t_zfree(void *unused)
{
    (void)unused;
    for (int i = 0; i < 2; i++) { free(t_zbuf[i]); t_zbuf[i] = NULL; t_zcap[i] = 0; }
}
static void t_zkey_make(void) { pthread_key_create(&t_zkey, t_zfree); }
static int read_fd_whole(int fd, int slot, unsigned char **out, size_t *n_out)
{
    struct stat zst;
    if (fstat(fd, &zst) != 0) return KSSD_HOST_ERR_IO;
    size_t zn = 0;
    for (;;) {
        const size_t want = zn + ((size_t)zst.st_size > zn ? (size_t)zst.st_size - zn : 0) + 4096;
        if (t_zcap[slot] < want) {
            pthread_once(&t_zkey_once, t_zkey_make);
            pthread_setspecific(t_zkey, (void *)1); /* (any non-NULL value: the destructor runs for threads that hold buffers) */
            const size_t nc = want + want / 4;
            unsigned char *q = realloc(t_zbuf[slot], nc);
            if (!q) return KSSD_HOST_ERR_NOMEM;
            t_zbuf[slot] = q;
            t_zcap[slot] = nc;
        }
        const ssize_t r = read(fd, t_zbuf[slot] + zn, t_zcap[slot] - zn);
        if (r < 0) return KSSD_HOST_ERR_IO;
        if (r == 0) break;
        zn += (size_t)r;
        zst.st_size = 0; /* (whatever comes beyond the hint: 4 KiB at a time, the buffer grows by a quarter) */
    }
    *out = t_zbuf[slot];
    *n_out = zn;
    return KSSD_HOST_OK;
}

/* a gzip'ed file through zlib's gzread (takes the descriptor over): the decoder of rounds 1 - 4, KSSD_ZLIB_GUNZIP=1, and the second
 * opinion when host/kssd_inflate.c refuses a stream */
static int slurp_gzread(int fd, unsigned char **buf, size_t *cap, size_t *len)
{
    gzFile g = gzdopen(fd, "rb");
    if (!g) { close(fd); return KSSD_HOST_ERR_IO; }
    gzbuffer(g, 1 << 20);
    size_t n = 0;
    *len = 0;
    for (;;) {
        if (*cap - n < (1u << 20)) {
            size_t nc = *cap + *cap / 2 + (1u << 22);
            unsigned char *q = realloc(*buf, nc);
            if (!q) { gzclose(g); return KSSD_HOST_ERR_NOMEM; }
            *buf = q;
            *cap = nc;
        }
        size_t room = *cap - n;
        int r = gzread(g, *buf + n, (unsigned)(room > (1u << 30) ? (1u << 30) : room));
        if (r < 0) { gzclose(g); return KSSD_HOST_ERR_IO; }
        if (r == 0) break;
        n += (size_t)r;
    }
    /* (gzread returns what it has and reports a damaged or truncated stream only through gzerror / gzclose) */
    int err = Z_OK;
    (void)gzerror(g, &err);
    const int eof_clean = err == Z_OK || err == Z_STREAM_END;
    if (gzclose(g) != Z_OK || !eof_clean) return KSSD_HOST_ERR_IO;
    *len = n;
    return KSSD_HOST_OK;
}

/* the same into a buffer the caller keeps from file to file (grown when needed): no allocation, no page faults of fresh
 * memory per file.  Plain files are read straight with read(2); gzip'ed ones (magic 1f 8b) go through zlib. */
int kssd_slurp_reuse(const char *path, unsigned char **buf, size_t *cap, size_t *len)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return KSSD_HOST_ERR_IO;
    unsigned char magic[2];
    ssize_t got = pread(fd, magic, 2, 0);
    *len = 0;
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b && !getenv("KSSD_ZLIB_GUNZIP")) {
        /* the compressed bytes whole, then host/kssd_inflate.c (KSSD_ZLIB_GUNZIP=1: zlib's gzread below, the decoder of rounds 1 - 4) */
        unsigned char *z = NULL;
        size_t zn = 0;
        int rc = read_fd_whole(fd, 0, &z, &zn);
        close(fd);
        if (rc == KSSD_HOST_OK) rc = kssd_gunzip_mem(z, zn, buf, cap, len);
        if (rc == KSSD_HOST_ERR_IO && (fd = open(path, O_RDONLY)) >= 0) rc = slurp_gzread(fd, buf, cap, len); /* zlib decides */
        return rc;
    }
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return slurp_gzread(fd, buf, cap, len);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return KSSD_HOST_ERR_IO; }
    size_t want = (size_t)st.st_size, n = 0;
    for (;;) { /* st_size is a hint: read until end of file */
        if (*cap < n + (want > n ? want - n : 0) + 4096) {
            size_t nc = n + (want > n ? want - n : 0) + (1u << 20);
            unsigned char *q = realloc(*buf, nc);
            if (!q) { close(fd); return KSSD_HOST_ERR_NOMEM; }
            *buf = q;
            *cap = nc;
        }
        ssize_t r = read(fd, *buf + n, *cap - n);
        if (r < 0) { close(fd); return KSSD_HOST_ERR_IO; }
        if (r == 0) break;
        n += (size_t)r;
        if (n >= want) want = n + (1u << 20);
    }
    close(fd);
    *len = n;
    return KSSD_HOST_OK;
}

/* two files by one thread: gzip'ed ones are unpacked in step (kssd_gunzip_mem2: the two decoders' lookups overlap); anything else
 * -- a plain file, KSSD_ZLIB_GUNZIP=1, a file that cannot be opened -- goes through kssd_slurp_reuse, one after the other */
void kssd_slurp_reuse2(const char *const path[2], unsigned char **buf[2], size_t *cap[2], size_t *len[2], int rc[2])
{
    unsigned char *z[2] = {NULL, NULL};
    size_t zn[2] = {0, 0};
    int gz = !getenv("KSSD_ZLIB_GUNZIP") && !getenv("KSSD_GZ_ONE_AT_A_TIME"); /* (the second: measurements) */
    for (int f = 0; f < 2 && gz; f++) {
        const int fd = open(path[f], O_RDONLY);
        unsigned char magic[2];
        if (fd < 0 || pread(fd, magic, 2, 0) != 2 || magic[0] != 0x1f || magic[1] != 0x8b || read_fd_whole(fd, f, &z[f], &zn[f]) != KSSD_HOST_OK) gz = 0;
        if (fd >= 0) close(fd);
    }
    if (gz) {
        const unsigned char *const in[2] = {z[0], z[1]};
        kssd_gunzip_mem2(in, zn, buf, cap, len, rc);
        for (int f = 0; f < 2; f++) {
            int fd;
            if (rc[f] == KSSD_HOST_ERR_IO && (fd = open(path[f], O_RDONLY)) >= 0) rc[f] = slurp_gzread(fd, buf[f], cap[f], len[f]); /* zlib decides */
        }
    } else {
        for (int f = 0; f < 2; f++) rc[f] = kssd_slurp_reuse(path[f], buf[f], cap[f], len[f]);
    }
}

/* is the file gzip'ed (magic 1f 8b), and how many bytes does it hold on disk */
int kssd_file_probe(const char *path, int *is_gz, uint64_t *size)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return KSSD_HOST_ERR_IO;
    unsigned char magic[2];
    struct stat st;
    const ssize_t got = pread(fd, magic, 2, 0);
    const int rc = fstat(fd, &st);
    close(fd);
    if (rc != 0) return KSSD_HOST_ERR_IO;
    *is_gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    *size = (uint64_t)st.st_size;
    return KSSD_HOST_OK;
}

/* a plain file's bytes straight into memory of the caller's (e.g. a page-locked buffer): at most cap bytes */
int kssd_read_into(const char *path, unsigned char *dst, size_t cap, size_t *len)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return KSSD_HOST_ERR_IO;
    size_t n = 0;
    while (n < cap) {
        const ssize_t r = read(fd, dst + n, cap - n);
        if (r < 0) { close(fd); return KSSD_HOST_ERR_IO; }
        if (r == 0) break;
        n += (size_t)r;
    }
    close(fd);
    *len = n;
    return KSSD_HOST_OK;
}

int kssd_batch_add_file(kssd_batch *b, const char *path, int is_fastq, int Q, uint64_t *n_lines)
{
    unsigned char *txt = NULL;
    size_t n = 0;
    int rc = kssd_slurp(path, &txt, &n);
    if (rc) return rc;
    rc = is_fastq == 2 ? kssd_batch_add_reads(b, txt, n, n_lines)
         : is_fastq ? kssd_batch_add_fastq(b, txt, n, Q, n_lines) : kssd_batch_add_fasta(b, txt, n);
    free(txt);
    return rc;
}

const uint32_t *kssd_batch_packed(const kssd_batch *b) { return b->packed; }
const uint32_t *kssd_batch_mask(const kssd_batch *b) { return b->mask; }
const uint64_t *kssd_batch_chunk_off(const kssd_batch *b) { return b->chunk_off; }
uint64_t kssd_batch_n_chunks(const kssd_batch *b) { return b->n_chunks; }
uint32_t kssd_batch_n_genomes(const kssd_batch *b) { return b->n_genomes; }
uint64_t kssd_batch_n_positions(const kssd_batch *b, uint32_t g) { return g < b->n_genomes ? b->n_pos[g] : 0; }

int kssd_batch_append(kssd_batch *dst, const kssd_batch *src)
{
    for (uint32_t g = 0; g < src->n_genomes; g++) {
        int rc = batch_begin(dst);
        if (rc) return rc;
        const uint64_t c0 = src->chunk_off[g], nc = src->chunk_off[g + 1] - c0;
        if ((rc = batch_reserve_chunks(dst, dst->n_chunks + nc + 1)) != 0) return rc;
        if (nc) {
            memcpy(dst->packed + dst->n_chunks * CHUNK_WORDS, src->packed + c0 * CHUNK_WORDS, (size_t)nc * CHUNK_WORDS * 4);
            memcpy(dst->mask + dst->n_chunks * CHUNK_MASKW, src->mask + c0 * CHUNK_MASKW, (size_t)nc * CHUNK_MASKW * 4);
        }
        dst->n_pos[dst->n_genomes] = src->n_pos[g];
        dst->n_chunks += nc;
        dst->n_genomes++;
        dst->chunk_off[dst->n_genomes] = dst->n_chunks;
    }
    return KSSD_HOST_OK;
}
