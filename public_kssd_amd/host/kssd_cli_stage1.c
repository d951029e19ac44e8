/* kssd_cli_stage1.c -- stage I of `kssd dist` on the GPU (run_stageI, command_dist.c:258-380): the inputs read (and unpacked) on the
 * host threads into page-locked text, device workers that tokenise and sketch them, the sketches in the reference's file order into
 * cofiles.stat / combco.*; --byread; and --allpairs: the search on the sketches the devices still hold (kssd_gpu_resident_*). */
#include "kssd_cli.h"

/* ---------------------------------------------------------------------------------------------------
 * stage I on the GPU (run_stageI, command_dist.c:258-380)
 * ------------------------------------------------------------------------------------------------- */

/* One unit of stage-I work: a run of consecutive input files of one kind (FASTA or FASTQ), tokenised into one packed
 * batch in page-locked memory; a device worker sketches it and leaves the genomes' ids in the reference's file order. */
typedef struct textbuf { /* the raw bytes of a job's files, every file on a 16-byte boundary, in page-locked memory by the time a device reads them */
    unsigned char *p;
    size_t cap;
    int kind; /* 1: hipHostMalloc; 0: malloc -- what the first buffers of a command are, read while the runtime starts -- ; 2: malloc,
               * registered with the runtime since (by the worker that takes the job: textbuf_lock) */
} textbuf;

static void textbuf_release(textbuf *tx)
{
    if (tx->p) {
        if (tx->kind == 1) {
            kssd_gpu_host_free(tx->p);
        } else {
            if (tx->kind == 2) kssd_gpu_host_unregister(tx->p);
            free(tx->p);
        }
    }
    tx->p = NULL;
    tx->cap = 0;
    tx->kind = 0;
}

/* room for `need` bytes: page-locked when the runtime is up, ordinary memory before -- on the first buffers of a command, whose
 * alternative is to wait 0.1 - 0.2 s for hipInit before the first byte is read.  A buffer that is large enough is kept as it is. */
static void textbuf_fit(textbuf *tx, size_t need, int runtime_ready)
{
    if (tx->p && tx->cap >= need) return;
    textbuf_release(tx);
    tx->cap = need + need / 4 + 64;
    tx->kind = runtime_ready ? 1 : 0;
    tx->p = runtime_ready ? kssd_gpu_host_alloc(tx->cap) : malloc(tx->cap);
    if (!tx->p) die(ENOMEM, "out of %smemory (%zu bytes)", runtime_ready ? "page-locked " : "", tx->cap);
}

/* before a device copies out of it: ordinary memory is registered in place (0.05 s per GB; a copy out of unregistered memory goes
 * through the runtime's staging buffers at 7 GB/s instead of 55, profiles/r05i_*) and stays so for the rest of the command */
static void textbuf_lock(textbuf *tx)
{
    if (tx->p && tx->kind == 0 && kssd_gpu_host_register(tx->p, tx->cap) == KSSD_OK) tx->kind = 2; /* (refused: the copy still works, slowly) */
}

typedef struct job {
    kssd_batch *b;      /* FASTQ with -Q > 0, -A: tokenised on the host */
    textbuf *tx;        /* FASTA, FASTQ with -Q 0: the raw bytes, tokenised on the device (kssd_gpu_sketch_fast[aq]_text) */
    kssd_batch *own_b;  /* a text job the device handed back: the host tokeniser's batch of it */
    int streamed;       /* one long file: the worker streams it into the device's text buffer (1 plain: stream_file_in, 2 gzip'ed: stream_gz_in) */
    int uploaded;
    uint64_t *toff, *tlen, *lines;
    int is_fq, first_file, n_files;
    int q;            /* the queue (device of the list) that takes it: 0 unless the inputs are dealt out to devices (--allpairs) */
    uint64_t *off;    /* n_files + 1 */
    uint32_t *ids;    /* slot order per genome */
    uint16_t *counts; /* -A */
    uint8_t *sub;     /* k - drlevel = 9: the tuples' low four bits (ids = tuple >> 4) */
    struct job *next;
} job;

/* host threads a device worker uses for its own post-processing (file order of the ids): small teams, so that they do not
 * fight the tokenisers' team for the cores (each pthread has its own OpenMP pool, idle pools spin) */
#define WORKER_OMP 4
/* Text buffers a command may fill while the HIP runtime is still starting (KSSD_TEXT_AHEAD; ordinary memory, registered with the
 * runtime by the worker that takes the job).  0: the first byte is read once the runtime is up, into page-locked memory -- the default,
 * because reading ahead did NOT pay on the measurement box (profiles/r05j_e2e_probe.txt, 1 024 x 5 Mb files in tmpfs, wall time of
 * the command): 0 / 8 / 16 / 32 / 64 buffers ahead 0.60 / 0.61 / 0.78 / 1.18 / 1.25 s -- hipInit and the context creation run slower
 * beside sixteen reading threads (0.22 -> 0.34 s), fresh pages cost their faults, and registered or not, gigabytes of ordinary memory
 * take 0.1 - 0.4 s to give back at the end; all of it unregistered (profiles/r05i_*): copies at 7 GB/s, 1.05 s. */
#define TEXT_BUFS_AHEAD 0

/* A long plain input is not read into a host buffer of its size: slices of it go through a small ring of page-locked
 * buffers into the context's device text buffer -- STREAM_READERS slices are read at a time (pread, one thread each) while
 * the copies of the slices before them run.  What the host holds is STREAM_BUFS x STREAM_SLICE bytes (128 MiB), whatever
 * the file's size; page-locking memory costs ~0.25 s per GB, which is why the ring is small. */
static uint64_t STREAM_MIN = 256ull << 20;  /* files from this size on (KSSD_STREAM_MIN, bytes) */
static uint64_t STREAM_MIN_GZ = 64ull << 20; /* gzip'ed files from this compressed size on (KSSD_STREAM_MIN_GZ) */
static uint64_t STREAM_SLICE = 8ull << 20; /* (KSSD_STREAM_SLICE, bytes; a multiple of 4096) */
#define STREAM_BUFS 16
#define STREAM_READERS 8 /* slices read at a time, one thread each (the ring holds two such groups) */
static void stream_env(void)
{
    const char *e = getenv("KSSD_STREAM_MIN");
    if (e) STREAM_MIN = strtoull(e, NULL, 10);
    e = getenv("KSSD_STREAM_MIN_GZ");
    if (e) STREAM_MIN_GZ = strtoull(e, NULL, 10);
    e = getenv("KSSD_STREAM_SLICE");
    if (e && strtoull(e, NULL, 10) >= 4096) STREAM_SLICE = strtoull(e, NULL, 10) / 4096 * 4096;
}
typedef struct {
    unsigned char *buf[STREAM_BUFS];
} stream_ring;

static uint64_t stream_file_in(kssd_gpu_ctx *ctx, stream_ring *ring, const char *path, uint64_t len)
{
    for (int b = 0; b < STREAM_BUFS; b++)
        if (!ring->buf[b] && !(ring->buf[b] = kssd_gpu_host_alloc(STREAM_SLICE))) die(ENOMEM, "out of page-locked memory");
    gck(kssd_gpu_text_reserve(ctx, len), "kssd_gpu_text_reserve");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) die(EIO, "%s: %s", path, kssd_host_strerror(KSSD_HOST_ERR_IO));
    const uint64_t n_slices = (len + STREAM_SLICE - 1) / STREAM_SLICE;
    int64_t ticket[STREAM_BUFS];
    uint64_t got[STREAM_BUFS], total = 0;
    for (int b = 0; b < STREAM_BUFS; b++) ticket[b] = -1;
    int short_read = 0;
    double t_wait = 0, t_read = 0, t_put = 0;
    for (uint64_t g0 = 0; g0 < n_slices && !short_read; g0 += STREAM_READERS) {
        const uint64_t g1 = g0 + STREAM_READERS < n_slices ? g0 + STREAM_READERS : n_slices;
        double t0 = now_s();
        for (uint64_t k = g0; k < g1; k++) /* the copies that last read these buffers (two groups ago) */
            if (ticket[k % STREAM_BUFS] >= 0) gck(kssd_gpu_text_wait(ctx, ticket[k % STREAM_BUFS]), "kssd_gpu_text_wait");
        t_wait += now_s() - t0;
        t0 = now_s();
        int io_err = 0;
#pragma omp parallel for num_threads(STREAM_READERS) schedule(static, 1) reduction(| : io_err)
        for (uint64_t k = g0; k < g1; k++) {
            const uint64_t at = k * STREAM_SLICE, want = at + STREAM_SLICE <= len ? STREAM_SLICE : len - at;
            uint64_t n = 0;
            while (n < want) {
                const ssize_t r = pread(fd, ring->buf[k % STREAM_BUFS] + n, (size_t)(want - n), (off_t)(at + n));
                if (r < 0) { io_err |= 1; break; }
                if (r == 0) break;
                n += (uint64_t)r;
            }
            got[k % STREAM_BUFS] = n;
        }
        if (io_err) die(EIO, "%s: %s", path, kssd_host_strerror(KSSD_HOST_ERR_IO));
        t_read += now_s() - t0;
        t0 = now_s();
        for (uint64_t k = g0; k < g1 && !short_read; k++) {
            const int b = (int)(k % STREAM_BUFS);
            const int64_t t = kssd_gpu_text_put(ctx, k * STREAM_SLICE, ring->buf[b], got[b]);
            if (t < 0) gck((int)t, "kssd_gpu_text_put");
            ticket[b] = t;
            total += got[b];
            if (k * STREAM_SLICE + got[b] < (k + 1 < n_slices ? (k + 1) * STREAM_SLICE : len)) short_read = 1; /* (a file that shrank meanwhile) */
        }
        t_put += now_s() - t0;
    }
    close(fd);
    if (getenv("KSSD_TIMING"))
        fprintf(stderr, "{\"kssd_timing\": \"stream\", \"bytes\": %llu, \"slices\": %llu, \"s_wait_copies\": %.6f, \"s_pread\": %.6f, \"s_put\": %.6f}\n",
                (unsigned long long)total, (unsigned long long)n_slices, t_wait, t_read, t_put);
    return total;
}

/* The same for a gzip'ed input: zlib inflates straight into the ring's slices (one thread: a gzip stream has no entry
 * points), the copies run under the inflating.  The device buffer starts from an estimate -- the size the file's trailer
 * states (modulo 2^32, and only the last member's) or four times the compressed size, whichever is larger -- and grows
 * (kssd_gpu_text_reserve keeps its content) when the stream turns out longer. */
static uint64_t stream_gz_in(kssd_gpu_ctx *ctx, stream_ring *ring, const char *path, uint64_t gz_size)
{
    for (int b = 0; b < STREAM_BUFS; b++)
        if (!ring->buf[b] && !(ring->buf[b] = kssd_gpu_host_alloc(STREAM_SLICE))) die(ENOMEM, "out of page-locked memory");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) die(EIO, "%s: %s", path, kssd_host_strerror(KSSD_HOST_ERR_IO));
    uint64_t room = gz_size * 4;
    unsigned char tr[4];
    if (gz_size >= 4 && pread(fd, tr, 4, (off_t)(gz_size - 4)) == 4) {
        const uint64_t isize = (uint64_t)tr[0] | ((uint64_t)tr[1] << 8) | ((uint64_t)tr[2] << 16) | ((uint64_t)tr[3] << 24);
        if (isize > room) room = isize;
    }
    room += room / 16 + (1u << 20);
    gck(kssd_gpu_text_reserve(ctx, room), "kssd_gpu_text_reserve");
    gzFile g = gzdopen(fd, "rb");
    if (!g) { close(fd); die(EIO, "%s: %s", path, kssd_host_strerror(KSSD_HOST_ERR_IO)); }
    gzbuffer(g, 1 << 20);
    int64_t ticket[STREAM_BUFS];
    for (int b = 0; b < STREAM_BUFS; b++) ticket[b] = -1;
    uint64_t total = 0, k = 0;
    double t_wait = 0, t_inflate = 0;
    for (;; k++) {
        const int b = (int)(k % STREAM_BUFS);
        double t0 = now_s();
        if (ticket[b] >= 0) gck(kssd_gpu_text_wait(ctx, ticket[b]), "kssd_gpu_text_wait");
        t_wait += now_s() - t0;
        t0 = now_s();
        uint64_t n = 0;
        while (n < STREAM_SLICE) {
            const uint64_t want = STREAM_SLICE - n;
            const int r = gzread(g, ring->buf[b] + n, (unsigned)(want > (1u << 30) ? (1u << 30) : want));
            if (r < 0) { gzclose(g); die(EIO, "%s: %s", path, kssd_host_strerror(KSSD_HOST_ERR_IO)); }
            if (r == 0) break;
            n += (uint64_t)r;
        }
        t_inflate += now_s() - t0;
        if (n == 0) break;
        if (total + n > room) { /* longer than the estimate */
            room = (total + n) * 2;
            gck(kssd_gpu_text_reserve(ctx, room), "kssd_gpu_text_reserve");
        }
        const int64_t t = kssd_gpu_text_put(ctx, total, ring->buf[b], n);
        if (t < 0) gck((int)t, "kssd_gpu_text_put");
        ticket[b] = t;
        total += n;
        if (n < STREAM_SLICE) break;
    }
    gzclose(g);
    if (getenv("KSSD_TIMING"))
        fprintf(stderr, "{\"kssd_timing\": \"stream_gz\", \"bytes\": %llu, \"compressed\": %llu, \"s_wait_copies\": %.6f, \"s_inflate\": %.6f}\n",
                (unsigned long long)total, (unsigned long long)gz_size, t_wait, t_inflate);
    return total;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* the job's genomes through the device: FASTA text is tokenised there, FASTQ batches arrive tokenised; pos may be NULL */
static int job_sketch(kssd_gpu_ctx *ctx, job *j, stream_ring *ring, const filelist *fl, uint32_t flags, uint32_t min_occ, uint64_t **off,
                      uint32_t **ids, uint32_t **pos, int64_t *bad)
{
    if ((j->tx || j->streamed) && !j->own_b) {
        const unsigned char *text = j->tx ? j->tx->p : NULL; /* NULL: already in the context's device buffer */
        if (j->is_fq)
            return kssd_gpu_sketch_fastq_text(ctx, text, j->toff, j->tlen, (uint32_t)j->n_files, flags, min_occ, off, ids, pos, j->lines, bad);
        return kssd_gpu_sketch_fasta_text(ctx, text, j->toff, j->tlen, (uint32_t)j->n_files, flags, min_occ, off, ids, pos, bad);
    }
    kssd_batch *b = j->own_b ? j->own_b : j->b;
    if (pos) return kssd_gpu_sketch_batch_pos(ctx, kssd_batch_packed(b), kssd_batch_mask(b), kssd_batch_chunk_off(b), kssd_batch_n_genomes(b),
                                              flags, min_occ, off, ids, pos, bad);
    return kssd_gpu_sketch_batch(ctx, kssd_batch_packed(b), kssd_batch_mask(b), kssd_batch_chunk_off(b), kssd_batch_n_genomes(b), flags,
                                 min_occ, off, ids, bad);
}

static void process_job(kssd_gpu_ctx *ctx, stream_ring *ring, job *j, const dist_opt *o, filelist *fl, uint32_t hashsize, uint32_t hashlimit, double *t_call,
                        kssd_gpu_resident *res, uint32_t res_first)
{   /* res != NULL: the job's sketches also stay on the device, as slots first_file - res_first .. of res (--allpairs) */
    const double tc0 = now_s();
    const int is_fq = j->is_fq;
    const uint32_t first_file = (uint32_t)j->first_file;
    uint32_t n = (uint32_t)j->n_files;
    uint32_t flags = is_fq ? (KSSD_SKETCH_KEEP_ZERO | KSSD_SKETCH_NO_CAPACITY) : (o->u ? KSSD_SKETCH_UNIQ : KSSD_SKETCH_FASTA);
    uint32_t min_occ = is_fq ? (uint32_t)o->kmerocrs : 1u;
    if (o->abundance) { /* mt_shortreads2koc (iseq2comem.c:554-615): every k-mer kept, -n not looked at, crowding is fatal */
        flags = KSSD_SKETCH_KEEP_ZERO;
        min_occ = 1;
    }
    uint64_t *off = NULL;
    uint32_t *ids = NULL, *pos = NULL;
    int64_t bad = -1;
    const uint32_t passes = kssd_gpu_tuple_passes(ctx);
    if (passes > 1 && res) die(ENOTSUP, "--allpairs with k - drlevel = 9: a directory of 256 components is not searched (the reference's own stage II does not survive it)");
    if (j->streamed && !j->uploaded) { /* the text's length (a gzip'ed input: only known now) decides what follows */
        const double tu0 = now_s();
        j->tlen[0] = j->streamed == 2 ? stream_gz_in(ctx, ring, fl->path[j->first_file], j->tlen[0])
                                      : stream_file_in(ctx, ring, fl->path[j->first_file], j->tlen[0]);
        j->uploaded = 1;
        *t_call += now_s() - tu0;
    }
    /* first positions (for the reference's exact file order) need genomes below 2^32 positions */
    int with_pos = 1;
    for (uint32_t g = 0; g < n; g++) {
        const uint64_t chunks = (j->tx || j->streamed) ? (j->tlen[g] + KSSD_CHUNK_BASES - 1) / KSSD_CHUNK_BASES
                                      : kssd_batch_chunk_off(j->b)[g + 1] - kssd_batch_chunk_off(j->b)[g];
        if (chunks >= (1ull << 20)) with_pos = 0;
    }
    /* fastq -n >= 2 and -u drop ids at dump time that sit in the reference's table all the same (and shift the probes of
     * later ids): for a byte-identical file the replay needs ALL distinct ids with their first positions, and which of
     * them are kept -- a pass without the keep rule, and a pass for the occurrences */
    const int replay_all = with_pos && !o->abundance && ((is_fq && min_occ > 1) || (!is_fq && o->u));
    const uint32_t keep_rule_occ = min_occ;
    if (replay_all) {
        flags &= ~KSSD_SKETCH_UNIQ;
        min_occ = 1;
    }
    if (passes > 1) {
        if (!with_pos) die(ENOTSUP, "genomes of 2^32 positions and more with k - drlevel = 9 are not built");
        gck(kssd_gpu_set_tuple_pass(ctx, 0), "kssd_gpu_set_tuple_pass");
    }
    int rc = job_sketch(ctx, j, ring, fl, flags, min_occ, &off, &ids, with_pos ? &pos : NULL, &bad);
    if (rc == KSSD_ERR_UNSUPPORTED && (j->tx || j->streamed) && is_fq) {
        /* an input the device tokeniser does not do exactly as fastq2co (no complete record, a line its fgets() buffer
         * splits, NUL or 8-bit bytes, with -Q a quality line shorter than its bases): the whole job through the host tokeniser */
        j->own_b = kssd_batch_create();
        uint64_t *maxpos = malloc((size_t)n * sizeof *maxpos);
        if (!j->own_b || !maxpos) die(ENOMEM, "out of memory");
        for (uint32_t g = 0; g < n; g++) maxpos[g] = j->tlen[g];
        uint32_t first = 0;
        if (kssd_batch_reserve(j->own_b, n, maxpos, &first)) die(ENOMEM, "out of memory");
        free(maxpos);
        int trc = 0;
        if (j->streamed) { /* (the bytes are on the device only) */
            unsigned char *txt = NULL;
            size_t cap = 0, len = 0;
            trc = kssd_slurp_reuse(fl->path[first_file], &txt, &cap, &len);
            if (!trc) trc = kssd_batch_fill_text(j->own_b, first, o->abundance ? 2 : 1, txt, len, o->kmerqlty, &j->lines[0]);
            if (trc == KSSD_HOST_ERR_EMPTY) trc = 0;
            free(txt);
        } else {
#pragma omp parallel for num_threads(WORKER_OMP) schedule(dynamic, 1) reduction(| : trc)
            for (uint32_t g = 0; g < n; g++) {
                const int r = kssd_batch_fill_text(j->own_b, first + g, o->abundance ? 2 : 1, j->tx->p + j->toff[g], j->tlen[g], o->kmerqlty, &j->lines[g]);
                if (r && r != KSSD_HOST_ERR_EMPTY) trc |= 1;
            }
        }
        if (trc) die(EIO, "%s ...: the host tokeniser failed", fl->path[first_file]);
        rc = job_sketch(ctx, j, ring, fl, flags, min_occ, &off, &ids, with_pos ? &pos : NULL, &bad);
    }
    if ((j->tx || j->streamed) && is_fq && !o->abundance) /* (mt_shortreads2koc prints no count) */
        for (uint32_t g = 0; g < n; g++) printf("%llu reads detected\n", (unsigned long long)j->lines[g]);
    if (rc == KSSD_ERR_INPUT) /* the host tokeniser's KSSD_HOST_ERR_HEADER (iseq2comem.c:233) */
        die(EIO, "%s: %s", fl->path[first_file + (bad >= 0 ? bad : 0)], kssd_host_strerror(KSSD_HOST_ERR_HEADER));
    if (rc == KSSD_ERR_CAPACITY)
        die(ENOSPC, "%s: the context space is too crowd, try rerun the program using -k%d", fl->path[first_file + (bad >= 0 ? bad : 0)], o->k + 1);
    gck(rc, "sketch");
    /* the ids this call left on the device (ascending per genome) are the job's sketches unless a keep rule is replayed on the
     * host below (-u, fastq -n > 1: the call was made without the rule) */
    if (res && !replay_all) gck(kssd_gpu_resident_put(res, ctx, first_file - res_first, n), "kssd_gpu_resident_put");
    if (passes > 1) {
        /* 36-bit tuples: the passes 1 .. 15 over the same job (pass s: the tuples with low bits s, ids = tuple >> 4), then
         * every genome's tuples of all passes in the reference's file order (its ONE table holds the whole tuples) */
        uint64_t *poff[16] = {off};
        uint32_t *pids[16] = {ids}, *ppos[16] = {pos};
        for (uint32_t sp = 1; sp < passes; sp++) {
            gck(kssd_gpu_set_tuple_pass(ctx, sp), "kssd_gpu_set_tuple_pass");
            rc = kssd_gpu_sketch_again(ctx, flags, min_occ, &poff[sp], &pids[sp], &ppos[sp], &bad); /* one scan for all sixteen */
            if (rc == KSSD_ERR_CAPACITY)
                die(ENOSPC, "%s: the context space is too crowd, try rerun the program using -k%d", fl->path[first_file + (bad >= 0 ? bad : 0)], o->k + 1);
            gck(rc, "sketch (tuple pass)");
        }
        /* -u, fastq -n > 1 (the keep rule is replayed on the host: the tuples the dump drops sit in the reference's table all the
         * same) and -A need every tuple's number of occurrences: sixteen more passes over the candidates of the same scan */
        const int want_counts = replay_all || o->abundance;
        uint64_t *coff[16] = {0};
        uint32_t *cids[16] = {0}, *ccnt[16] = {0};
        for (uint32_t sp = 0; want_counts && sp < passes; sp++) {
            gck(kssd_gpu_set_tuple_pass(ctx, sp), "kssd_gpu_set_tuple_pass");
            gck(kssd_gpu_sketch_again(ctx, flags | KSSD_SKETCH_NO_CAPACITY | KSSD_SKETCH_COUNTS, 1u, &coff[sp], &cids[sp], &ccnt[sp], &bad), "sketch (occurrences, tuple pass)");
            if (coff[sp][n] != poff[sp][n]) die(EIO, "sketch (occurrences): %llu ids against %llu", (unsigned long long)coff[sp][n], (unsigned long long)poff[sp][n]);
        }
        gck(kssd_gpu_set_tuple_pass(ctx, 0), "kssd_gpu_set_tuple_pass");
        uint64_t *moff = calloc((size_t)n + 1, sizeof *moff);
        if (!moff) die(ENOMEM, "out of memory");
        for (uint32_t g = 0; g < n; g++) {
            uint64_t m = 0;
            for (uint32_t sp = 0; sp < passes; sp++) m += poff[sp][g + 1] - poff[sp][g];
            moff[g + 1] = moff[g] + m;
            /* keycount > hashlimit over the ONE table of whole tuples (iseq2comem.c:261-263; -u :686-688; -A :598-600).  The device
             * applies the rule per pass (a sixteenth of the keys against the whole limit: it never fires first); here the passes'
             * distinct tuples are added up.  What is not added: the occurrences of the tuple 0 itself, which fasta2co counts one
             * by one (:255-263) -- they decide only for a genome within a handful of k-mers of 322 million distinct ones. */
            if ((!is_fq || o->abundance) && m > (uint64_t)hashlimit)
                die(ENOSPC, "%s: the context space is too crowd, try rerun the program using -k%d", fl->path[first_file + g], o->k + 1);
        }
        uint32_t *mids = malloc((size_t)(moff[n] ? moff[n] : 1) * 4);
        uint8_t *msub = malloc((size_t)(moff[n] ? moff[n] : 1));
        uint16_t *mcnt = o->abundance ? malloc((size_t)(moff[n] ? moff[n] : 1) * 2) : NULL;
        uint64_t *kept = calloc((size_t)n + 1, sizeof *kept);
        if (!mids || !msub || !kept || (o->abundance && !mcnt)) die(ENOMEM, "out of memory");
#pragma omp parallel for num_threads(WORKER_OMP) schedule(dynamic, 16)
        for (uint32_t g = 0; g < n; g++) {
            uint64_t m = moff[g + 1] - moff[g];
            uint64_t *t = malloc((size_t)(m ? m : 1) * 8);
            uint32_t *p = malloc((size_t)(m ? m : 1) * 4);
            uint8_t *keep = replay_all ? malloc((size_t)(m ? m : 1)) : NULL;
            uint64_t *ct = o->abundance ? malloc((size_t)(m ? m : 1) * 8) : NULL; /* tuple << 16 | occurrences, ascending: looked up behind the ordering */
            if (!t || !p || (replay_all && !keep) || (o->abundance && !ct)) die(ENOMEM, "out of memory");
            uint64_t w = 0;
            for (uint32_t sp = 0; sp < passes; sp++)
                for (uint64_t i = poff[sp][g], c = want_counts ? coff[sp][g] : 0; i < poff[sp][g + 1]; i++, c++) {
                    t[w] = ((uint64_t)pids[sp][i] << 4) | sp;
                    p[w] = ppos[sp][i];
                    if (want_counts && cids[sp][c] != pids[sp][i]) die(EIO, "sketch (occurrences): the two passes list different ids");
                    if (keep) keep[w] = is_fq ? ccnt[sp][c] >= keep_rule_occ : ccnt[sp][c] == 1; /* both passes list a pass's ids ascending */
                    if (ct) ct[w] = (t[w] << 16) | (ccnt[sp][c] & 0xFFFFu);
                    w++;
                }
            if (keep) {
                m = kssd_slot_order_pos64_keep(t, p, keep, m, hashsize); /* kept tuples to the front, file order */
                if (m == UINT64_MAX) die(ENOMEM, "out of memory");
            } else if (kssd_slot_order_pos64(t, p, m, hashsize)) {
                die(ENOMEM, "out of memory");
            }
            kept[g + 1] = m;
            if (ct) { /* the counts follow their tuples: ct in ascending tuple order, a bisection per tuple */
                const uint64_t all = moff[g + 1] - moff[g];
                qsort(ct, (size_t)all, sizeof *ct, cmp_u64);
                for (uint64_t i = 0; i < m; i++) {
                    uint64_t lo = 0, hi = all;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if ((ct[mid] >> 16) < t[i]) lo = mid + 1;
                        else hi = mid;
                    }
                    mcnt[moff[g] + i] = (uint16_t)(ct[lo] & 0xFFFFu);
                }
            }
            for (uint64_t i = 0; i < m; i++) {
                mids[moff[g] + i] = (uint32_t)(t[i] >> 4);
                msub[moff[g] + i] = (uint8_t)(t[i] & 15u);
            }
            free(t);
            free(p);
            free(keep);
            free(ct);
        }
        if (replay_all) { /* close the gaps the dropped tuples leave */
            uint64_t at = 0;
            for (uint32_t g = 0; g < n; g++) {
                const uint64_t m = kept[g + 1];
                memmove(mids + at, mids + moff[g], (size_t)m * 4);
                memmove(msub + at, msub + moff[g], (size_t)m);
                kept[g + 1] = at + m;
                at += m;
            }
            memcpy(moff, kept, ((size_t)n + 1) * sizeof *moff);
        }
        free(kept);
        for (uint32_t sp = 0; sp < passes; sp++) {
            kssd_gpu_free(poff[sp]);
            kssd_gpu_free(pids[sp]);
            kssd_gpu_free(ppos[sp]);
            kssd_gpu_free(coff[sp]);
            kssd_gpu_free(cids[sp]);
            kssd_gpu_free(ccnt[sp]);
        }
        j->counts = mcnt;
        *t_call += now_s() - tc0;
        j->off = moff;
        j->ids = mids;
        j->sub = msub;
        return;
    }
    if (replay_all) {
        uint64_t *coff = NULL;
        uint32_t *cids = NULL, *ccnt = NULL;
        gck(job_sketch(ctx, j, ring, fl, flags | KSSD_SKETCH_NO_CAPACITY | KSSD_SKETCH_COUNTS, 1u, &coff, &cids, &ccnt, &bad), "sketch (occurrences)");
        if (coff[n] != off[n]) die(EIO, "sketch (occurrences): %llu ids against %llu", (unsigned long long)coff[n], (unsigned long long)off[n]);
        uint64_t *koff = calloc((size_t)n + 1, sizeof *koff);
        if (!koff) die(ENOMEM, "out of memory");
#pragma omp parallel for num_threads(WORKER_OMP) schedule(dynamic, 16)
        for (uint32_t g = 0; g < n; g++) {
            const uint64_t m = off[g + 1] - off[g];
            uint8_t *keep = malloc(m ? m : 1);
            if (!keep) die(ENOMEM, "out of memory");
            for (uint64_t i = 0; i < m; i++) /* both passes list a genome's distinct ids ascending */
                keep[i] = is_fq ? ccnt[coff[g] + i] >= keep_rule_occ : ccnt[coff[g] + i] == 1;
            koff[g + 1] = kssd_slot_order_pos_keep(ids + off[g], pos + off[g], keep, m, hashsize); /* kept ids to the front, file order */
            if (koff[g + 1] == UINT64_MAX) die(ENOMEM, "out of memory");
            free(keep);
        }
        uint64_t at = 0;
        for (uint32_t g = 0; g < n; g++) { /* close the gaps the dropped ids leave */
            const uint64_t m = koff[g + 1];
            memmove(ids + at, ids + off[g], (size_t)m * 4);
            koff[g + 1] = at + m;
            at += m;
        }
        memcpy(off, koff, ((size_t)n + 1) * sizeof *off);
        free(koff);
        kssd_gpu_free(coff);
        kssd_gpu_free(cids);
        kssd_gpu_free(ccnt);
        with_pos = -1; /* already in file order */
        if (res) gck(kssd_gpu_resident_put_host(res, first_file - res_first, n, off, ids), "kssd_gpu_resident_put_host");
    }
    *t_call += now_s() - tc0;
    /* -A: a second pass over the same batch returns the occurrences of every id (ids ascending inside a genome,
     * the same set as above) */
    uint64_t *aoff = NULL;
    uint32_t *aids = NULL, *acnt = NULL;
    if (o->abundance) {
        gck(job_sketch(ctx, j, ring, fl, KSSD_SKETCH_KEEP_ZERO | KSSD_SKETCH_NO_CAPACITY | KSSD_SKETCH_COUNTS, 1u, &aoff, &aids, &acnt, &bad),
            "sketch (abundances)");
        if (aoff[n] != off[n]) die(EIO, "sketch (abundances): %llu ids against %llu", (unsigned long long)aoff[n], (unsigned long long)off[n]);
    }
    /* file order inside a genome = the reference's hash-slot order, insertions replayed in sequence order (or, for
     * a genome of 2^32 positions and more, in ascending id order: exact unless two of its ids probe the same slot) */
#pragma omp parallel for num_threads(WORKER_OMP) schedule(dynamic, 16)
    for (uint32_t g = 0; g < n; g++) {
        int orc = 0;
        if (with_pos > 0) orc = kssd_slot_order_pos(ids + off[g], pos + off[g], off[g + 1] - off[g], hashsize);
        else if (with_pos == 0) orc = kssd_slot_order(ids + off[g], off[g + 1] - off[g], hashsize);
        if (orc) die(ENOMEM, "out of memory");
    }
    if (o->abundance) {
        uint16_t *counts = malloc((size_t)(off[n] ? off[n] : 1) * 2);
        if (!counts) die(ENOMEM, "out of memory");
        int bad_follow = 0;
#pragma omp parallel for num_threads(WORKER_OMP) schedule(dynamic, 16) reduction(| : bad_follow)
        for (uint32_t g = 0; g < n; g++) {
            const uint64_t m = off[g + 1] - off[g];
            if (aoff[g + 1] - aoff[g] != m) { bad_follow |= 1; continue; }
            uint16_t *c16 = malloc((m ? m : 1) * 2);
            for (uint64_t i = 0; i < m; i++) c16[i] = (uint16_t)acnt[aoff[g] + i]; /* saturated at 65535 on the device */
            bad_follow |= kssd_counts_follow(aids + aoff[g], c16, ids + off[g], counts + off[g], m) != 0;
            free(c16);
        }
        if (bad_follow) die(EIO, "sketch (abundances): the two passes disagree");
        kssd_gpu_free(aoff);
        kssd_gpu_free(aids);
        kssd_gpu_free(acnt);
        j->counts = counts;
    }
    j->off = off;
    j->ids = ids;
    kssd_gpu_free(pos);
}

/* the stage-I pipeline: the main thread reads and tokenises waves of files on all host threads; device workers (two per
 * GPU, each with its own context and stream, so that one batch's transfer runs under the other's kernels) sketch the
 * batches; batches are recycled through a small pool of page-locked buffers, which also bounds what is in flight */
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    job *todo_head[64], *todo_tail[64]; /* tokenised, waiting for a device: one queue, or one per device of the list (--allpairs) */
    int n_q;
    kssd_gpu_resident **res;    /* --allpairs: per device, what it has sketched */
    const uint32_t *first;      /* ... and the first input of every device (kssd_shard_plan) */
    job *done;                  /* sketched */
    kssd_batch **pool;          /* free batches */
    textbuf **tpool;            /* free text buffers */
    int n_pool, n_tpool, closed;
    int n_text_made;            /* text buffers in existence (the pool holds the free ones) */
    const dist_opt *o;
    filelist *fl;
    uint32_t hashsize, hashlimit;
    kssd_shuf_hdr hdr;
    const uint32_t *accepted; /* the .shuf's accepted sub-contexts (load_shuf) */
    uint32_t n_accepted;
    double t_ctx_destroy;
    double t_ctx;   /* the slowest worker's context creation (HIP initialisation, code object load, table upload) */
    double t_gpu;   /* summed over the workers: seconds inside process_job */
    double t_call;  /* ... of which inside the kssd_gpu_sketch_batch* calls (H2D, kernels, D2H) */
} pipeline;

typedef struct {
    pipeline *pl;
    int device, q; /* the device, and the queue it takes jobs from */
    pthread_t th;
} worker;

static void *worker_main(void *arg)
{
    worker *w = arg;
    pipeline *pl = w->pl;
    kssd_gpu_ctx *ctx = NULL;
    const double tc0 = now_s();
    {
        const int have = kssd_gpu_device_count();
        if (have <= 0) die(ENODEV, "kssd_gpu_create: %s", kssd_gpu_strerror(KSSD_ERR_NO_DEVICE));
        if (w->device >= have) die(ENODEV, "device %d of the list: only %d device(s) visible", w->device, have);
    }
    gck(kssd_gpu_create_compact(&ctx, &pl->hdr, pl->accepted, pl->n_accepted, w->device), "kssd_gpu_create");
    g_runtime_ready = 1;
    if (pl->o->abundance) gck(kssd_gpu_set_fastq_reads(ctx, 1), "kssd_gpu_set_fastq_reads");
    else if (pl->o->kmerqlty > 0 && pl->o->kmerqlty <= 127) gck(kssd_gpu_set_fastq_quality(ctx, pl->o->kmerqlty), "kssd_gpu_set_fastq_quality");
    pthread_mutex_lock(&pl->mu);
    if (now_s() - tc0 > pl->t_ctx) pl->t_ctx = now_s() - tc0;
    pthread_mutex_unlock(&pl->mu);
    stream_ring ring = {{0}};
    for (;;) {
        pthread_mutex_lock(&pl->mu);
        while (!pl->todo_head[w->q] && !pl->closed) pthread_cond_wait(&pl->cv, &pl->mu);
        job *j = pl->todo_head[w->q];
        if (j) {
            pl->todo_head[w->q] = j->next;
            if (!pl->todo_head[w->q]) pl->todo_tail[w->q] = NULL;
        }
        pthread_mutex_unlock(&pl->mu);
        if (!j) break;
        const double t0 = now_s();
        double tcall = 0;
        if (j->tx) textbuf_lock(j->tx);
        process_job(ctx, &ring, j, pl->o, pl->fl, pl->hashsize, pl->hashlimit, &tcall, pl->res ? pl->res[w->q] : NULL, pl->res ? pl->first[w->q] : 0u);
        const double dt = now_s() - t0;
        if (j->b) kssd_batch_clear(j->b);
        if (j->own_b) kssd_batch_destroy(j->own_b);
        j->own_b = NULL;
        free(j->lines);
        j->lines = NULL;
        pthread_mutex_lock(&pl->mu);
        if (j->b) pl->pool[pl->n_pool++] = j->b; /* the buffer goes back to the tokeniser */
        if (j->tx) pl->tpool[pl->n_tpool++] = j->tx;
        j->b = NULL;
        j->tx = NULL;
        free(j->toff);
        free(j->tlen);
        j->toff = j->tlen = NULL;
        j->next = pl->done;
        pl->done = j;
        pl->t_gpu += dt;
        pl->t_call += tcall;
        pthread_cond_broadcast(&pl->cv);
        pthread_mutex_unlock(&pl->mu);
    }
    const double td0 = now_s();
    for (int b = 0; b < STREAM_BUFS; b++)
        if (ring.buf[b]) kssd_gpu_host_free(ring.buf[b]);
    kssd_gpu_destroy(ctx);
    pthread_mutex_lock(&pl->mu);
    if (now_s() - td0 > pl->t_ctx_destroy) pl->t_ctx_destroy = now_s() - td0;
    pthread_mutex_unlock(&pl->mu);
    return NULL;
}

static int cmp_job(const void *a, const void *b)
{
    const job *x = *(job *const *)a, *y = *(job *const *)b;
    return x->first_file < y->first_file ? -1 : x->first_file > y->first_file;
}

/* dist --byread (run_stageI command_dist.c:267-273 + reads2mco iseq2comem.c:78-186): every input file is one genome on
 * the device; KSSD_SKETCH_BY_POS returns its whole k-mer stream with positions, the tokeniser's cut points turn the
 * positions into reads.  Like the reference, every file overwrites combco.* of the one before.  (One deviation: a
 * gzip'ed input is unpacked; the reference opens --byread inputs without zcat and scans the compressed bytes.) */
static void sketch_files_byread(const dist_opt *o, filelist *fl, const char *outdir)
{
    shuf_core sc;
    load_shuf(o, &sc);
    const kssd_shuf shuf = sc.h;
    kssd_derived d;
    if (kssd_derive(&d, shuf.k, shuf.subk, shuf.drlevel))
        die(EINVAL, "get_hashsz(): primer_ind out of range(0 ~ 24): this might caused by too small or too large k (k=%d, level=%d)", shuf.k, shuf.drlevel);
    printf("rand_id=%d\thalf_ctx_len=%d\thashsize=%d\thashlimit=%d\n", shuf.id, shuf.k, (int)d.hashsize, (int)d.hashlimit);
    kssd_shuf_hdr hdr = {shuf.id, shuf.k, shuf.subk, shuf.drlevel};
    gck(kssd_gpu_create_compact(&g_ctx, &hdr, sc.accepted, sc.n_accepted, o->device), "kssd_gpu_create");
    free(sc.accepted);
    /* the device tokenises the text and says where the reads begin (kssd_gpu_fasta_read_starts); KSSD_HOST_BYREAD=1 keeps the
     * host scanner (kssd_batch_add_fasta_reads), the two are compared by tests/test_gpu_cli.py */
    const int on_host = getenv("KSSD_HOST_BYREAD") != NULL;
    kssd_batch *b = kssd_batch_create();
    unsigned char *txt = NULL;
    size_t cap = 0;
    for (int i = 0; i < fl->n; i++) {
        printf("decomposing %s by reads\n", fl->path[i]);
        size_t len = 0;
        int rc = kssd_slurp_reuse(fl->path[i], &txt, &cap, &len);
        if (rc) die(EIO, "reads2mco():%s: %s", fl->path[i], kssd_host_strerror(rc));
        if (len == 0) die(EIO, "reads2mco():eof or fread error file=%s", fl->path[i]);
        uint64_t *cuts = NULL, n_reads = 0;
        uint64_t *off = NULL;
        uint32_t *ids = NULL, *pos = NULL;
        int64_t bad = -1;
        if (on_host) {
            rc = kssd_batch_add_fasta_reads(b, txt, len, &cuts, &n_reads);
            if (rc == KSSD_HOST_ERR_HEADER) die(EIO, "fasta2co(): can not find seqences head start from '>' 0");
            if (rc) die(EIO, "%s: %s", fl->path[i], kssd_host_strerror(rc));
            gck(kssd_gpu_sketch_batch_pos(g_ctx, kssd_batch_packed(b), kssd_batch_mask(b), kssd_batch_chunk_off(b), 1, KSSD_SKETCH_BY_POS,
                                          1u, &off, &ids, &pos, &bad),
                "sketch (by read)");
        } else {
            const uint64_t t_off = 0, t_len = len;
            rc = kssd_gpu_sketch_fasta_text(g_ctx, txt, &t_off, &t_len, 1, KSSD_SKETCH_BY_POS, 1u, &off, &ids, &pos, &bad);
            if (rc == KSSD_ERR_INPUT) die(EIO, "fasta2co(): can not find seqences head start from '>' 0");
            gck(rc, "sketch (by read)");
            gck(kssd_gpu_fasta_read_starts(g_ctx, 0, &cuts, &n_reads), "read starts");
        }
        rc = kssd_byread_write(outdir, (uint32_t)hdr.id, hdr.k, hdr.drlevel, (const char (*)[KSSD_PATHLEN])fl->path, (uint32_t)fl->n, ids,
                               pos, off[1], cuts, n_reads);
        if (rc) die(EIO, "%s: %s", outdir, kssd_host_strerror(rc));
        kssd_gpu_free(off);
        kssd_gpu_free(ids);
        kssd_gpu_free(pos);
        if (on_host) kssd_host_free(cuts);
        else kssd_gpu_free(cuts);
        kssd_batch_clear(b);
        printf("decomposing %s by reads is complete!\n", fl->path[i]);
    }
    free(txt);
    kssd_batch_destroy(b);
    kssd_gpu_destroy(g_ctx);
    g_ctx = NULL;
}

void sketch_files(const dist_opt *o, filelist *fl, const char *outdir)
{
    if (fl->n == 0) die(EINVAL, "no valid input .fas/.fq file");
    if (o->pipecmd[0]) die(ENOTSUP, "--pipecmd is outside the GPU hot path of this build (SURVEY.md section 8f)");
    if (o->byread) {
        sketch_files_byread(o, fl, outdir);
        return;
    }
    int abundance = o->abundance;
    if (abundance) /* command_dist.c:297-301: one non-FASTQ input closes the mode (here: for the whole run, up front) */
        for (int i = 0; i < fl->n; i++)
            if (!has_fmt(fl->path[i], fq_fmt)) {
                printf("Warning: close abundance mode (-A) since non-fastq file input.\n");
                abundance = 0;
                break;
            }
    dist_opt oa = *o;
    oa.abundance = abundance;
    o = &oa;
    if (abundance) printf("running mt_shortreads2koc()\n");
    const double t_start = now_s();
    /* --allpairs: the inputs are dealt out to the devices of the list in contiguous runs (kssd_shard_plan) -- a device sketches
     * its run, keeps the sketches, and owns their rows of the matrix later.  Decided here, before anything touches a GPU: a list
     * that names a device twice is refused with the reason (RCCL admits one rank per device). */
    const int n_dev = o->n_devs;
    uint32_t first[65] = {0};
    if (o->allpairs) {
        if (abundance) die(ENOTSUP, "--allpairs with -A: abundance sketches are not searched (mco_cbdco_nobin_dist reads plain sketches)");
        int ranks[64];
        for (int i = 0; i < n_dev; i++) ranks[i] = o->fake_ranks ? i : o->devs[i]; /* (KSSD_EXCHANGE_FAKE_RANKS: the ranks share a device on purpose) */
        if (kssd_shard_plan(ranks, n_dev, (uint32_t)fl->n, first))
            die(EINVAL, "--allpairs: the device list names a device twice (or none): one rank per device");
    }
    pthread_t warm;
    int warm_dev = o->devs[0];
    const int warming = pthread_create(&warm, NULL, warm_device, &warm_dev) == 0; /* under the .shuf read */
    int warm_joined = 0;
    shuf_core sc;
    load_shuf(o, &sc);
    const kssd_shuf shuf = sc.h;
    kssd_derived d;
    if (kssd_derive(&d, shuf.k, shuf.subk, shuf.drlevel))
        die(EINVAL, "get_hashsz(): primer_ind out of range(0 ~ 24): this might caused by too small or too large k (k=%d, level=%d)", shuf.k, shuf.drlevel);
    printf("rand_id=%d\thalf_ctx_len=%d\thashsize=%d\thashlimit=%d\n", shuf.id, shuf.k, (int)d.hashsize, (int)d.hashlimit);
    kssd_shuf_hdr hdr = {shuf.id, shuf.k, shuf.subk, shuf.drlevel};
    const double t_shuf = now_s() - t_start;
    /* (nobody waits for the runtime here: the workers' context creation does, on their own threads, and the main thread reads the
     * first inputs into ordinary memory meanwhile -- 0.1 - 0.2 s of hipInit used to stand in front of the first read) */
    pthread_t xwarm;
    xwarm_arg xa = {o->devs, n_dev};
    const int xwarming = o->allpairs && !o->fake_ranks && (n_dev > 1 || getenv("KSSD_EXCHANGE_ONE_RANK")) && pthread_create(&xwarm, NULL, warm_exchange, &xa) == 0;
    /* sketch workers per device (each with its own context and stream) and text buffers beyond one per worker: tuning knobs of the
     * pipeline, measured in profiles/r04I_e2e_workers_buffers.txt */
    /* (round 6, profiles/r06c_e2e_steady_probe.txt: once start-up is amortised -- 8 192 inputs, 512 jobs -- two workers leave the
     * device's copy engine idle a third of the time: a worker's job is copy -> kernels -> results -> slot order on the host, and
     * four of them keep a copy in flight: 5 150 -> 7 200 genomes/s in the steady state; a command of a thousand inputs is start-up
     * for half of its time and keeps two) */
    int wpd = fl->n >= 2048 ? 4 : 2, extra_bufs = 1;
    if (getenv("KSSD_WORKERS_PER_DEVICE")) wpd = atoi(getenv("KSSD_WORKERS_PER_DEVICE"));
    if (getenv("KSSD_TEXT_BUFFERS_EXTRA")) extra_bufs = atoi(getenv("KSSD_TEXT_BUFFERS_EXTRA"));
    if (wpd < 1) wpd = 1;
    if (wpd > 8) wpd = 8;
    if (extra_bufs < 1) extra_bufs = 1;
    if (extra_bufs > 32) extra_bufs = 32;
    const int n_workers = wpd * n_dev;
    const int threads = o->p > 0 ? o->p : 1;
    pipeline pl;
    memset(&pl, 0, sizeof pl);
    pthread_mutex_init(&pl.mu, NULL);
    pthread_cond_init(&pl.cv, NULL);
    pl.o = o;
    pl.fl = fl;
    pl.hashsize = d.hashsize;
    pl.hashlimit = d.hashlimit;
    pl.hdr = hdr;
    pl.accepted = sc.accepted;
    pl.n_accepted = sc.n_accepted;
    pl.n_q = o->allpairs ? n_dev : 1;
    kssd_gpu_resident *res[64] = {0};
    if (o->allpairs) {
        for (int d = 0; d < n_dev; d++) gck(kssd_gpu_resident_create(&res[d], o->devs[d], first[d + 1] - first[d]), "kssd_gpu_resident_create");
        pl.res = res;
        pl.first = first;
    }
    const int n_batches = n_workers + extra_bufs; /* one (or more) being filled, one per worker */
    pl.pool = calloc((size_t)n_batches, sizeof *pl.pool);
    for (int i = 0; i < n_batches; i++) {
        pl.pool[i] = kssd_batch_create_ex(kssd_gpu_host_alloc, kssd_gpu_host_free);
        if (!pl.pool[i]) die(ENOMEM, "out of memory");
    }
    pl.n_pool = n_batches;
    /* A job's text lives in ordinary memory that the worker registers with the runtime before its first copy (textbuf_lock): the
     * sixteen readers fault it in while they fill it, and registering 100 MB takes 5 ms on the worker's thread -- hipHostMalloc
     * takes 14 - 19 ms per buffer on the main thread, in front of the first five waves (profiles/r05v_startup_probe.txt; end to end
     * 0.55 - 0.62 -> 0.54 - 0.56 s, profiles/r05B_text_registered_probe.txt).  KSSD_TEXT_PAGE_LOCKED=1: hipHostMalloc. */
    const int text_page_locked = getenv("KSSD_TEXT_PAGE_LOCKED") != NULL;
    int text_ahead = TEXT_BUFS_AHEAD; /* (KSSD_TEXT_AHEAD: a tuning knob, 0 = no read-ahead: the first byte is read once the runtime is up) */
    if (getenv("KSSD_TEXT_AHEAD")) text_ahead = atoi(getenv("KSSD_TEXT_AHEAD"));
    if (text_ahead < 0) text_ahead = 0;
    if (text_ahead > 256) text_ahead = 256;
    pl.tpool = calloc((size_t)(n_batches + text_ahead), sizeof *pl.tpool);
    for (int i = 0; i < n_batches; i++) pl.tpool[i] = calloc(1, sizeof(textbuf));
    pl.n_tpool = pl.n_text_made = n_batches;
    worker *ws = calloc((size_t)n_workers, sizeof *ws);
    for (int i = 0; i < n_workers; i++) {
        ws[i].pl = &pl;
        ws[i].device = o->devs[i % n_dev];
        ws[i].q = o->allpairs ? i % n_dev : 0; /* (one queue for all workers unless the inputs belong to devices) */
        if (pthread_create(&ws[i].th, NULL, worker_main, &ws[i])) die(EAGAIN, "pthread_create");
    }

    const double t_workers_started = now_s();
    /* a device batch: at most ~0.5 Gbases (so that several are in flight and the transfers hide under the kernels) */
    const uint64_t max_chunks = 1ull << 17;
    double t_read = 0, t_tok = 0;
    double t_unpack_thr = 0, t_copy_thr = 0, t_wait_text = 0; /* summed over the host threads: inside the readers / unpackers, inside the copies into the job's text; the main thread waiting for a free text buffer */
    uint64_t n_bytes = 0;
    int done = 0, n_jobs = 0;
    /* A wave = `threads` consecutive inputs, read (and unpacked) one per host thread.  Its buffers are kept and reused (no fresh
     * memory per file).  While the HIP runtime is still starting -- 0.2 s in which nothing can be handed to a device, and nothing
     * page-locked be had -- the waves of gzip'ed inputs are unpacked AHEAD into further sets of buffers (KSSD_GZ_AHEAD sets, 8): their
     * bytes go through a scratch buffer into the job's text anyway, and sixteen threads inflate ~1 GB in that time; plain inputs are
     * only measured there (reading them ahead into ordinary memory did not pay: TEXT_BUFS_AHEAD). */
    typedef struct {
        unsigned char **txt;
        size_t *txt_cap, *len;
        int *trc, *direct; /* direct: plain file for the device tokeniser, read straight into the job's text buffer (2: streamed by the worker) */
        int i0, nw;
    } wave_buf;
    int gz_ahead = 8;
    if (getenv("KSSD_GZ_AHEAD")) gz_ahead = atoi(getenv("KSSD_GZ_AHEAD"));
    if (gz_ahead < 1) gz_ahead = 1; /* (two waves are read together) */
    if (gz_ahead > 64) gz_ahead = 64;
    wave_buf *waves = calloc((size_t)gz_ahead + 1, sizeof *waves); /* a ring: waves read and not yet turned into jobs */
    for (int w = 0; w <= gz_ahead; w++) {
        waves[w].txt = calloc((size_t)threads, sizeof(unsigned char *));
        waves[w].txt_cap = calloc((size_t)threads, sizeof(size_t));
        waves[w].len = calloc((size_t)threads, sizeof(size_t));
        waves[w].trc = calloc((size_t)threads, sizeof(int));
        waves[w].direct = calloc((size_t)threads, sizeof(int));
    }
    int w_head = 0, w_count = 0, next_read = 0;
    uint64_t *lines = calloc((size_t)threads, sizeof *lines);
    /* FASTQ text is tokenised on the device: fastq2co's framing with its quality rule (-Q) or, under -A, the framing of
     * mt_shortreads2koc (kssd_gpu_set_fastq_quality / _reads in worker_main); inputs only the reference's own fgets()
     * sequence reproduces come back (KSSD_ERR_UNSUPPORTED) and go through the host tokeniser */
    const int fq_dev = (o->abundance || (o->kmerqlty >= 0 && o->kmerqlty <= 127)) && !getenv("KSSD_HOST_FASTQ");
    stream_env();
    for (int i0 = 0; i0 < fl->n; i0 += threads) {
        /* read the next wave -- and, as long as the runtime is not up, the waves behind it.  Two waves go through one parallel region,
         * file i of the one and file i of the other on the same thread: two gzip'ed files are unpacked in step (kssd_slurp_reuse2) */
        while (next_read < fl->n && (w_count == 0 || (!g_runtime_ready && w_count <= gz_ahead))) {
            wave_buf *wv[2] = {NULL, NULL};
            int n_wv = 0;
            for (; n_wv < 2 && next_read < fl->n && w_count <= gz_ahead; n_wv++) {
                wave_buf *wb = wv[n_wv] = &waves[(w_head + w_count) % (gz_ahead + 1)];
                wb->i0 = next_read;
                wb->nw = (next_read + threads < fl->n ? next_read + threads : fl->n) - next_read;
                next_read += wb->nw;
                w_count++;
            }
            double t0 = now_s();
            /* read + gunzip.  gzip'ed files (and FASTQ files the host tokenises) go into the wave's scratch buffers; a plain file
             * the device tokenises is only measured here and read straight into page-locked memory below */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
            for (int i = 0; i < wv[0]->nw; i++) {
                const double ti0 = now_s();
                int slurp[2] = {0, 0};
                for (int v = 0; v < n_wv; v++) {
                    wave_buf *wb = wv[v];
                    if (i >= wb->nw) continue;
                    const char *path = fl->path[wb->i0 + i];
                    int gz = 0;
                    uint64_t sz = 0;
                    wb->direct[i] = 0;
                    wb->trc[i] = kssd_file_probe(path, &gz, &sz);
                    if (wb->trc[i]) continue;
                    const int dev_tok = fq_dev || !has_fmt(path, fq_fmt); /* the device tokenises it */
                    if (!gz && dev_tok) {
                        wb->direct[i] = 1;
                        wb->len[i] = (size_t)sz;
                    } else if (gz && dev_tok && sz >= STREAM_MIN_GZ) {
                        wb->direct[i] = 2; /* inflated by the device worker, slice by slice, on its way to the device */
                        wb->len[i] = (size_t)sz;
                    } else {
                        slurp[v] = 1;
                    }
                }
                if (slurp[0] && slurp[1]) {
                    const char *const path[2] = {fl->path[wv[0]->i0 + i], fl->path[wv[1]->i0 + i]};
                    unsigned char **buf[2] = {&wv[0]->txt[i], &wv[1]->txt[i]};
                    size_t *cap[2] = {&wv[0]->txt_cap[i], &wv[1]->txt_cap[i]}, *len[2] = {&wv[0]->len[i], &wv[1]->len[i]};
                    int rc[2];
                    kssd_slurp_reuse2(path, buf, cap, len, rc);
                    wv[0]->trc[i] = rc[0];
                    wv[1]->trc[i] = rc[1];
                } else {
                    for (int v = 0; v < 2; v++)
                        if (slurp[v]) wv[v]->trc[i] = kssd_slurp_reuse(fl->path[wv[v]->i0 + i], &wv[v]->txt[i], &wv[v]->txt_cap[i], &wv[v]->len[i]);
                }
                const double dti = now_s() - ti0;
#pragma omp atomic
                t_unpack_thr += dti;
            }
            t_read += now_s() - t0;
        }
        wave_buf *wb = &waves[w_head];
        w_head = (w_head + 1) % (gz_ahead + 1);
        w_count--;
        const int nw = wb->nw;
        unsigned char **txt = wb->txt;
        size_t *len = wb->len;
        int *trc = wb->trc, *direct = wb->direct;
        double t0;
        for (int i = 0; i < nw; i++)
            if (trc[i]) die(EIO, "%s: %s", fl->path[i0 + i], kssd_host_strerror(trc[i]));
        /* runs of one kind that fit a device batch */
        for (int r0 = 0; r0 < nw;) {
            const int fq = has_fmt(fl->path[i0 + r0], fq_fmt);
            uint64_t chunks = 0;
            int r1 = r0;
            /* a long file is a job of its own */
            const int stream0 = direct[r0] == 2 ? 2 : (direct[r0] == 1 && len[r0] >= STREAM_MIN);
            int q0 = 0; /* the device (queue) that owns input i0 + r0 */
            if (o->allpairs)
                while ((uint32_t)(i0 + r0) >= first[q0 + 1]) q0++;
            while (r1 < nw && has_fmt(fl->path[i0 + r1], fq_fmt) == fq) {
                const uint64_t c = (len[r1] + KSSD_CHUNK_BASES - 1) / KSSD_CHUNK_BASES;
                if (o->allpairs && (uint32_t)(i0 + r1) >= first[q0 + 1]) break; /* a job stays inside one device's run */
                if (r1 > r0 && (stream0 || direct[r1] == 2 || (direct[r1] == 1 && len[r1] >= STREAM_MIN))) break;
                if (r1 > r0 && chunks + c > max_chunks) break;
                chunks += c;
                r1++;
            }
            job *j = calloc(1, sizeof *j);
            j->is_fq = fq;
            j->first_file = i0 + r0;
            j->n_files = r1 - r0;
            j->q = q0;
            if (!fq || fq_dev) {
                /* FASTA, FASTQ with -Q 0: the raw bytes go to the device, which tokenises them (csrc/kssd_tok.inc) */
                if (fq) j->lines = calloc((size_t)(r1 - r0) + 1, sizeof(uint64_t));
                j->toff = calloc((size_t)(r1 - r0) + 1, sizeof(uint64_t));
                j->tlen = calloc((size_t)(r1 - r0) + 1, sizeof(uint64_t));
                uint64_t at = 0;
                for (int i = r0; i < r1; i++) {
                    j->toff[i - r0] = at;
                    j->tlen[i - r0] = len[i];
                    at += (len[i] + 15) / 16 * 16;
                }
                if (stream0) { /* the worker reads it, slice by slice, on its way to the device */
                    j->streamed = stream0;
                    n_bytes += len[r0]; /* (a gzip'ed one: its compressed size) */
                    printf("%d/%d decomposing %s\r", ++done, fl->n, fl->path[i0 + r0]);
                    goto queue_job;
                }
                /* a free text buffer -- or, while the runtime is still starting (nothing is taken off the queues yet), one more: up
                 * to TEXT_BUFS_AHEAD of them are read ahead into ordinary memory; they stay the command's buffers afterwards */
                const double tw0 = now_s();
                pthread_mutex_lock(&pl.mu);
                while (pl.n_tpool == 0 && (g_runtime_ready || pl.n_text_made >= text_ahead)) pthread_cond_wait(&pl.cv, &pl.mu);
                textbuf *tx;
                if (pl.n_tpool) {
                    tx = pl.tpool[--pl.n_tpool];
                } else {
                    tx = calloc(1, sizeof *tx);
                    pl.n_text_made++;
                }
                pthread_mutex_unlock(&pl.mu);
                if (!tx->p && !g_runtime_ready && text_ahead == 0 && text_page_locked) { /* (page-locked buffers take the runtime; ordinary ones are filled while it starts) */
                    if (warming) pthread_join(warm, NULL);
                    warm_joined = 1;
                    g_runtime_ready = 1; /* (a runtime that failed to start is reported by the workers' context creation) */
                }
                textbuf_fit(tx, at + 64, g_runtime_ready && text_page_locked);
                t_wait_text += now_s() - tw0; /* (a free buffer, the runtime's start, page-locked memory for a new one) */
                t0 = now_s();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
                for (int i = r0; i < r1; i++) {
                    const double ti0 = now_s();
                    unsigned char *dst = tx->p + j->toff[i - r0];
                    if (direct[i] == 1) {
                        size_t got = 0;
                        trc[i] = kssd_read_into(fl->path[i0 + i], dst, len[i], &got);
                        j->tlen[i - r0] = got; /* (a file that shrank meanwhile) */
                    } else {
                        memcpy(dst, txt[i], len[i]);
                    }
                    const double dti = now_s() - ti0;
#pragma omp atomic
                    t_copy_thr += dti;
                }
                t_read += now_s() - t0;
                for (int i = r0; i < r1; i++) {
                    if (trc[i]) die(EIO, "%s: %s", fl->path[i0 + i], kssd_host_strerror(trc[i]));
                    n_bytes += j->tlen[i - r0];
                    printf("%d/%d decomposing %s\r", ++done, fl->n, fl->path[i0 + i]);
                }
                j->tx = tx;
            } else {
                pthread_mutex_lock(&pl.mu);
                while (pl.n_pool == 0) pthread_cond_wait(&pl.cv, &pl.mu);
                kssd_batch *b = pl.pool[--pl.n_pool];
                pthread_mutex_unlock(&pl.mu);
                uint64_t *maxpos = malloc((size_t)(r1 - r0) * sizeof *maxpos);
                for (int i = r0; i < r1; i++) maxpos[i - r0] = len[i];
                uint32_t first = 0;
                if (kssd_batch_reserve(b, (uint32_t)(r1 - r0), maxpos, &first)) die(ENOMEM, "out of memory");
                free(maxpos);
                t0 = now_s();
                /* -A reads the bases only: no quality filter (iseq2comem.c:566-573) */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
                for (int i = r0; i < r1; i++) {
                    trc[i] = kssd_batch_fill_text(b, first + (uint32_t)(i - r0), o->abundance ? 2 : 1, txt[i], len[i], o->kmerqlty, &lines[i]);
                    if (trc[i] == KSSD_HOST_ERR_EMPTY) trc[i] = 0; /* an empty file is an empty genome here */
                }
                t_tok += now_s() - t0;
                for (int i = r0; i < r1; i++) {
                    if (trc[i]) die(EIO, "%s: %s", fl->path[i0 + i], kssd_host_strerror(trc[i]));
                    n_bytes += len[i];
                    if (!o->abundance) printf("%llu reads detected\n", (unsigned long long)lines[i]);
                    printf("%d/%d decomposing %s\r", ++done, fl->n, fl->path[i0 + i]);
                }
                j->b = b;
            }
        queue_job:
            pthread_mutex_lock(&pl.mu);
            if (pl.todo_tail[j->q]) pl.todo_tail[j->q]->next = j;
            else pl.todo_head[j->q] = j;
            pl.todo_tail[j->q] = j;
            pthread_cond_broadcast(&pl.cv);
            pthread_mutex_unlock(&pl.mu);
            n_jobs++;
            r0 = r1;
        }
    }
    for (int w = 0; w <= gz_ahead; w++) {
        for (int i = 0; i < threads; i++) free(waves[w].txt[i]);
        free(waves[w].txt);
        free(waves[w].txt_cap);
        free(waves[w].len);
        free(waves[w].trc);
        free(waves[w].direct);
    }
    free(waves);
    free(lines);
    pthread_mutex_lock(&pl.mu);
    pl.closed = 1;
    pthread_cond_broadcast(&pl.cv);
    pthread_mutex_unlock(&pl.mu);
    for (int i = 0; i < n_workers; i++) pthread_join(ws[i].th, NULL);
    if (warming && !warm_joined) pthread_join(warm, NULL);
    printf("\n");
    free(sc.accepted);
    const double t_sketched = now_s();

    /* the jobs' results in file order */
    job **jl = calloc((size_t)(n_jobs ? n_jobs : 1), sizeof *jl);
    int nj = 0;
    for (job *j = pl.done; j; j = j->next) jl[nj++] = j;
    if (nj != n_jobs) die(EIO, "stage I: %d of %d batches came back", nj, n_jobs);
    qsort(jl, (size_t)nj, sizeof *jl, cmp_job);
    uint64_t total = 0;
    for (int i = 0; i < nj; i++) total += jl[i]->off[jl[i]->n_files];
    kssd_sketchset s = {0};
    s.off = calloc((size_t)fl->n + 1, sizeof(uint64_t));
    s.ids = malloc((size_t)(total ? total : 1) * 4);
    s.counts = o->abundance ? malloc((size_t)(total ? total : 1) * 2) : NULL;
    s.sub = (nj && jl[0]->sub) ? malloc((size_t)(total ? total : 1)) : NULL; /* k - drlevel = 9 */
    if (!s.off || !s.ids || (o->abundance && !s.counts) || (nj && jl[0]->sub && !s.sub)) die(ENOMEM, "out of memory");
    uint64_t at = 0;
    for (int i = 0; i < nj; i++) {
        job *j = jl[i];
        const uint64_t m = j->off[j->n_files];
        for (int g = 0; g < j->n_files; g++) s.off[j->first_file + g + 1] = at + j->off[g + 1];
        memcpy(s.ids + at, j->ids, (size_t)m * 4);
        if (o->abundance) memcpy(s.counts + at, j->counts, (size_t)m * 2);
        if (s.sub) memcpy(s.sub + at, j->sub, (size_t)m);
        at += m;
        kssd_gpu_free(j->off);
        kssd_gpu_free(j->ids);
        free(j->counts);
        free(j->sub);
        free(j);
    }
    free(jl);
    for (int i = 0; i < pl.n_pool; i++) kssd_batch_destroy(pl.pool[i]);
    free(pl.pool);
    for (int i = 0; i < pl.n_tpool; i++) {
        textbuf_release(pl.tpool[i]);
        free(pl.tpool[i]);
    }
    free(pl.tpool);
    free(ws);
    s.shuf_id = (uint32_t)hdr.id;
    s.kmerlen = d.kmerlen;
    s.dim_rd_len = d.dim_rd_len;
    s.comp_num = d.comp_num;
    s.n = (uint32_t)fl->n;
    s.names = fl->path;
    s.koc = o->abundance;
    int rc = kssd_sketchset_write(&s, outdir, d.hashsize, 0); /* already in slot order */
    if (rc) die(EIO, "%s: %s", outdir, kssd_host_strerror(rc));
    const double t_written = now_s();
    double t_allpairs = 0, t_report = 0;
    if (o->allpairs) {
        /* what `kssd dist -r <outdir> -o <outdir> <outdir>` would do from the files just written (mco_cbdco_nobin_dist +
         * dist_print_nobin, command_dist.c:670-808,1161-1250), on the sketches the devices still hold: one all-gather, every
         * device's index, every device's own rows straight into the count matrix */
        if (d.comp_num != 1) die(ENOTSUP, "--allpairs: %d components", (int)d.comp_num);
        char skf[KSSD_PATHLEN + 32], distf[KSSD_PATHLEN + 32];
        snprintf(skf, sizeof skf, "%s/sharedk_ct.dat", outdir);
        snprintf(distf, sizeof distf, "%s/distance.out", outdir);
        const size_t cells = (size_t)s.n * s.n;
        uint32_t *shared = NULL;
        int skfd = -1;
        if (access(skf, F_OK) == 0) die(EEXIST, " mco_cbdco_nobin_dist():%s", skf); /* the reference refuses to overwrite */
        printf("disf_sz=%zu\trefnum=%u\tqrynum=%u\n", cells * 4, s.n, s.n);
        if (o->keep_skf) {
            skfd = open(skf, O_RDWR | O_CREAT | O_EXCL, 0600);
            if (skfd < 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
            snprintf(g_unlink_on_die, sizeof g_unlink_on_die, "%s", skf);
            if (ftruncate(skfd, (off_t)(cells * 4)) != 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
            if (cells) shared = mmap(NULL, cells * 4, PROT_READ | PROT_WRITE, MAP_SHARED, skfd, 0);
        } else if (cells) {
            shared = mmap(NULL, cells * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        }
        if (cells && shared == MAP_FAILED) die(errno, "mco_cbdco_nobin_dist(): %zu bytes of shared k-mer counts", cells * 4);
        {   /* the devices' sketch sizes are the files': the two were made by the same calls, one stayed behind */
            uint32_t *sz = malloc(((size_t)s.n + 1) * 4);
            for (int dv = 0; dv < n_dev; dv++) {
                gck(kssd_gpu_resident_sizes(res[dv], sz + first[dv]), "kssd_gpu_resident_sizes");
                for (uint32_t g = first[dv]; g < first[dv + 1]; g++)
                    if (sz[g] != (uint32_t)(s.off[g + 1] - s.off[g])) die(EIO, "--allpairs: genome %u: %u ids on the device, %llu in the file", g, sz[g], (unsigned long long)(s.off[g + 1] - s.off[g]));
            }
            free(sz);
        }
        if (xwarming) pthread_join(xwarm, NULL);
        gck(kssd_gpu_resident_allpairs(res, n_dev, d.kmerlen, shared, NULL, NULL, NULL, NULL), "all-pairs on the resident sketches");
        if (o->keep_skf && cells && msync(shared, cells * 4, MS_SYNC) != 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
        t_allpairs = now_s() - t_written;
        kssd_print_opt po = {o->metric, o->outfields, o->correction, o->mut_dist_max, o->num_neigb, o->p};
        rc = kssd_distance_print(distf, shared, &s, &s, &po);
        if (rc != 0)
            die(rc == KSSD_HOST_ERR_PARAM ? EINVAL : EIO, "dist_print_nobin():%s: neighborN_max %d should smaller than NREF 1024 and ref_num %u", distf, o->num_neigb, s.n);
        g_unlink_on_die[0] = 0;
        if (cells && shared) munmap(shared, cells * 4);
        if (skfd >= 0) close(skfd);
        t_report = now_s() - t_written - t_allpairs;
        for (int dv = 0; dv < n_dev; dv++) kssd_gpu_resident_destroy(res[dv]);
        if (getenv("KSSD_TIMING"))
            fprintf(stderr, "{\"kssd_timing\": \"allpairs\", \"genomes\": %u, \"gpus\": %d, \"s_exchange_index_rows\": %.6f, \"s_report\": %.6f}\n", s.n, n_dev,
                    t_allpairs, t_report);
    }
    free(s.off);
    free(s.ids);
    free(s.counts);
    free(s.sub);
    if (getenv("KSSD_TIMING")) /* machine-readable stage split (metrics / logging) */
        fprintf(stderr, "{\"kssd_timing\": \"stage1\", \"files\": %d, \"text_bytes\": %llu, \"ids\": %llu, \"batches\": %d, \"gpus\": %d, "
                        "\"host_threads\": %d, \"workers\": %d, \"s_total\": %.6f, \"s_context_create_max\": %.6f, \"s_context_destroy_max\": %.6f, \"s_before_workers\": %.6f, \"s_shuf\": %.6f, \"shuf_core_cached\": %d, \"s_read_gunzip\": %.6f, \"s_unpack_threads_summed\": %.6f, \"s_copy_threads_summed\": %.6f, \"s_wait_text_buffer\": %.6f, \"s_tokenise\": %.6f, "
                        "\"s_workers_summed\": %.6f, \"s_device_calls_summed\": %.6f, \"s_assemble_write\": %.6f}\n",
                fl->n, (unsigned long long)n_bytes, (unsigned long long)total, n_jobs, n_dev, threads, n_workers, now_s() - t_start, pl.t_ctx, pl.t_ctx_destroy, t_workers_started - t_start, t_shuf, sc.from_cache, t_read, t_unpack_thr, t_copy_thr, t_wait_text, t_tok,
                pl.t_gpu, pl.t_call, t_written - t_sketched);
}

