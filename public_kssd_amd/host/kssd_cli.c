/*
 * kssd_cli.c -- `kssd shuffle` and `kssd dist` on MI355X.
 *
 * Same sub-commands, flags, directory protocol and files as the reference front end
 * (global_wrapper.c:82-145, command_shuffle.c:33-130, command_dist_wrapper.c:41-319, dist_dispatch
 * command_dist.c:53-192); the two hot loops run on the GPU through include/kssd_gpu.h.
 * Host C; there is no CPU implementation of the hot path behind it: without a gfx950 device the
 * sketch and search modes stop with an error.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <getopt.h>
#include <pthread.h>
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/kssd_gpu.h"
#include "kssd_host.h"

#define VERSION "kssd-mi355x 0.1 (formats and results of KSSD version 1.2.21)"

static char g_unlink_on_die[KSSD_PATHLEN + 64]; /* a file this run created and has not finished (sharedk_ct.dat) */

static void die(int code, const char *fmt, ...)
{
    if (g_unlink_on_die[0]) unlink(g_unlink_on_die);
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "kssd: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    exit(code ? code : 1);
}

/* ---------------------------------------------------------------------------------------------------
 * kssd shuffle  (command_shuffle.c)
 * ------------------------------------------------------------------------------------------------- */
static int cmd_shuffle(int argc, char **argv)
{
    int k = 8, subk = 5, lvl = 2; /* defaults of dim_shuffle_stat, command_shuffle.c:48-53 */
    unsigned long long seed = 0;
    char prefix[KSSD_PATHLEN] = "./default";
    static struct option lo[] = {{"halfKmerLen", 1, 0, 'k'}, {"halfSubstrLen", 1, 0, 's'}, {"level", 1, 0, 'l'},
                                 {"outfile", 1, 0, 'o'},     {"usedefault", 0, 0, 999},    {"seed", 1, 0, 998},
                                 {0, 0, 0, 0}};
    if (argc < 2) die(EINVAL, "usage: kssd shuffle -k <halfKmerLen> -s <halfSubstrLen> -l <level> -o <prefix> [--seed N]");
    int c;
    while ((c = getopt_long(argc, argv, "k:s:l:o:", lo, NULL)) != -1) {
        switch (c) {
        case 'k': k = atoi(optarg); break;
        case 's': subk = atoi(optarg); break;
        case 'l': lvl = atoi(optarg); break;
        case 'o':
            if (strlen(optarg) + strlen(".shuf") >= KSSD_PATHLEN) die(ENAMETOOLONG, "output path name %s should less than %d characters", optarg, KSSD_PATHLEN);
            strcpy(prefix, optarg);
            break;
        case 999: printf("use default values for all options\n"); break;
        case 998: seed = strtoull(optarg, NULL, 10); break; /* extension: reproducible .shuf */
        default: die(EINVAL, "shuffle: unknown option");
        }
    }
    if (k < subk) die(EINVAL, "write_dim_shuffle_file(): half-context len: %d should larger than half-subcontext len (or dimension reduce level + 2) %d", k, subk);
    if (subk >= 8) die(EINVAL, "write_dim_shuffle_file(): subk shoud smaller than 8");
    if ((1 << 4 * (subk - lvl > 0 ? subk - lvl : 0)) < 4096)
        fprintf(stderr, "kssd: dimension after reduction %d is smaller than the suggested minimal dimension sample size %d, "
                        "which might cause loss of robutness, -s %d is suggested\n", 1 << 4 * (subk - lvl > 0 ? subk - lvl : 0), 4096, lvl + 3);
    kssd_shuf s;
    int rc = kssd_shuf_generate(&s, k, subk, lvl, seed);
    if (rc) die(EINVAL, "shuffle: %s", kssd_host_strerror(rc));
    char out[KSSD_PATHLEN + 8];
    snprintf(out, sizeof out, "%s.shuf", prefix);
    if ((rc = kssd_shuf_write(&s, out)) != 0) die(EIO, "write_dim_shuffle_file(): open file %s failed", out);
    printf("kssd shuffle: shuf_id=%d, k = %d, halfCtxLen = %d, level= %d\n", s.id, s.k, s.subk, s.drlevel);
    kssd_shuf_release(&s);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * input discovery (organize_infile_frm_arg / organize_infile_list, global_basic.c:143-283)
 * ------------------------------------------------------------------------------------------------- */
static const char *acpt[] = {"fna", "fas", "fasta", "fq", "fastq", "fa", "co", NULL};
static const char *fq_fmt[] = {"fq", "fastq", NULL};

static int has_fmt(const char *name, const char **fmts)
{ /* isOK_fmt_infile, global_basic.h:129-150: suffix test after stripping .gz / .bz2 */
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s", name);
    size_t L = strlen(tmp);
    if (L > 3 && strcmp(tmp + L - 3, ".gz") == 0) tmp[L - 3] = 0;
    else if (L > 4 && strcmp(tmp + L - 4, ".bz2") == 0) tmp[L - 4] = 0;
    L = strlen(tmp);
    for (int i = 0; fmts[i]; i++) {
        size_t fl = strlen(fmts[i]);
        if (L > fl + 1 && tmp[L - fl - 1] == '.' && strcmp(tmp + L - fl, fmts[i]) == 0) return 1;
    }
    return 0;
}

typedef struct {
    char (*path)[KSSD_PATHLEN];
    int n, cap;
} filelist;

static void fl_add(filelist *f, const char *p)
{
    if (strlen(p) >= KSSD_PATHLEN) die(ENAMETOOLONG, "path: %s exceed maximal path lenth %d", p, KSSD_PATHLEN);
    if (f->n == f->cap) {
        f->cap = f->cap ? f->cap * 2 : 1024;
        f->path = realloc(f->path, (size_t)f->cap * KSSD_PATHLEN);
    }
    memset(f->path[f->n], 0, KSSD_PATHLEN);
    strcpy(f->path[f->n++], p);
}

static int cmp_path(const void *a, const void *b) { return strcmp((const char *)a, (const char *)b); }

static void collect_inputs(filelist *f, int nargs, char **args, const char *listfile)
{
    char full[4096];
    if (listfile && listfile[0]) {
        FILE *l = fopen(listfile, "r");
        if (!l) die(errno, "can't open file %s", listfile);
        char line[4096];
        while (fgets(line, sizeof line, l)) {
            char *p = line;
            while (*p == ' ' || *p == '\t') p++;
            p[strcspn(p, "\r\n")] = 0;
            if (!*p) continue;
            struct stat st;
            if (stat(p, &st) != 0 || !S_ISREG(st.st_mode)) die(ENOENT, "%dth line: %s", f->n, p);
            if (!has_fmt(p, acpt)) die(EINVAL, "isOK_fmt_infile(): wrong format %dth line: %s", f->n, p);
            fl_add(f, p);
        }
        fclose(l);
        return;
    }
    for (int i = 0; i < nargs; i++) {
        struct stat st;
        if (stat(args[i], &st) != 0) die(errno, "%dth argument: can't open %s", i + 1, args[i]);
        if (S_ISDIR(st.st_mode)) {
            DIR *d = opendir(args[i]);
            if (!d) die(errno, "%dth argument: can't open %s", i + 1, args[i]);
            int first = f->n;
            struct dirent *e;
            while ((e = readdir(d)) != NULL) {
                snprintf(full, sizeof full, "%s/%s", args[i], e->d_name);
                if (has_fmt(full, acpt)) fl_add(f, full);
            }
            closedir(d);
            /* the reference shuffles the order with a time seed (command_dist.c:168); any order is valid,
             * we keep it reproducible */
            qsort(f->path + first, (size_t)(f->n - first), KSSD_PATHLEN, cmp_path);
        } else if (has_fmt(args[i], acpt)) {
            fl_add(f, args[i]);
        } else {
            die(EINVAL, "wrong format %dth argument: %s\nSupported format are: .fna .fas .fasta .fq .fastq .fa .co", i + 1, args[i]);
        }
    }
}

/* default -p: the processors of the machine like the reference (omp_get_num_procs, command_dist_wrapper.c:284-287), but
 * not more than the CPU time the container is allowed (cgroup cpu.max / cfs quota): 256 threads on a 16-CPU quota
 * only take turns */
static int default_threads(void)
{
    int n = 1;
#ifdef _OPENMP
    n = omp_get_num_procs();
#endif
    long long quota = -1, period = 100000;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[64];
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL) {
        if (fscanf(f, "%lld", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) != NULL) {
            if (fscanf(f, "%lld", &period) != 1) period = 100000;
            fclose(f);
        }
    }
    if (quota > 0 && period > 0) {
        long long c = (quota + period - 1) / period;
        if (c >= 1 && c < n) n = (int)c;
    }
    return n < 1 ? 1 : n;
}

/* ---------------------------------------------------------------------------------------------------
 * dist options (command_dist_wrapper.c:41-100)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int k, p, dr_level, kmerocrs, kmerqlty, num_neigb, metric, outfields, correction, u, keep_skf, abundance, byread;
    double mut_dist_max;
    char dr_file[KSSD_PATHLEN], refpath[KSSD_PATHLEN], fpath[KSSD_PATHLEN], outdir[KSSD_PATHLEN], skf[KSSD_PATHLEN],
        pipecmd[KSSD_PATHLEN];
    int nargs;
    char **args;
    int device, gpus; /* first device, number of devices (--gpus / KSSD_GPUS) */
    int fake_ranks;   /* KSSD_EXCHANGE_FAKE_RANKS: the entries of devs[] are ranks that share one device (development) */
    int allpairs;     /* --allpairs: stage I, then all-pairs among the inputs in the same run, sketches resident on the devices */
    int devs[64], n_devs; /* the device list: device .. device + gpus - 1, or KSSD_DEVICE_LIST=a,b,c */
    unsigned long long seed;
} dist_opt;

static kssd_gpu_ctx *g_ctx;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}


static void gck(int rc, const char *what)
{
    if (rc != KSSD_OK) die(rc == KSSD_ERR_CAPACITY ? ENOSPC : rc == KSSD_ERR_NO_DEVICE ? ENODEV : EIO, "%s: %s", what, kssd_gpu_strerror(rc));
}

/* the .shuf of this run: -L <file>, or a fresh default.shuf in the output directory (get_dim_shuffle, command_dist.c:193-216).
 * What comes back is the header and the accepted sub-contexts -- all the sketch path looks at (kssd_shuf_read_core: from the
 * .core file beside the .shuf when there is one, so that a run does not start by reading 64 MiB or 1 GiB for 16 KiB of it) */
typedef struct {
    kssd_shuf h; /* table == NULL */
    uint32_t *accepted;
    uint32_t n_accepted;
    int from_cache;
} shuf_core;

static void load_shuf(const dist_opt *o, shuf_core *s)
{
    char path[KSSD_PATHLEN + 16];
    if (o->dr_file[0]) {
        snprintf(path, sizeof path, "%s", o->dr_file);
    } else {
        int subk = o->dr_level + 3; /* add_len_drlevel2subk() == 3, command_shuffle.c:154-160 */
        if (o->k < subk) die(EINVAL, "write_dim_shuffle_file(): half-context len: %d should larger than half-subcontext len (or dimension reduce level + 2) %d", o->k, subk);
        if (subk >= 8) die(EINVAL, "write_dim_shuffle_file(): subk shoud smaller than 8");
        kssd_shuf g;
        int rc = kssd_shuf_generate(&g, o->k, subk, o->dr_level, o->seed);
        if (rc) die(EINVAL, "shuffle: %s", kssd_host_strerror(rc));
        mkdir(o->outdir, 0777);
        snprintf(path, sizeof path, "%s/default.shuf", o->outdir);
        if (kssd_shuf_write(&g, path)) die(EIO, "write_dim_shuffle_file(): open file %s failed", path);
        printf("kssd shuffle: shuf_id=%d, k = %d, halfCtxLen = %d, level= %d\n", g.id, g.k, g.subk, g.drlevel);
        kssd_shuf_release(&g);
    }
    int rc = kssd_shuf_read_core(path, &s->h, &s->accepted, &s->n_accepted, &s->from_cache);
    if (rc) die(EIO, "read_dim_shuffle_file(): %s: %s", path, kssd_host_strerror(rc));
}

/* HIP initialisation (driver, device context, code object) costs 0.1 - 0.2 s per process: started on a thread of its own
 * at the top of a command, it runs while the command reads its .shuf or its sketch directories */
/* --allpairs over several devices: librccl.so and the communicators come up while stage I runs (kssd_gpu_exchange_warm_up) */
typedef struct {
    const int *devs;
    int n;
} xwarm_arg;
static void *warm_exchange(void *arg)
{
    const xwarm_arg *x = arg;
    kssd_gpu_exchange_warm_up(x->devs, x->n); /* (a failure is reported by the exchange itself) */
    return NULL;
}

/* set once the HIP runtime and a device's context are up (warm_device, or the first worker's context): from then on text buffers
 * are page-locked */
static volatile int g_runtime_ready;

static void *warm_device(void *arg)
{
    if (kssd_gpu_warm_up(*(const int *)arg) == KSSD_OK) g_runtime_ready = 1; /* (a missing device is reported by the call that needs it) */
    return NULL;
}

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

static void sketch_files(const dist_opt *o, filelist *fl, const char *outdir)
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
    int wpd = 2, extra_bufs = 1;
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
    uint64_t n_bytes = 0;
    int done = 0, n_jobs = 0;
    /* the text buffers of a wave's files are kept from wave to wave (no fresh memory per file) */
    unsigned char **txt = calloc((size_t)threads, sizeof *txt);
    size_t *txt_cap = calloc((size_t)threads, sizeof *txt_cap);
    size_t *len = calloc((size_t)threads, sizeof *len);
    int *trc = calloc((size_t)threads, sizeof *trc);
    uint64_t *lines = calloc((size_t)threads, sizeof *lines);
    int *direct = calloc((size_t)threads, sizeof *direct); /* plain file for the device tokeniser: read straight into the job's text buffer */
    /* FASTQ text is tokenised on the device: fastq2co's framing with its quality rule (-Q) or, under -A, the framing of
     * mt_shortreads2koc (kssd_gpu_set_fastq_quality / _reads in worker_main); inputs only the reference's own fgets()
     * sequence reproduces come back (KSSD_ERR_UNSUPPORTED) and go through the host tokeniser */
    const int fq_dev = (o->abundance || (o->kmerqlty >= 0 && o->kmerqlty <= 127)) && !getenv("KSSD_HOST_FASTQ");
    stream_env();
    for (int i0 = 0; i0 < fl->n; i0 += threads) {
        const int i1 = i0 + threads < fl->n ? i0 + threads : fl->n, nw = i1 - i0;
        double t0 = now_s();
        /* read + gunzip, one file each.  gzip'ed files (and FASTQ files the host tokenises) go into the wave's scratch
         * buffers; a plain file the device tokenises is only measured here and read straight into page-locked memory below */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int i = 0; i < nw; i++) {
            const char *path = fl->path[i0 + i];
            int gz = 0;
            uint64_t sz = 0;
            direct[i] = 0;
            trc[i] = kssd_file_probe(path, &gz, &sz);
            if (trc[i]) continue;
            const int dev_tok = fq_dev || !has_fmt(path, fq_fmt); /* the device tokenises it */
            if (!gz && dev_tok) {
                direct[i] = 1;
                len[i] = (size_t)sz;
            } else if (gz && dev_tok && sz >= STREAM_MIN_GZ) {
                direct[i] = 2; /* inflated by the device worker, slice by slice, on its way to the device */
                len[i] = (size_t)sz;
            } else {
                trc[i] = kssd_slurp_reuse(path, &txt[i], &txt_cap[i], &len[i]);
            }
        }
        t_read += now_s() - t0;
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
                /* a free text buffer -- or, while the devices lag behind the readers (the runtime is still starting: nothing is
                 * taken off the queues yet), one more: the inputs are read ahead into memory up to the budget */
                /* a free text buffer -- or, while the runtime is still starting (nothing is taken off the queues yet), one more: up
                 * to TEXT_BUFS_AHEAD of them are read ahead into ordinary memory; they stay the command's buffers afterwards */
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
                if (!tx->p && !g_runtime_ready && text_ahead == 0) { /* (no reading ahead: the buffers are page-locked ones, which takes the runtime) */
                    if (warming) pthread_join(warm, NULL);
                    warm_joined = 1;
                    g_runtime_ready = 1; /* (a runtime that failed to start is reported by the workers' context creation) */
                }
                textbuf_fit(tx, at + 64, g_runtime_ready);
                t0 = now_s();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
                for (int i = r0; i < r1; i++) {
                    unsigned char *dst = tx->p + j->toff[i - r0];
                    if (direct[i] == 1) {
                        size_t got = 0;
                        trc[i] = kssd_read_into(fl->path[i0 + i], dst, len[i], &got);
                        j->tlen[i - r0] = got; /* (a file that shrank meanwhile) */
                    } else {
                        memcpy(dst, txt[i], len[i]);
                    }
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
    free(direct);
    for (int i = 0; i < threads; i++) free(txt[i]);
    free(txt);
    free(txt_cap);
    free(len);
    free(trc);
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
                        "\"host_threads\": %d, \"s_total\": %.6f, \"s_context_create_max\": %.6f, \"s_context_destroy_max\": %.6f, \"s_before_workers\": %.6f, \"s_shuf\": %.6f, \"shuf_core_cached\": %d, \"s_read_gunzip\": %.6f, \"s_tokenise\": %.6f, "
                        "\"s_workers_summed\": %.6f, \"s_device_calls_summed\": %.6f, \"s_assemble_write\": %.6f}\n",
                fl->n, (unsigned long long)n_bytes, (unsigned long long)total, n_jobs, n_dev, threads, now_s() - t_start, pl.t_ctx, pl.t_ctx_destroy, t_workers_started - t_start, t_shuf, sc.from_cache, t_read, t_tok,
                pl.t_gpu, pl.t_call, t_written - t_sketched);
}

/* stage II (run_stageII, command_dist.c:381-417): the index FILES are for the reference binary; our own search
 * builds its index on the device straight from the sketches */
static void build_index_files(const char *codir, const char *mcodir)
{
    kssd_sketchset s;
    int rc = kssd_sketchset_read(&s, codir);
    if (rc) die(EIO, "run_stageII(): %s: %s", codir, kssd_host_strerror(rc));
    struct stat st;
    if (stat(mcodir, &st) == 0) printf("Warning: write mco file to an exists outdir:%s\n", mcodir);
    if ((rc = kssd_index_write(&s, mcodir)) != 0) die(EIO, "combco2mco(): %s: %s", mcodir, kssd_host_strerror(rc));
    kssd_sketchset_release(&s);
}

/* search (mco_cbdco_nobin_dist + dist_print_nobin, command_dist.c:670-808,1161-1250) */
static void search(const dist_opt *o, const char *refdir, const char *qrydir)
{
    const double t_start = now_s();
    pthread_t warm;
    int warm_dev = o->device;
    const int warming = !o->skf[0] && pthread_create(&warm, NULL, warm_device, &warm_dev) == 0;
    kssd_sketchset ref, qry;
    int rc;
    if (kssd_probe_dir(refdir) & 1) rc = kssd_sketchset_read(&ref, refdir);
    else rc = kssd_index_read(&ref, refdir);
    if (rc) die(EIO, "need provied mco dir path: %s: %s", refdir, kssd_host_strerror(rc));
    if ((rc = kssd_sketchset_read(&qry, qrydir)) != 0) die(EIO, "need provied co dir path: %s: %s", qrydir, kssd_host_strerror(rc));
    if (ref.comp_num != qry.comp_num)
        die(EINVAL, "query args not match ref args: ref.comp_num = %d vs. %d = qry.comp_num", ref.comp_num, qry.comp_num);
    if (ref.shuf_id != qry.shuf_id)
        die(EINVAL, "query args not match ref args: ref.shuf_id = %d vs. %d = qry.shuf_id", (int)ref.shuf_id, (int)qry.shuf_id);
    const double t_read = now_s();
    mkdir(o->outdir, 0700);
    char skf[KSSD_PATHLEN + 32], distf[KSSD_PATHLEN + 32];
    snprintf(skf, sizeof skf, "%s/sharedk_ct.dat", o->outdir);
    snprintf(distf, sizeof distf, "%s/distance.out", o->outdir);
    const size_t cells = (size_t)ref.n * qry.n;
    /* With --keepskf the count matrix lives in the file itself, mapped like the reference maps it (command_dist.c:741-748):
     * Q x R may exceed the host's memory, the device works it off in row tiles (kssd_gpu_dist).  Without it the reference
     * removes the file when the report is written (:1249), so it is never created here: the counts stay in anonymous
     * memory.  A selecting report (-N / -D) that does not keep the file needs no dense matrix on the host at all. */
    uint32_t *shared = NULL;
    uint64_t *pair_off = NULL;
    uint32_t *pair_ref = NULL, *pair_shared = NULL;
    int skfd = -1;
    const int selecting = !o->skf[0] && (o->num_neigb > 0 || o->mut_dist_max < 1.0);
    double t_ctx = 0, t_dist = 0;
    if (o->skf[0]) { /* -f: reuse a kept shared-k-mer file (command_dist.c:735-738) */
        skfd = open(o->skf, O_RDONLY);
        struct stat st;
        if (skfd < 0 || fstat(skfd, &st) != 0 || (size_t)st.st_size != cells * 4) die(EIO, "open %s failed", o->skf);
        if (cells) shared = mmap(NULL, cells * 4, PROT_READ, MAP_PRIVATE, skfd, 0);
        if (cells && shared == MAP_FAILED) die(errno, "mmap %s", o->skf);
    } else {
        if (access(skf, F_OK) == 0) die(EEXIST, " mco_cbdco_nobin_dist():%s", skf); /* the reference refuses to overwrite */
        printf("disf_sz=%zu\trefnum=%u\tqrynum=%u\n", cells * 4, ref.n, qry.n);
        if (o->keep_skf) {
            skfd = open(skf, O_RDWR | O_CREAT | O_EXCL, 0600);
            if (skfd < 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
            snprintf(g_unlink_on_die, sizeof g_unlink_on_die, "%s", skf); /* a failed run must not leave a zero-filled file behind */
            if (ftruncate(skfd, (off_t)(cells * 4)) != 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
            if (cells) shared = mmap(NULL, cells * 4, PROT_READ | PROT_WRITE, MAP_SHARED, skfd, 0);
            if (cells && shared == MAP_FAILED) die(errno, "mmap %s", skf);
        } else if (!selecting && cells) {
            shared = mmap(NULL, cells * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (shared == MAP_FAILED) die(errno, "mco_cbdco_nobin_dist(): %zu bytes of shared k-mer counts", cells * 4);
        }
        if (warming) pthread_join(warm, NULL);
        const int n_dev = o->n_devs;
        const int have = kssd_gpu_device_count();
        if (have <= 0) die(ENODEV, "kssd_gpu_create_for_dist: %s", kssd_gpu_strerror(KSSD_ERR_NO_DEVICE));
        for (int i = 0; i < n_dev; i++)
            if (o->devs[i] >= have) die(ENODEV, "device %d of the list: only %d device(s) visible", o->devs[i], have);
        const double t0 = now_s();
        if (selecting) {
            /* -N / -D leave few lines: let the device pick the pairs that can be printed (output_ctrl's rules with a
             * margin), the host ranks / tests those exactly and formats only them */
            if (o->num_neigb > 1024 || (uint32_t)o->num_neigb > ref.n)
                die(EINVAL, "dist_print_nobin():%s: neighborN_max %d should smaller than NREF 1024 and ref_num %u", distf, o->num_neigb, ref.n);
            gck(kssd_gpu_create_for_dist(&g_ctx, qry.kmerlen, o->device), "kssd_gpu_create_for_dist");
            t_ctx = now_s() - t0;
            gck(kssd_gpu_dist_select(g_ctx, ref.off, ref.ids, ref.n, qry.off, qry.ids, qry.n, o->metric, o->correction, qry.dim_rd_len,
                                     o->mut_dist_max, o->num_neigb, shared, &pair_off, &pair_ref, &pair_shared), "dist");
            kssd_gpu_destroy(g_ctx);
            g_ctx = NULL;
        } else {
            /* query rows in contiguous blocks, one per device; every device indexes all references (command_dist.c:774-785) */
            gck(kssd_gpu_dist_multi(o->devs, n_dev, qry.kmerlen, ref.off, ref.ids, ref.n, qry.off, qry.ids, qry.n, shared, NULL, NULL, NULL, NULL), "dist");
        }
        t_dist = now_s() - t0;
        if (o->keep_skf && cells && msync(shared, cells * 4, MS_SYNC) != 0) die(errno, "mco_cbdco_nobin_dist()::%s", skf);
    }
    const double t_print0 = now_s();
    kssd_print_opt po = {o->metric, o->outfields, o->correction, o->mut_dist_max, o->num_neigb, o->p};
    rc = pair_off ? kssd_distance_print_pairs(distf, pair_off, pair_ref, pair_shared, &ref, &qry, &po)
                  : kssd_distance_print(distf, shared, &ref, &qry, &po);
    kssd_gpu_free(pair_off);
    kssd_gpu_free(pair_ref);
    kssd_gpu_free(pair_shared);
    if (rc != 0)
        die(rc == KSSD_HOST_ERR_PARAM ? EINVAL : EIO, "dist_print_nobin():%s: neighborN_max %d should smaller than NREF 1024 and ref_num %u", distf, o->num_neigb, ref.n);
    g_unlink_on_die[0] = 0; /* the run is through: a kept count file stays */
    if (cells && shared) munmap(shared, cells * 4);
    if (skfd >= 0) close(skfd);
    if (getenv("KSSD_TIMING"))
        fprintf(stderr, "{\"kssd_timing\": \"search\", \"refs\": %u, \"queries\": %u, \"host_threads\": %d, \"s_total\": %.6f, \"s_read_sketches\": %.6f, "
                        "\"s_device\": %.6f, \"s_context_create\": %.6f, \"s_report\": %.6f}\n",
                ref.n, qry.n, o->p, now_s() - t_start, t_read - t_start, t_dist, t_ctx, now_s() - t_print0);
    kssd_sketchset_release(&ref);
    kssd_sketchset_release(&qry);
}

/* several sketch directories -> one (combine_queries, command_dist.c:1323-1475): the first directory sets the shuffle;
 * later ones that are not sketch directories, carry another shuf_id or the abundance flag are skipped with the
 * reference's messages; genomes keep their order, ids keep the order they have in the files */
static void combine_queries(const dist_opt *o)
{
    if (o->abundance) die(EINVAL, "combine_queries(): abundance model not supported yet");
    kssd_sketchset all;
    memset(&all, 0, sizeof all);
    int rc = kssd_sketchset_read(&all, o->args[0]);
    if (rc) die(EIO, "combine_queries():%s/cofiles.stat: %s", o->args[0], kssd_host_strerror(rc));
    if (all.koc) die(EINVAL, "combine_queries(): abundance model not supported yet");
    for (int i = 1; i < o->nargs; i++) {
        if (!(kssd_probe_dir(o->args[i]) & 1)) {
            printf("%dth query %s is not a valid query: no cofiles.stat file\n", i, o->args[i]);
            continue;
        }
        kssd_sketchset it;
        memset(&it, 0, sizeof it);
        rc = kssd_sketchset_read(&it, o->args[i]);
        if (rc) {
            printf("combine_queries(): %dth query can not open %s/cofiles.stat\n", i, o->args[i]);
            continue;
        }
        if (it.shuf_id != all.shuf_id) {
            printf("combine_queries(): %dth shuf_id: %u not match 0th shuf_id: %u\n", i, it.shuf_id, all.shuf_id);
            kssd_sketchset_release(&it);
            continue;
        }
        if (it.koc) {
            printf("combine_queries(): %dth query abundance model not supported yet \n", i);
            kssd_sketchset_release(&it);
            continue;
        }
        const uint64_t a = all.off[all.n], b = it.off[it.n];
        all.off = realloc(all.off, ((size_t)all.n + it.n + 1) * sizeof(uint64_t));
        all.ids = realloc(all.ids, (size_t)(a + b ? a + b : 1) * 4);
        all.names = realloc(all.names, ((size_t)all.n + it.n) * sizeof *all.names);
        if (!all.off || !all.ids || !all.names) die(ENOMEM, "out of memory");
        for (uint32_t g = 0; g < it.n; g++) all.off[all.n + g + 1] = a + it.off[g + 1];
        memcpy(all.ids + a, it.ids, (size_t)b * 4);
        memcpy(all.names + all.n, it.names, (size_t)it.n * sizeof *all.names);
        all.n += it.n;
        kssd_sketchset_release(&it);
    }
    mkdir(o->outdir, 0700);
    rc = kssd_sketchset_write(&all, o->outdir, 0, 0);
    if (rc) die(EIO, "%s: %s", o->outdir, kssd_host_strerror(rc));
    kssd_sketchset_release(&all);
}

static int cmd_dist(int argc, char **argv)
{
    dist_opt o;
    memset(&o, 0, sizeof o);
    o.k = 8; o.dr_level = 2; o.kmerocrs = 1; o.mut_dist_max = 1; o.outfields = 2;
    strcpy(o.outdir, ".");
    o.device = getenv("KSSD_DEVICE") ? atoi(getenv("KSSD_DEVICE")) : 0;
    o.gpus = getenv("KSSD_GPUS") ? atoi(getenv("KSSD_GPUS")) : 1; /* extension: devices device .. device+gpus-1 */
    static struct option lo[] = {
        {"halfKmerlength", 1, 0, 'k'}, {"threadN", 1, 0, 'p'},      {"list", 1, 0, 'l'},       {"DimRdcLevel", 1, 0, 'L'},
        {"maxMemory", 1, 0, 'm'},      {"LstKmerOcrs", 1, 0, 'n'},  {"quality", 1, 0, 'Q'},    {"reference_dir", 1, 0, 'r'},
        {"outdir", 1, 0, 'o'},         {"neighborN_max", 1, 0, 'N'}, {"mutDist_max", 1, 0, 'D'}, {"metric", 1, 0, 'M'},
        {"outfields", 1, 0, 'O'},      {"correction", 1, 0, 333},   {"abundance", 0, 0, 'A'},  {"dedup", 0, 0, 'u'},
        {"keepcofile", 0, 0, 888},     {"pipecmd", 1, 0, 'P'},      {"keepskf", 0, 0, 777},    {"skf", 1, 0, 'f'},
        {"byread", 0, 0, 555},         {"seed", 1, 0, 998},         {"gpus", 1, 0, 997},       {"allpairs", 0, 0, 996},
        {0, 0, 0, 0}};
    if (argc < 2) die(EINVAL, "usage: kssd dist [-L <.shuf|level>] [-k K] [-r <reference>] [-o <outdir>] [<query> ...]");
    int c;
    while ((c = getopt_long(argc, argv, "k:p:l:L:m:n:Q:r:o:N:D:M:O:AuP:f:", lo, NULL)) != -1) {
        switch (c) {
        case 'k': o.k = atoi(optarg); break;
        case 'p': o.p = atoi(optarg); break;
        case 'l': snprintf(o.fpath, sizeof o.fpath, "%s", optarg); break;
        case 'L': {
            struct stat st;
            if (stat(optarg, &st) == 0 && S_ISREG(st.st_mode)) snprintf(o.dr_file, sizeof o.dr_file, "%s", optarg);
            else {
                if (atoi(optarg) >= o.k - 2 || atoi(optarg) < 0)
                    die(EINVAL, "-L: dimension reduction level should never larger than Kmer length - 2, which is %d here", o.k - 2);
                o.dr_level = atoi(optarg);
            }
            break;
        }
        case 'm': break; /* memory budgeting of the CPU hash tables: nothing to budget here */
        case 'n': o.kmerocrs = atoi(optarg) > 7 ? 7 : atoi(optarg) < 1 ? 1 : atoi(optarg); break;
        case 'Q': o.kmerqlty = atoi(optarg); break;
        case 'r': snprintf(o.refpath, sizeof o.refpath, "%s", optarg); break;
        case 'o': snprintf(o.outdir, sizeof o.outdir, "%s", optarg); break;
        case 'N': o.num_neigb = atoi(optarg); break;
        case 'D': o.mut_dist_max = atof(optarg); break;
        case 'M': o.metric = atoi(optarg) ? 1 : 0; break;
        case 'O': o.outfields = atoi(optarg) < 0 ? 0 : atoi(optarg) > 2 ? 2 : atoi(optarg); break;
        case 333: o.correction = atoi(optarg); break;
        case 'A': o.abundance = 1; break;
        case 'u': o.u = 1; break;
        case 888: break;
        case 'P': snprintf(o.pipecmd, sizeof o.pipecmd, "%s", optarg); break;
        case 777: o.keep_skf = 1; break;
        case 'f': snprintf(o.skf, sizeof o.skf, "%s", optarg); break;
        case 555: o.byread = 1; break;
        case 998: o.seed = strtoull(optarg, NULL, 10); break;
        case 997: o.gpus = atoi(optarg); break;
        case 996: o.allpairs = 1; break; /* extension: stage I + all-pairs in one run, sketches resident on the devices */
        default: die(EINVAL, "dist: unknown option");
        }
    }
    if (o.gpus < 1 || o.gpus > 64) die(EINVAL, "--gpus: between 1 and 64 devices");
    o.n_devs = o.gpus;
    for (int i = 0; i < o.gpus; i++) o.devs[i] = o.device + i;
    if (getenv("KSSD_DEVICE_LIST")) { /* an explicit list, e.g. 2,3,6 (development: 0,0 must be refused by --allpairs) */
        o.n_devs = 0;
        for (const char *p = getenv("KSSD_DEVICE_LIST"); *p;) { /* whole non-negative numbers between commas, at most 64 */
            char *end = NULL;
            errno = 0;
            const long v = strtol(p, &end, 10);
            if (end == p || errno || v < 0 || v > 1 << 20 || (*end && *end != ',')) die(EINVAL, "KSSD_DEVICE_LIST: '%s' is not a list of device numbers", getenv("KSSD_DEVICE_LIST"));
            if (o.n_devs >= 64) die(EINVAL, "KSSD_DEVICE_LIST: more than 64 devices");
            o.devs[o.n_devs++] = (int)v;
            p = *end == ',' ? end + 1 : end;
            if (*end == ',' && !*p) die(EINVAL, "KSSD_DEVICE_LIST: '%s' ends in a comma", getenv("KSSD_DEVICE_LIST"));
        }
        if (o.n_devs < 1) die(EINVAL, "KSSD_DEVICE_LIST: no device");
        o.gpus = o.n_devs;
        o.device = o.devs[0];
    }
    if (o.allpairs && getenv("KSSD_EXCHANGE_FAKE_RANKS")) {
        /* development: n "ranks" on the FIRST device of the list -- the whole N-rank orchestration of --allpairs (plan, residents, one
         * host thread per rank, unit padding, the ranks' rows into one mapping) with the collective replaced by device-to-device
         * copies (kssd_gpu_resident_allpairs): what a one-GPU box can execute of --gpus n */
        const long n = strtol(getenv("KSSD_EXCHANGE_FAKE_RANKS"), NULL, 10);
        if (n < 1 || n > 64) die(EINVAL, "KSSD_EXCHANGE_FAKE_RANKS: between 1 and 64 ranks");
        o.n_devs = o.gpus = (int)n;
        for (int i = 0; i < o.n_devs; i++) o.devs[i] = o.device;
        o.fake_ranks = 1;
    }
    if (o.allpairs && (o.byread || o.pipecmd[0])) die(ENOTSUP, "--allpairs with --byread / --pipecmd");
    o.nargs = argc - optind;
    o.args = argv + optind;
    if (o.allpairs) {
        /* --allpairs is stage I + the search in one run: it needs raw sequences as its inputs and no -r (dist_dispatch,
         * command_dist.c:159-189, is the branch it extends).  Said here, before anything is sketched or overwritten. */
        if (o.refpath[0]) die(EINVAL, "--allpairs: all-pairs among the inputs of this run; -r <reference> names another search (sketch first, then kssd dist -r)");
        if (o.nargs > 0 && !o.pipecmd[0] && kssd_probe_dir(o.args[0])) die(EINVAL, "--allpairs: %s holds sketches; all-pairs among sketches is kssd dist -r %s -o <out> %s", o.args[0], o.args[0], o.args[0]);
        if (o.nargs < 1 && !o.fpath[0]) die(EINVAL, "--allpairs: no input sequences");
        char skf[KSSD_PATHLEN + 32];
        snprintf(skf, sizeof skf, "%s/sharedk_ct.dat", o.outdir);
        if (access(skf, F_OK) == 0) die(EEXIST, " mco_cbdco_nobin_dist():%s", skf); /* the reference refuses to overwrite (command_dist.c:707-748): before stage I runs */
    }
    if (o.p == 0) o.p = default_threads();

    /* dist_dispatch (command_dist.c:53-192) */
    if (o.refpath[0]) {
        int probe = kssd_probe_dir(o.refpath);
        if (probe == 0) { /* raw sequences as reference: sketch and index them into -o */
            filelist fl = {0};
            struct stat st;
            if (stat(o.refpath, &st) != 0) die(errno, "dist_organize_refpath():%s", o.refpath);
            char *one[1] = {o.refpath};
            if (S_ISDIR(st.st_mode) || has_fmt(o.refpath, acpt)) collect_inputs(&fl, 1, one, NULL);
            else collect_inputs(&fl, 0, NULL, o.refpath);
            sketch_files(&o, &fl, o.outdir);
            build_index_files(o.outdir, o.outdir);
        }
        /* probe == 1 (sketches without mco.*): the reference would run stage II here (command_dist.c:100-103) only
         * because its search needs the 2 GiB offset file; the device index is built from the sketches, so nothing
         * has to be written.  `kssd dist -o <dir> <dir>` still writes mco.* for the reference binary. */
    }
    if (o.nargs > 0 || o.fpath[0]) {
        int qprobe = (o.nargs > 0 && !o.pipecmd[0]) ? kssd_probe_dir(o.args[0]) : 0;
        if (o.refpath[0]) {
            if (!kssd_probe_dir(o.refpath)) die(EINVAL, "need speficy the ref-sketch path for -r to run the query-ref search model");
            if (qprobe & 1) search(&o, o.refpath, o.args[0]);
            else if (qprobe & 2) die(EINVAL, "when -r specified, the query sould not be .mco format, the valid query format shoulde be .fas/.fq file or .co");
            else die(EINVAL, "please sketch the query sequences first (kssd dist -L <.shuf> -o <qrydir> <seqs>), then search with -r");
        } else if (qprobe & 1) {
            if (o.nargs == 1) build_index_files(o.args[0], o.outdir);
            else combine_queries(&o);
        } else {
            filelist fl = {0};
            collect_inputs(&fl, o.nargs, o.args, o.fpath);
            sketch_files(&o, &fl, o.outdir);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * kssd set (command_set.c): union / uniq union of a sketch directory into a pan-sketch, subtraction of or
 * intersection with a pan-sketch, genome names.  The 2^28-bit dictionary walks run on the GPU
 * (kssd_gpu_set_union / kssd_gpu_set_filter); files keep the reference's layouts:
 *   pan directory     cofiles.stat = the 32-byte header of the input only + pan.<c> | uniq_pan.<c> (u32 ids ascending)
 *   filtered sketches cofiles.stat (input's, per-genome counts replaced; header untouched) + combco.<c> + combco.index.<c>
 * Not built: -c (combine pans) and -g (grouping by a taxonomy table), the reference's taxonomy extras.
 * ------------------------------------------------------------------------------------------------- */
static void *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1);
    if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    if (len) *len = n > 0 ? (size_t)n : 0;
    return p;
}

static void spill(const char *path, const void *p, size_t len)
{
    FILE *f = fopen(path, "wb");
    if (!f) die(errno, "%s", path);
    if (len && fwrite(p, 1, len, f) != len) die(EIO, "%s", path);
    if (fclose(f) != 0) die(EIO, "%s", path);
}

typedef struct { /* co_dstat_t, global_basic.h:94-103 */
    uint32_t shuf_id;
    uint32_t koc;
    int32_t kmerlen, dim_rd_len, comp_num, infile_num;
    uint64_t all_ctx_ct;
} co_hdr;

static int cmd_set(int argc, char **argv)
{
    int op = -1, print = 0, device = 0; /* 0 subtract, 1 intersect, 2 union, 3 uniq union (command_set.c:104-143) */
    char pan[4096] = "", outdir[4096] = "./";
    static struct option lo[] = {{"union", 0, 0, 'u'}, {"subtract", 1, 0, 's'}, {"intsect", 1, 0, 'i'}, {"uniq_union", 0, 0, 'q'},
                                 {"combin_pan", 0, 0, 'c'}, {"threads", 1, 0, 'p'}, {"print", 0, 0, 'P'}, {"grouping", 1, 0, 'g'},
                                 {"outdir", 1, 0, 'o'}, {"device", 1, 0, 997}, {0, 0, 0, 0}};
    int c;
    optind = 1;
    while ((c = getopt_long(argc, argv, "us:i:qcp:Pg:o:", lo, NULL)) != -1) {
        switch (c) {
        case 'u': if (op != -1) printf("set operation is already set, -u is ignored.\n"); else op = 2; break;
        case 'q': if (op != -1) printf("set operation is already set, -q is ignored.\n"); else op = 3; break;
        case 's': if (op != -1) printf("set operation is already set, -s is ignored.\n"); else { op = 0; snprintf(pan, sizeof pan, "%s", optarg); } break;
        case 'i': if (op != -1) printf("set operation is already set, -i is ignored.\n"); else { op = 1; snprintf(pan, sizeof pan, "%s", optarg); } break;
        case 'c': case 'g': die(ENOTSUP, "set -c / -g (pan combination, taxonomy grouping) are outside this build (SURVEY.md section 8f)");
        case 'p': break; /* host threads: nothing to spread, the dictionary work is on the device */
        case 'P': print = 1; break;
        case 'o': snprintf(outdir, sizeof outdir, "%s", optarg); break;
        case 997: device = atoi(optarg); break;
        default: die(EINVAL, "set: unknown option");
        }
    }
    if (argc - optind < 1) die(EINVAL, "set: need a sketch directory");
    const char *in = argv[optind];
    char path[8192];
    snprintf(path, sizeof path, "%s/cofiles.stat", in);
    size_t stat_len = 0;
    unsigned char *stat_bytes = slurp(path, &stat_len);
    if (!stat_bytes || stat_len < sizeof(co_hdr)) die(ENOENT, "cannot find cofiles.stat under %s ", in);
    co_hdr h;
    memcpy(&h, stat_bytes, sizeof h);
    if (stat_len != sizeof h + (size_t)h.infile_num * (4 + KSSD_PATHLEN)) die(EINVAL, "%s: not a cofiles.stat", path);
    if (print) { /* print_gnames, command_set.c:515-532 */
        const char *names = (const char *)stat_bytes + sizeof h + (size_t)h.infile_num * 4;
        for (int i = 0; i < h.infile_num; i++) printf("%s\n", names + (size_t)i * KSSD_PATHLEN);
        free(stat_bytes);
        return 0;
    }
    if (op == -1) die(EINVAL, "set: choose one of -u, -q, -s <pan>, -i <pan>, -P");
    gck(kssd_gpu_create_for_dist(&g_ctx, h.kmerlen, device), "kssd_gpu_create_for_dist");
    mkdir(outdir, 0777);
    if (op >= 2) { /* sketch_union / uniq_sketch_union */
        snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
        spill(path, &h, sizeof h);
        for (int comp = 0; comp < h.comp_num; comp++) {
            size_t len = 0;
            snprintf(path, sizeof path, "%s/combco.%d", in, comp);
            uint32_t *ids = slurp(path, &len);
            if (!ids) die(errno ? errno : ENOENT, "sketch_union():%s", path);
            uint32_t *out = NULL;
            uint64_t n_out = 0;
            gck(kssd_gpu_set_union(g_ctx, ids, len / 4, op == 3, &out, &n_out), "set union");
            snprintf(path, sizeof path, "%s/%s.%d", outdir, op == 3 ? "uniq_pan" : "pan", comp);
            spill(path, out, (size_t)n_out * 4);
            kssd_gpu_free(out);
            free(ids);
        }
    } else { /* sketch_operate */
        snprintf(path, sizeof path, "%s/cofiles.stat", pan);
        size_t plen = 0;
        co_hdr *ph = slurp(path, &plen);
        if (!ph || plen < sizeof(co_hdr)) die(ENOENT, "cannot find cofiles.stat under %s ", pan);
        if (ph->shuf_id != h.shuf_id) die(EINVAL, "sketcing id not match(%d Vs. %d)", (int)h.shuf_id, (int)ph->shuf_id);
        const int pan_comp = ph->comp_num;
        free(ph);
        uint32_t *ctx_ct = (uint32_t *)(stat_bytes + sizeof h);
        memset(ctx_ct, 0, (size_t)h.infile_num * 4);
        for (int comp = 0; comp < pan_comp; comp++) {
            size_t pl = 0, il = 0, xl = 0;
            snprintf(path, sizeof path, "%s/pan.%d", pan, comp);
            uint32_t *pids = slurp(path, &pl);
            if (!pids) {
                snprintf(path, sizeof path, "%s/uniq_pan.%d", pan, comp);
                pids = slurp(path, &pl);
                if (!pids) die(ENOENT, "sketch_operate():%s", path);
            }
            snprintf(path, sizeof path, "%s/combco.index.%d", in, comp);
            uint64_t *idx = slurp(path, &xl);
            snprintf(path, sizeof path, "%s/combco.%d", in, comp);
            uint32_t *ids = slurp(path, &il);
            if (!idx || !ids || xl != ((size_t)h.infile_num + 1) * 8 || il != (size_t)idx[h.infile_num] * 4)
                die(EINVAL, "sketch_operate():%s", path);
            uint64_t *ooff = NULL;
            uint32_t *oids = NULL;
            gck(kssd_gpu_set_filter(g_ctx, idx, ids, (uint32_t)h.infile_num, pids, pl / 4, op, &ooff, &oids), "set filter");
            for (int g = 0; g < h.infile_num; g++) ctx_ct[g] += (uint32_t)(ooff[g + 1] - ooff[g]);
            snprintf(path, sizeof path, "%s/combco.%d", outdir, comp);
            spill(path, oids, (size_t)ooff[h.infile_num] * 4);
            snprintf(path, sizeof path, "%s/combco.index.%d", outdir, comp);
            spill(path, ooff, ((size_t)h.infile_num + 1) * 8);
            kssd_gpu_free(ooff);
            kssd_gpu_free(oids);
            free(pids);
            free(idx);
            free(ids);
        }
        snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
        spill(path, stat_bytes, stat_len);
    }
    free(stat_bytes);
    kssd_gpu_destroy(g_ctx);
    g_ctx = NULL;
    return 0;
}

/* kssd reverse -L <.shuf> -o <outdir> <sketch dir> (command_reverse.c): the k-mers behind the sketches, host only */
static int cmd_reverse(int argc, char **argv)
{
    char shuf_path[4096] = "", outdir[4096] = ".";
    static struct option lo[] = {{"shufFile", 1, 0, 'L'}, {"outdir", 1, 0, 'o'}, {"threads", 1, 0, 'p'}, {"byreads", 0, 0, 'b'}, {0, 0, 0, 0}};
    int c, byreads = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "L:o:p:b", lo, NULL)) != -1) {
        switch (c) {
        case 'L': snprintf(shuf_path, sizeof shuf_path, "%s", optarg); break;
        case 'o': snprintf(outdir, sizeof outdir, "%s", optarg); break;
        case 'p': break;
        case 'b': byreads = 1; break;
        default: die(EINVAL, "reverse: unknown option");
        }
    }
    if (argc - optind < 1) die(EINVAL, "need speficy the query path");
    if (!(kssd_probe_dir(argv[optind]) & 1)) die(EINVAL, "%s is not a valid query folder", argv[optind]);
    kssd_shuf sh;
    int rc = kssd_shuf_read(&sh, shuf_path);
    if (rc) die(EIO, "read_dim_shuffle_file(): %s: %s", shuf_path, kssd_host_strerror(rc));
    if (byreads) { /* co_rvs2kmer_byreads prints to stdout (command_reverse.c:201-212) */
        rc = kssd_reverse_byreads(&sh, argv[optind], stdout);
        if (rc) die(EIO, "co_rvs2kmer_btreads(): %s", kssd_host_strerror(rc));
        kssd_shuf_release(&sh);
        return 0;
    }
    mkdir(outdir, 0777);
    rc = kssd_reverse_dir(&sh, argv[optind], outdir);
    if (rc) die(EIO, "co_reverse2kmer(): %s", kssd_host_strerror(rc));
    kssd_shuf_release(&sh);
    return 0;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IOLBF, 0);
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
        printf("%s\n\nUsage: kssd <subcommand> [OPTION...] [arguments ...]\nSupported subcommands are:\n\n"
               "  shuffle\tshuffle/sampling k-mer substring space.\n\n  dist   \tsequences sketching and distance estimation.\n\n"
               "  set    \tset operations on sketches: union, uniq union, subtract, intersect.\n\n"
               "  reverse\trecover the k-mers behind sketches.\n",
               VERSION);
        return argc < 2;
    }
    if (!strcmp(argv[1], "--version") || !strcmp(argv[1], "-V")) { printf("%s\n", VERSION); return 0; }
    int rc;
    if (!strcmp(argv[1], "shuffle")) rc = cmd_shuffle(argc - 1, argv + 1);
    else if (!strcmp(argv[1], "dist")) rc = cmd_dist(argc - 1, argv + 1);
    else if (!strcmp(argv[1], "set")) rc = cmd_set(argc - 1, argv + 1);
    else if (!strcmp(argv[1], "reverse")) rc = cmd_reverse(argc - 1, argv + 1);
    else die(EINVAL, "%s is not a valid subcommand (this build implements: shuffle, dist, set, reverse)", argv[1]);
    /* every file is closed, every context destroyed: leave without the GPU runtime's exit handlers (tens of milliseconds
     * of a command that takes 0.2 s; KSSD_SLOW_EXIT=1 runs them) */
    fflush(NULL);
    if (!getenv("KSSD_SLOW_EXIT")) _exit(rc);
    return rc;
}
