/* kssd_cli.h -- what the pieces of the `kssd` command line share (host/kssd_cli*.c): the options of `dist`
 * (command_dist_wrapper.c:41-100), the input list, error exit, the .shuf of a run, the runtime warm-up.
 *   kssd_cli.c         main, shuffle, the dist options and dist_dispatch (command_dist.c:53-192)
 *   kssd_cli_stage1.c  stage I: reading, the device workers, cofiles.stat / combco.*, --allpairs (run_stageI, command_dist.c:258-380)
 *   kssd_cli_search.c  stage II files, the search and its report, combine_queries (command_dist.c:381-417, 670-808, 1161-1475)
 *   kssd_cli_set.c     kssd set, kssd reverse (command_set.c, command_reverse.c) */
#ifndef KSSD_CLI_H
#define KSSD_CLI_H
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

extern char g_unlink_on_die[KSSD_PATHLEN + 64]; /* a file this run created and has not finished (sharedk_ct.dat) */
void die(int code, const char *fmt, ...) __attribute__((noreturn, format(printf, 2, 3)));

extern const char *acpt[];   /* accepted input suffixes (global_basic.h:129-150) */
extern const char *fq_fmt[];
int has_fmt(const char *name, const char **fmts);

typedef struct {
    char (*path)[KSSD_PATHLEN];
    int n, cap;
} filelist;
void collect_inputs(filelist *f, int nargs, char **args, const char *listfile);
int default_threads(void);

/* dist options (command_dist_wrapper.c:41-100) */
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

extern kssd_gpu_ctx *g_ctx;
double now_s(void);
void gck(int rc, const char *what);

/* the .shuf of this run: header + accepted sub-contexts (kssd_shuf_read_core) */
typedef struct {
    kssd_shuf h; /* table == NULL */
    uint32_t *accepted;
    uint32_t n_accepted;
    int from_cache;
} shuf_core;
void load_shuf(const dist_opt *o, shuf_core *s);

typedef struct {
    const int *devs;
    int n;
} xwarm_arg;
void *warm_exchange(void *arg);
extern volatile int g_runtime_ready;
void *warm_device(void *arg);

void sketch_files(const dist_opt *o, filelist *fl, const char *outdir);
void build_index_files(const char *codir, const char *mcodir);
void search(const dist_opt *o, const char *refdir, const char *qrydir);
void combine_queries(const dist_opt *o);
int cmd_set(int argc, char **argv);
int cmd_reverse(int argc, char **argv);
#endif
