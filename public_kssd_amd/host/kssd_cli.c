/*
 * kssd_cli.c -- `kssd shuffle`, the options of `kssd dist` and its dispatch, on MI355X.
 *
 * Same sub-commands, flags, directory protocol and files as the reference front end
 * (global_wrapper.c:82-145, command_shuffle.c:33-130, command_dist_wrapper.c:41-319, dist_dispatch
 * command_dist.c:53-192); the two hot loops run on the GPU through include/kssd_gpu.h.
 * Host C; there is no CPU implementation of the hot path behind it: without a gfx950 device the
 * sketch and search modes stop with an error.  The stages themselves: kssd_cli_stage1.c, kssd_cli_search.c,
 * kssd_cli_set.c (see kssd_cli.h).
 */
#include "kssd_cli.h"


char g_unlink_on_die[KSSD_PATHLEN + 64]; /* a file this run created and has not finished (sharedk_ct.dat) */

void die(int code, const char *fmt, ...)
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
const char *acpt[] = {"fna", "fas", "fasta", "fq", "fastq", "fa", "co", NULL};
const char *fq_fmt[] = {"fq", "fastq", NULL};

int has_fmt(const char *name, const char **fmts)
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

void collect_inputs(filelist *f, int nargs, char **args, const char *listfile)
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
int default_threads(void)
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

kssd_gpu_ctx *g_ctx;

double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}


void gck(int rc, const char *what)
{
    if (rc != KSSD_OK) die(rc == KSSD_ERR_CAPACITY ? ENOSPC : rc == KSSD_ERR_NO_DEVICE ? ENODEV : EIO, "%s: %s", what, kssd_gpu_strerror(rc));
}

/* the .shuf of this run: -L <file>, or a fresh default.shuf in the output directory (get_dim_shuffle, command_dist.c:193-216).
 * What comes back is the header and the accepted sub-contexts -- all the sketch path looks at (kssd_shuf_read_core: from the
 * .core file beside the .shuf when there is one, so that a run does not start by reading 64 MiB or 1 GiB for 16 KiB of it) */

void load_shuf(const dist_opt *o, shuf_core *s)
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
void *warm_exchange(void *arg)
{
    const xwarm_arg *x = arg;
    kssd_gpu_exchange_warm_up(x->devs, x->n); /* (a failure is reported by the exchange itself) */
    return NULL;
}

/* set once the HIP runtime and a device's context are up (warm_device, or the first worker's context): from then on text buffers
 * are page-locked */
volatile int g_runtime_ready;

void *warm_device(void *arg)
{
    if (kssd_gpu_warm_up(*(const int *)arg) == KSSD_OK) g_runtime_ready = 1; /* (a missing device is reported by the call that needs it) */
    return NULL;
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

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IOLBF, 0);
    struct timespec t_main;
    clock_gettime(CLOCK_REALTIME, &t_main);
    /* The host threads of this command wait for one another every few milliseconds (a wave of files, a job's genomes), and libgomp's
     * threads wait by spinning unless told otherwise BEFORE the library initialises: under a CPU quota (a container's cpu.max) spinning
     * teams get the whole command stopped for the rest of every accounting period (profiles/r05v_throttle_probe.txt).  libkssd_env.so's
     * constructor has set the passive policy in front of libgomp's own (host/kssd_env.c) -- the thread limit it set with it is the
     * proof.  Where the loader ran the two the other way round the command goes on as it is: spinning costs time, and a process that
     * links the GPU runtime is never replaced by another (rounds 5's restart is gone); KSSD_TIMING says so. */
    const char *mark = getenv("OMP_THREAD_LIMIT"); /* (set by the constructor, yet without effect: it ran behind libgomp's) */
    if (getenv("KSSD_TIMING") && omp_get_thread_limit() != 1000003 && ((mark && !strcmp(mark, "1000003")) || (!getenv("OMP_WAIT_POLICY") && !getenv("GOMP_SPINCOUNT"))))
        fprintf(stderr, "{\"kssd_timing\": \"wait_policy\", \"note\": \"libkssd_env.so was initialised behind libgomp: host threads spin; set OMP_WAIT_POLICY=passive\"}\n");
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
    if (getenv("KSSD_TIMING")) { /* wall-clock stamps of main()'s two ends: what a caller's clock sees before and after them is the loader's and the kernel's */
        struct timespec t_end;
        clock_gettime(CLOCK_REALTIME, &t_end);
        fprintf(stderr, "{\"kssd_timing\": \"process\", \"unix_main\": %.6f, \"unix_exit\": %.6f}\n", (double)t_main.tv_sec + 1e-9 * (double)t_main.tv_nsec,
                (double)t_end.tv_sec + 1e-9 * (double)t_end.tv_nsec);
    }
    fflush(NULL);
    if (!getenv("KSSD_SLOW_EXIT")) _exit(rc);
    return rc;
}
