/* kssd_cli_set.c -- `kssd set` and `kssd reverse` (command_set.c, command_reverse.c). */
#include "kssd_cli.h"

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

int cmd_set(int argc, char **argv)
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
int cmd_reverse(int argc, char **argv)
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

