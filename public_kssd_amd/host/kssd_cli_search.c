/* kssd_cli_search.c -- what `kssd dist` does with sketch directories: the stage II files for the reference binary (run_stageII,
 * command_dist.c:381-417), the search and its report (mco_cbdco_nobin_dist + dist_print_nobin, :670-808, :1161-1250), and
 * combine_queries (:1323-1475). */
#include "kssd_cli.h"

/* stage II (run_stageII, command_dist.c:381-417): the index FILES are for the reference binary; our own search
 * builds its index on the device straight from the sketches */
void build_index_files(const char *codir, const char *mcodir)
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
void search(const dist_opt *o, const char *refdir, const char *qrydir)
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
void combine_queries(const dist_opt *o)
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

