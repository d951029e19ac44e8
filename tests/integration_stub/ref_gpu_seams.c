/*
 * ref_gpu_seams.c -- the binding INTEGRATION.md describes, written out against the REFERENCE's own headers
 * (yhg926/public_kssd v1.2.21: command_dist.h, command_dist_wrapper.h, command_shuffle.h, global_basic.h): the two
 * functions a kssd maintainer would add to command_dist.c to route the hot loops through libkssd_gpu.so.
 *
 * This file is not part of the product and is never linked into it.  tests/test_integration_stub.py compiles it with
 *     gcc -std=gnu11 -fsyntax-only -I/root/reference -I include -I public_kssd_amd/host
 * where the reference sources exist (the dev container), which keeps the stubs honest: the reference's types, globals
 * and helper names used below are the real ones.  Original code; it cites the reference lines it replaces.
 */
#include <err.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "command_dist.h"         /* reference: dim_shuffle, hashsize, component_num, dist_opt_val_t */
#include "command_dist_wrapper.h"
#include "command_shuffle.h"      /* reference: dim_shuffle_t */
#include "global_basic.h"         /* reference: infile_tab_t, isOK_fmt_infile, fastq_fmt, co_dstat_t, PATHLEN */

#include "kssd_gpu.h"             /* this repository: include/kssd_gpu.h */
#include "kssd_host.h"            /* this repository: public_kssd_amd/host/kssd_host.h */

/* replaces the body of run_stageI (command_dist.c:258-380): every file of the table through the device, the
 * sketch container written in the reference's layout.  Returns the path of cofiles.stat like run_stageI does. */
const char *run_stageI_gpu(dist_opt_val_t *opt, infile_tab_t *files, int *order, const char *co_dir, int device)
{
    kssd_shuf_hdr hdr = {dim_shuffle->dim_shuffle_stat.id, dim_shuffle->dim_shuffle_stat.k, dim_shuffle->dim_shuffle_stat.subk,
                         dim_shuffle->dim_shuffle_stat.drlevel};
    kssd_gpu_ctx *ctx = NULL;
    if (kssd_gpu_create(&ctx, &hdr, dim_shuffle->shuffled_dim, device) != KSSD_OK) err(EIO, "kssd_gpu_create: %s", kssd_gpu_last_hip_error());
    kssd_gpu_info info;
    kssd_gpu_get_info(ctx, &info);
    if (info.hashsize != hashsize || info.comp_num != component_num) errx(EINVAL, "derived constants differ from seq2co_global_var_initial()");

    /* tokenise (the byte rules of iseq2comem.c:213-242, 289-321) into page-locked memory */
    kssd_batch *b = kssd_batch_create_ex(kssd_gpu_host_alloc, kssd_gpu_host_free);
    int all_fq = 1;
    for (int i = 0; i < files->infile_num; i++) {
        char *p = files->organized_infile_tab[order[i]].fpath;
        int fq = isOK_fmt_infile(p, fastq_fmt, FQ_FMT_SZ);
        all_fq &= fq;
        if (kssd_batch_add_file(b, p, fq, opt->kmerqlty, NULL) != KSSD_HOST_OK) err(EIO, "%s", p);
    }
    uint32_t flags = all_fq ? (KSSD_SKETCH_KEEP_ZERO | KSSD_SKETCH_NO_CAPACITY) : (opt->u ? KSSD_SKETCH_UNIQ : KSSD_SKETCH_FASTA);
    uint64_t *off = NULL;
    uint32_t *ids = NULL, *pos = NULL;
    int64_t bad = -1;
    int rc = kssd_gpu_sketch_batch_pos(ctx, kssd_batch_packed(b), kssd_batch_mask(b), kssd_batch_chunk_off(b), kssd_batch_n_genomes(b), flags,
                                       all_fq ? (uint32_t)opt->kmerocrs : 1u, &off, &ids, &pos, &bad);
    if (rc == KSSD_ERR_CAPACITY) /* the abort of iseq2comem.c:262-263 */
        err(errno, "the context space is too crowd, try rerun the program using -k%d", hdr.k + 1);
    if (rc != KSSD_OK) err(EIO, "sketch: %s", kssd_gpu_strerror(rc));
    /* the reference's file order inside a genome: its hash-slot order (iseq2comem.c:538-546) */
    for (uint32_t g = 0; g < kssd_batch_n_genomes(b); g++) kssd_slot_order_pos(ids + off[g], pos + off[g], off[g + 1] - off[g], hashsize);

    kssd_sketchset s;
    memset(&s, 0, sizeof s);
    s.shuf_id = (uint32_t)hdr.id;
    s.kmerlen = info.kmerlen;
    s.dim_rd_len = info.dim_rd_len;
    s.comp_num = info.comp_num;
    s.n = kssd_batch_n_genomes(b);
    s.off = off;
    s.ids = ids;
    s.names = malloc((size_t)s.n * KSSD_PATHLEN);
    for (uint32_t g = 0; g < s.n; g++) strncpy(s.names[g], files->organized_infile_tab[order[g]].fpath, KSSD_PATHLEN - 1);
    if (kssd_sketchset_write(&s, co_dir, hashsize, 0) != KSSD_HOST_OK) err(EIO, "%s", co_dir); /* command_dist.c:314-378 */
    free(s.names);
    kssd_gpu_free(off);
    kssd_gpu_free(ids);
    kssd_gpu_free(pos);
    kssd_batch_destroy(b);
    kssd_gpu_destroy(ctx);
    char *full = malloc(PATHLEN);
    snprintf(full, PATHLEN, "%s/%s", co_dir, co_dstat);
    return full;
}

/* replaces the counting loop of mco_cbdco_nobin_dist (command_dist.c:763-790): shared-k-mer counts of every query row
 * against every reference straight into the mapped sharedk_ct.dat, query rows sharded over n_gpus devices;
 * dist_print_nobin (command_dist.c:1161) runs unchanged afterwards */
void mco_cbdco_nobin_dist_gpu(const char *refdir, const char *qrydir, const char *distout_dir, int n_gpus)
{
    kssd_sketchset ref, qry;
    if (kssd_sketchset_read(&ref, refdir) != KSSD_HOST_OK || kssd_sketchset_read(&qry, qrydir) != KSSD_HOST_OK) err(EIO, "sketch directories");
    char path[PATHLEN];
    snprintf(path, sizeof path, "%s/sharedk_ct.dat", distout_dir);
    const size_t bytes = (size_t)ref.n * qry.n * sizeof(ctx_obj_ct_t);
    int fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600); /* the reference refuses to overwrite, command_dist.c:741-746 */
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) err(errno, "mco_cbdco_nobin_dist()::%s", path);
    ctx_obj_ct_t *ctx_obj_ct = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (ctx_obj_ct == MAP_FAILED) err(errno, "mmap %s", path);
    int devices[64];
    if (n_gpus < 1 || n_gpus > 64) n_gpus = 1;
    for (int i = 0; i < n_gpus; i++) devices[i] = i;
    if (kssd_gpu_dist_multi(devices, n_gpus, qry.kmerlen, ref.off, ref.ids, ref.n, qry.off, qry.ids, qry.n, ctx_obj_ct, NULL, NULL, NULL, NULL) != KSSD_OK)
        err(EIO, "kssd_gpu_dist_multi: %s", kssd_gpu_last_hip_error());
    msync(ctx_obj_ct, bytes, MS_SYNC);
    munmap(ctx_obj_ct, bytes);
    close(fd);
    kssd_sketchset_release(&ref);
    kssd_sketchset_release(&qry);
}
