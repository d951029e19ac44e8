/*
 * kssd_host.h -- host-side C of the MI355X kssd path (libkssd_host.so): everything the reference
 * does around its hot loops that has to stay bit-compatible -- .shuf files, FASTA/FASTQ
 * tokenisation into the packed device layout, the on-disk sketch / index / distance formats.
 * Pure C, no GPU calls: the kernels are reached through include/kssd_gpu.h only.
 */
#ifndef KSSD_HOST_H
#define KSSD_HOST_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSSD_HOST_OK 0
#define KSSD_HOST_ERR_IO (-101)
#define KSSD_HOST_ERR_PARAM (-102)  /* command_shuffle.c:163-168 */
#define KSSD_HOST_ERR_HEADER (-103) /* fasta header not closed before EOF, iseq2comem.c:233 */
#define KSSD_HOST_ERR_EMPTY (-104)  /* no input bytes, iseq2comem.c:201-202 */
#define KSSD_HOST_ERR_NOMEM (-105)
#define KSSD_HOST_ERR_FORMAT (-106)

#define KSSD_PATHLEN 256 /* PATHLEN, global_basic.h:40: name records in cofiles.stat */

const char *kssd_host_strerror(int code);

/* ---- .shuf (command_shuffle.c:131-207) ------------------------------------------------------------ */
typedef struct kssd_shuf {
    int32_t id, k, subk, drlevel; /* dim_shuffle_stat_t, command_shuffle.h:17-23 */
    int32_t *table;               /* permutation of 0..16^subk-1 */
} kssd_shuf;

/* Fisher-Yates permutation like write_dim_shuffle_file; `seed` makes it reproducible (the reference
 * seeds rand() with time(NULL); pass seed = 0 for that behaviour). id = a draw of the same PRNG. */
int kssd_shuf_generate(kssd_shuf *s, int k, int subk, int drlevel, uint64_t seed);
int kssd_shuf_write(const kssd_shuf *s, const char *path);
int kssd_shuf_read(kssd_shuf *s, const char *path);
void kssd_shuf_release(kssd_shuf *s);

/* ---- packed batches (layout: include/kssd_gpu.h) ----------------------------------------------- */
typedef struct kssd_batch kssd_batch;
kssd_batch *kssd_batch_create(void);
void kssd_batch_destroy(kssd_batch *b);
void kssd_batch_clear(kssd_batch *b);
/* append one genome; tokenisation rules of fasta2co (iseq2comem.c:213-242) */
int kssd_batch_add_fasta(kssd_batch *b, const unsigned char *text, size_t n);
/* append one read set; framing and quality rule of fastq2co (iseq2comem.c:289-321);
 * *n_lines receives the reference's "reads detected" figure (4 x records) */
int kssd_batch_add_fastq(kssd_batch *b, const unsigned char *text, size_t n, int Q, uint64_t *n_lines);
/* read a file (gzip or plain, like `zcat -fc`, iseq2comem.c:187) and append it */
int kssd_batch_add_file(kssd_batch *b, const char *path, int is_fastq, int Q, uint64_t *n_lines);
const uint32_t *kssd_batch_packed(const kssd_batch *b);
const uint32_t *kssd_batch_mask(const kssd_batch *b);
const uint64_t *kssd_batch_chunk_off(const kssd_batch *b);
uint64_t kssd_batch_n_chunks(const kssd_batch *b);
uint32_t kssd_batch_n_genomes(const kssd_batch *b);
uint64_t kssd_batch_n_positions(const kssd_batch *b, uint32_t genome); /* bases + run breaks */

/* whole file into memory through zlib; caller frees *buf */
int kssd_slurp(const char *path, unsigned char **buf, size_t *len);

#ifdef __cplusplus
}
#endif
#endif
