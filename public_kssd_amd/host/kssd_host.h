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
#define KSSD_HOST_ERR_WIDE (-107)   /* a sketch directory of 36-bit tuples (256 components): only written, never read back (see kssd_sketchset.sub) */

#define KSSD_PATHLEN 256 /* PATHLEN, global_basic.h:40: name records in cofiles.stat */

const char *kssd_host_strerror(int code);
void kssd_host_free(void *p); /* for the arrays this library mallocs for its caller */

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
/* what the sketch path needs of a .shuf: its header (table stays NULL) and accepted[r] = the sub-context of rank r,
 * r < n = max(16^(subk - drlevel), 4096) (iseq2comem.c:74-76, 247-249; malloc'd, free()).  Read from "<path>.core" when
 * that file describes the .shuf as it is now (size, mtime, header), otherwise scanned out of the mapped table and the core
 * written for the next run; KSSD_NO_SHUF_CORE=1 neither reads nor writes it.  from_cache (may be NULL): 1 if the core was used */
int kssd_shuf_read_core(const char *path, kssd_shuf *hdr, uint32_t **accepted, uint32_t *n, int *from_cache);
void kssd_shuf_release(kssd_shuf *s);

/* ---- packed batches (layout: include/kssd_gpu.h) ----------------------------------------------- */
typedef struct kssd_batch kssd_batch;
kssd_batch *kssd_batch_create(void);
/* the same with the packed / mask arrays in memory of the caller's choice (e.g. page-locked memory from
 * kssd_gpu_host_alloc, so that the batch can travel to the device by DMA without a staging copy) */
kssd_batch *kssd_batch_create_ex(void *(*alloc)(size_t), void (*release)(void *));
void kssd_batch_destroy(kssd_batch *b);
void kssd_batch_clear(kssd_batch *b);
/* append one genome; tokenisation rules of fasta2co (iseq2comem.c:213-242) */
int kssd_batch_add_fasta(kssd_batch *b, const unsigned char *text, size_t n);
/* append one FASTA file as ONE genome for dist --byread (reads2mco, iseq2comem.c:78-186): same tokenisation, and
 * *read_start (malloc'd, n_reads entries, caller frees) receives for every '>' the position inside the genome at which
 * the bases behind that header begin -- the cut points of the k-mer stream KSSD_SKETCH_BY_POS returns */
int kssd_batch_add_fasta_reads(kssd_batch *b, const unsigned char *text, size_t n, uint64_t **read_start, uint64_t *n_reads);
/* append one read set; framing and quality rule of fastq2co (iseq2comem.c:289-321);
 * *n_lines receives the reference's "reads detected" figure (4 x records) */
int kssd_batch_add_fastq(kssd_batch *b, const unsigned char *text, size_t n, int Q, uint64_t *n_lines);
/* append one read set the way the abundance scanner of dist -A frames it (mt_shortreads2koc, iseq2comem.c:554-581):
 * four lines per read, no quality rule; *n_reads receives the number of reads */
int kssd_batch_add_reads(kssd_batch *b, const unsigned char *text, size_t n, uint64_t *n_reads);
/* read a file (gzip or plain, like `zcat -fc`, iseq2comem.c:187) and append it; is_fastq: 0 FASTA, 1 FASTQ,
 * 2 FASTQ for abundance sketches (kssd_batch_add_reads) */
int kssd_batch_add_file(kssd_batch *b, const char *path, int is_fastq, int Q, uint64_t *n_lines);
const uint32_t *kssd_batch_packed(const kssd_batch *b);
const uint32_t *kssd_batch_mask(const kssd_batch *b);
const uint64_t *kssd_batch_chunk_off(const kssd_batch *b);
uint64_t kssd_batch_n_chunks(const kssd_batch *b);
uint32_t kssd_batch_n_genomes(const kssd_batch *b);
uint64_t kssd_batch_n_positions(const kssd_batch *b, uint32_t genome); /* bases + run breaks */

/* Parallel fill.  kssd_batch_reserve appends n empty genomes with room for max_pos[i] positions each (an input of
 * B bytes never needs more than B positions) and returns the index of the first; kssd_batch_fill_text then tokenises
 * one input into one reserved genome and may run on many threads at once, one genome each.  kind: 0 FASTA, 1 FASTQ
 * (quality floor Q), 2 reads of dist -A.  Unused room stays padding (invalid positions). */
int kssd_batch_reserve(kssd_batch *b, uint32_t n, const uint64_t *max_pos, uint32_t *first);
int kssd_batch_fill_text(kssd_batch *b, uint32_t genome, int kind, const unsigned char *text, size_t n, int Q, uint64_t *n_lines);

/* append every genome of src to dst (used to tokenise files in parallel, one batch per thread) */
int kssd_batch_append(kssd_batch *dst, const kssd_batch *src);

/* whole file into memory through zlib; caller frees *buf */
int kssd_slurp(const char *path, unsigned char **buf, size_t *len);
/* the same into a malloc'd buffer the caller keeps from file to file: *buf / *cap are grown (realloc) when the file
 * does not fit, *len receives the bytes read; the caller frees *buf in the end */
int kssd_slurp_reuse(const char *path, unsigned char **buf, size_t *cap, size_t *len);
/* two files by one thread; gzip'ed ones are unpacked in step (kssd_gunzip_mem2).  rc[f]: kssd_slurp_reuse's for file f */
void kssd_slurp_reuse2(const char *const path[2], unsigned char **buf[2], size_t *cap[2], size_t *len[2], int rc[2]);
/* is the file gzip'ed (magic 1f 8b), and its size on disk; kssd_read_into copies a plain file's bytes (at most cap) into
 * memory of the caller's choice -- e.g. a page-locked buffer the device tokeniser reads from */
int kssd_file_probe(const char *path, int *is_gz, uint64_t *size);
int kssd_read_into(const char *path, unsigned char *dst, size_t cap, size_t *len);
/* gzip members in memory -> their bytes (what `zcat -fc` writes for the file, iseq2comem.c:187,196-208): *out / *cap as
 * kssd_slurp_reuse keeps them (grown with realloc), *len the bytes.  Every member's CRC-32 and length are checked.
 * KSSD_HOST_ERR_IO for a stream that is not gzip or is corrupt.  host/kssd_inflate.c: a table-driven inflate (up to three
 * literals per lookup) -- sequence text is what zlib's byte loop is slowest on.  kssd_crc32: zlib's crc32(), by carry-less
 * multiplication where the CPU has it. */
int kssd_gunzip_mem(const unsigned char *in, size_t in_len, unsigned char **out, size_t *cap, size_t *len);
/* two files by one thread, their symbol loops in step (two independent chains of table lookups fill the core where one leaves it
 * half idle); rc[f] is what kssd_gunzip_mem returns for file f */
void kssd_gunzip_mem2(const unsigned char *const in[2], const size_t in_len[2], unsigned char **out[2], size_t *cap[2], size_t *len[2], int rc[2]);
uint32_t kssd_crc32(uint32_t crc, const unsigned char *p, size_t len);

/* ---- derived constants (seq2co_global_var_initial iseq2comem.c:54-77, get_hashsz command_dist.c:217-236) */
typedef struct kssd_derived {
    int k, subk, drlevel, kmerlen, dim_rd_len;
    int comp_num, comp_bits;
    uint32_t hashsize, hashlimit;
} kssd_derived;
int kssd_derive(kssd_derived *d, int k, int subk, int drlevel);

/* ---- on-disk sketch / index formats -------------------------------------------- */
typedef struct kssd_sketchset {
    uint32_t shuf_id;
    int koc;                 /* abundance flag of co_dstat_t (dist -A): counts below go to / come from combco.<c>.a */
    int kmerlen, dim_rd_len; /* 2k, 2*drlevel */
    int comp_num;
    uint32_t n;              /* genomes */
    uint64_t *off;           /* n+1, exclusive prefix */
    uint32_t *ids;           /* FULL reduced tuples (component folded back in), genome after genome */
    char (*names)[KSSD_PATHLEN];
    uint16_t *counts;        /* koc != 0: occurrences of ids[i], saturated at 65535 (write_fqkoc2files, iseq2comem.c:435-471); else NULL */
    uint8_t *sub;            /* k - drlevel = 9 (36-bit tuples, 256 components): the tuple's low four bits, ids[i] = tuple >> 4 (the device's
                              * passes, include/kssd_gpu.h kssd_gpu_set_tuple_pass); NULL for every other parameter set.  Only the writer
                              * knows it: the reference's own index builder does not survive 256 components (tests/golden/make_golden_k12.py) */
} kssd_sketchset;
void kssd_sketchset_release(kssd_sketchset *s);

/* Order one genome's distinct ids the way the reference's dump leaves them: ascending slot of its
 * double-hashing table (global_basic.h:228-230, iseq2comem.c:538-546).  Ties between ids that probe
 * the same slot depend on which the reference met first in the sequence; here the smaller id wins, so
 * the result is byte-identical to the reference's file unless two ids of the genome collide. */
int kssd_slot_order(uint32_t *ids, uint64_t n, uint32_t hashsize); /* 0, or KSSD_HOST_ERR_NOMEM with the ids untouched (all three) */
/* The same with every id's first position in the genome (kssd_gpu_sketch_batch_pos): the insertions are replayed
 * in sequence order like fasta2co makes them (iseq2comem.c:254-268), so colliding ids land where the reference
 * puts them and the file is byte-identical.  (fastq -n >= 2 and -u also let ids that are dropped later occupy
 * slots: kssd_slot_order_pos_keep below.) */
int kssd_slot_order_pos(uint32_t *ids, const uint32_t *first_pos, uint64_t n, uint32_t hashsize);
/* the same for tuples of more than 32 bits (k - drlevel = 9) */
int kssd_slot_order_pos64(uint64_t *tuples, const uint32_t *first_pos, uint64_t n, uint32_t hashsize);
/* For the modes whose dump drops ids that nevertheless sit in the reference's table and shift later probes (fastq
 * -n >= 2: fewer than n occurrences; -u: seen more than once): ALL distinct ids of the genome with their first
 * positions, keep[i] != 0 for the ids the dump writes.  Returns how many are kept; they come back at the front of
 * ids, in the reference's file order.  UINT64_MAX: out of memory. */
uint64_t kssd_slot_order_pos_keep(uint32_t *ids, const uint32_t *first_pos, const uint8_t *keep, uint64_t n, uint32_t hashsize);
/* the same for tuples of more than 32 bits */
uint64_t kssd_slot_order_pos64_keep(uint64_t *tuples, const uint32_t *first_pos, const uint8_t *keep, uint64_t n, uint32_t hashsize);

/* abundances follow a reordering of one genome's (distinct) ids: counts_after[i] = the count ids_after[i] had in
 * (ids_before, counts_before); KSSD_HOST_ERR_PARAM if an id of ids_after is not among ids_before */
int kssd_counts_follow(const uint32_t *ids_before, const uint16_t *counts_before, const uint32_t *ids_after,
                       uint16_t *counts_after, uint64_t n);

/* write cofiles.stat + combco.<c> + combco.index.<c> (+ combco.<c>.a when koc) like run_stageI (command_dist.c:314-378).
 * slot_order != 0 applies kssd_slot_order per genome (ids are modified in place). */
int kssd_sketchset_write(const kssd_sketchset *s, const char *dir, uint32_t hashsize, int slot_order);
/* what reads2mco leaves in a sketch directory (iseq2comem.c:78-186, run_stageI command_dist.c:267-273,360-378) for ONE
 * input file (with several, every file overwrites the one before): combco.<c> = the k-mer stream of component c
 * (ids[i] % comp_num == c, stored as ids[i] >> comp_bits), combco.index.<c> = cumulative counts over the reads
 * 0..n_reads as int64 WITHOUT a leading zero, cofiles.stat naming all n_files inputs with all_ctx_ct = 0 and zero
 * counts (the reference writes uninitialised memory there).
 * ids/pos = the stream KSSD_SKETCH_BY_POS returned for the file's genome (FULL tuples, positions ascending),
 * read_start = the cut points from kssd_batch_add_fasta_reads. */
int kssd_byread_write(const char *dir, uint32_t shuf_id, int k, int drlevel, const char (*names)[KSSD_PATHLEN], uint32_t n_files,
                      const uint32_t *ids, const uint32_t *pos, uint64_t n, const uint64_t *read_start, uint64_t n_reads);
/* read them back (all components folded into full tuples) */
int kssd_sketchset_read(kssd_sketchset *s, const char *dir);
/* mcofiles.stat + mco.index.<c> + mco.<c> from a sketch set: combco2mco + run_stageII
 * (co2mco.c:25-77, command_dist.c:381-417).  mco.index.<c> is the dense size_t[16^7] the reference maps. */
int kssd_index_write(const kssd_sketchset *s, const char *dir);
/* rebuild a sketch set from mcofiles.stat + mco.* when no combco.* is around */
int kssd_index_read(kssd_sketchset *s, const char *dir);
/* 1 if dir holds cofiles.stat, 2 if mcofiles.stat, 3 both, 0 none (dist_dispatch probing, command_dist.c:62-63) */
int kssd_probe_dir(const char *dir);

/* Stage I and the all-pairs search sharded over a device list (no counterpart in the reference: its threads take files from
 * one OpenMP loop, command_dist.c:277, and rows from another, :774): device d of the list sketches -- and later owns the
 * query rows of -- the inputs [first[d], first[d + 1]) of the (sorted) input list, runs of ceil(n_files / n_devices) files,
 * the last ones shorter or empty.  That is the layout an all-gather of fixed-size units leaves, so a genome's number on every
 * device is its input's index.  first: n_devices + 1 entries.  KSSD_HOST_ERR_PARAM: no device, a negative device, or one
 * named twice (one rank per device). */
int kssd_shard_plan(const int *devices, int n_devices, uint32_t n_files, uint32_t *first);

/* kssd reverse (command_reverse.c:219-321): the canonical 2k-mer behind a sketch id.  accepted[r] = the sub-context
 * whose permutation rank is r (r < 4096, the inverse of the .shuf table on its first 4096 ranks, :223-231);
 * full_id = the reduced tuple with the component folded back in.  Returns the 2k-mer, 2 bits per base, first base in
 * the highest of the 4k bits (core_reverse2unituple, :311-321). */
uint64_t kssd_reverse_id(uint32_t full_id, int k, int subk, int drlevel, const uint32_t *accepted);
/* the 4096 sub-contexts with rank < 4096 out of a full .shuf table; 0 on success */
int kssd_shuf_accepted(const kssd_shuf *s, uint32_t *accepted /*4096*/);
/* write one text file per genome of a sketch directory (named like the genome's file) with one 2k-mer per line, in the
 * order of the ids in combco.<c>, component after component -- co_reverse2kmer (command_reverse.c:219-310) */
int kssd_reverse_dir(const kssd_shuf *s, const char *sketch_dir, const char *outdir);

/* kssd reverse --byreads (co_rvs2kmer_byreads, command_reverse.c:147-218): a --byread sketch directory as text,
 * ">read n" followed by the 2k-mers of the read, on `out` */
int kssd_reverse_byreads(const kssd_shuf *s, const char *sketch_dir, FILE *out);

/* ---- distance report (dist_print_nobin + output_ctrl, command_dist.c:1161-1287) ------------------------- */
typedef struct kssd_print_opt {
    int metric;        /* -M 0 Jaccard / 1 containment */
    int pfield;        /* -O 0/1/2 */
    int correction;    /* --correction */
    double dthreshold; /* -D */
    int n_max;         /* -N, 0 = all */
    int threads;
} kssd_print_opt;
int kssd_distance_print(const char *path, const uint32_t *shared, const kssd_sketchset *ref, const kssd_sketchset *qry,
                        const kssd_print_opt *opt);
/* the same report from the pairs a device-side selection left (kssd_gpu_dist_select): row q's candidates are
 * pair_ref / pair_shared [pair_off[q], pair_off[q+1]), references ascending.  The selection must be a superset of what
 * the options print (every pair with a positive metric for -N, every pair within -D otherwise); the exact -N ranking
 * and the exact -D test are applied here, on the host's libm, so the text equals the dense report's byte for byte. */
int kssd_distance_print_pairs(const char *path, const uint64_t *pair_off, const uint32_t *pair_ref, const uint32_t *pair_shared,
                              const kssd_sketchset *ref, const kssd_sketchset *qry, const kssd_print_opt *opt);

/* the report's two number formats, byte for byte what the C library's "%.6lf" / "%E" write (exact integer arithmetic,
 * the library itself for non-finite and borderline values); p must have room for 32 bytes when |x| < 2^39 or the format
 * is "%E", for 330 otherwise; returns the end of the text */
char *kssd_fmt_f6(char *p, double x);
char *kssd_fmt_e6(char *p, double x);

#ifdef __cplusplus
}
#endif
#endif
