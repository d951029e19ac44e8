/*
 * kssd_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the kssd sketch + distance hot path, written from the behaviour of
 * the reference (yhg926/public_kssd v1.2.21, /root/reference) -- every function cites the
 * reference file:line it follows.  It is the parity checker for the HIP path and the "port" CPU
 * baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing under public_kssd_amd/ links, imports or calls it.
 *
 * Pinning: the reference ships no tests or golden vectors, so this oracle is pinned
 * against outputs of the reference itself, compiled from its own sources into oracle/_ref/kssd
 * (oracle/Makefile) and run on the fixtures under tests/golden/ (tests/golden/make_golden.py), and
 * live against oracle/_ref/kssd whenever that binary is present (tests/test_oracle_vs_ref.py).
 */
#ifndef KSSD_ORACLE_H
#define KSSD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (negative returns) */
#define KO_ERR_CAPACITY  (-2) /* iseq2comem.c:262-263 "the context space is too crowd"            */
#define KO_ERR_HEADER    (-3) /* iseq2comem.c:233 header not terminated before EOF                  */
#define KO_ERR_EMPTY     (-4) /* iseq2comem.c:202 first fread returned nothing                      */
#define KO_ERR_PARAM     (-5) /* command_dist.c:221-234 primer index out of range, subk>=8 ...      */
#define KO_ERR_IO        (-6)
#define KO_ERR_BUFSZ     (-7) /* caller's output buffer too small                                   */

typedef struct ko_params {
    int shuf_id, k, subk, drlevel;        /* command_shuffle.h:17-23                                */
    int TL;                               /* 2k                                   iseq2comem.c:68   */
    int out;                              /* k - subk                             iseq2comem.c:59   */
    int comp_num, comp_bits;              /* iseq2comem.c:63-64, :527                               */
    int rc_shift;                         /* 4k-2                                 iseq2comem.c:66   */
    uint64_t tupmask, domask, undomask;   /* iseq2comem.c:67-71                                     */
    int64_t dim_end;                      /* iseq2comem.c:74-76                                     */
    uint32_t hashsize, hashlimit;         /* command_dist.c:217-236, iseq2comem.c:61                */
} ko_params;

typedef struct ko_ctx ko_ctx;

/* derive constants only (no table): returns 0 or KO_ERR_PARAM */
int ko_params_init(ko_params *p, int shuf_id, int k, int subk, int drlevel);

/* table = the .shuf permutation, int32[16^subk]; borrowed, must outlive the ctx */
ko_ctx *ko_open(const int32_t *table, int shuf_id, int k, int subk, int drlevel);
void ko_close(ko_ctx *c);
const ko_params *ko_get_params(const ko_ctx *c);

/* .shuf file io (command_shuffle.c:184-207). ko_shuf_read mallocs *table (free with ko_free). */
int ko_shuf_read(const char *path, int hdr[4], int32_t **table);
void ko_free(void *p);

/*
 * Sketch one FASTA byte stream (fasta2co + wrt_co2cmpn_use_inn_subctx, iseq2comem.c:188-273,525-551;
 * uniq!=0 -> uniq_fasta2co, :616-703).  ids/comps receive the dump in HASH-SLOT ORDER (the order of
 * the reference's <i>.co.<c> files): ids[i] = drtuple >> comp_bits, comps[i] = drtuple % comp_num.
 * Returns the number of entries, or a negative KO_ERR_*.
 */
long ko_fasta2co(ko_ctx *c, const unsigned char *text, size_t n, int uniq,
                 uint32_t *ids, uint8_t *comps, size_t cap);

/* FASTQ: fastq2co + write_fqco2file (iseq2comem.c:277-356,499-524). Q = raw ASCII quality floor,
 * M = least occurrence (1..7). */
long ko_fastq2co(ko_ctx *c, const unsigned char *text, size_t n, int Q, int M,
                 uint32_t *ids, uint8_t *comps, size_t cap);

/* FASTQ with occurrence counts (dist -A): mt_shortreads2koc on one thread + write_fqkoc2files
 * (iseq2comem.c:554-615,435-471); counts[i] = occurrences of ids[i], saturated at 65535 */
long ko_fastq2koc(ko_ctx *c, const unsigned char *text, size_t n, uint32_t *ids, uint8_t *comps,
                  uint16_t *counts, size_t cap);

/* FASTA by read (dist --byread): reads2mco (iseq2comem.c:78-186).  The stream of sampled k-mers in sequence order,
 * repeats included; read_of[i] = number of '>' met before entry i; *n_reads = number of '>' in the text */
long ko_reads2mco(ko_ctx *c, const unsigned char *text, size_t n, uint32_t *ids, uint8_t *comps,
                  uint32_t *read_of, size_t cap, uint64_t *n_reads);

/* same, reading a (possibly gzip'ed) file through zlib; is_fastq selects the scanner */
long ko_sketch_file(ko_ctx *c, const char *path, int is_fastq, int uniq, int Q, int M,
                    uint32_t *ids, uint8_t *comps, size_t cap);

/* many files, OpenMP over files like run_stageI (command_dist.c:277-312). off[nfiles+1] exclusive
 * prefix, ids concatenated in hash-slot order (component 0 only; comp_num must be 1).
 * Returns total ids or negative error. */
long ko_sketch_files(const int32_t *table, int shuf_id, int k, int subk, int drlevel,
                     const char *const *paths, int nfiles, int threads,
                     uint64_t *off, uint32_t *ids, size_t cap);

/* many in-memory FASTA texts (bench.py cpu_baseline "port" leg) */
long ko_sketch_texts(const int32_t *table, int shuf_id, int k, int subk, int drlevel,
                     const unsigned char *const *texts, const size_t *lens, int ntexts, int threads,
                     uint64_t *off, uint32_t *ids, size_t cap);

/*
 * Inverted index + posting traversal: shared[q*R + r] = |S_q intersect S_r|
 * (combco2mco co2mco.c:25-77 + mco_cbdco_nobin_dist command_dist.c:763-790).
 * Sketches are CSR (off[n+1], ids); ids need not be sorted. Returns 0.
 */
int ko_shared_counts(const uint64_t *roff, const uint32_t *rids, int R,
                     const uint64_t *qoff, const uint32_t *qids, int Q,
                     uint32_t *shared, int threads);

/* the inverted index itself, compact form: returns number of distinct ids U; uid[U] ascending,
 * upos[U+1] exclusive offsets into post[] (gids ascending inside each posting, co2mco.c:40-55).
 * Buffers must hold roff[R] entries (+1 for upos). */
long ko_build_index(const uint64_t *roff, const uint32_t *rids, int R,
                    uint32_t *uid, uint64_t *upos, uint32_t *post);

typedef struct ko_metric {
    double metric;   /* Jaccard or containment                    command_dist.c:1262-1264 */
    double dist;     /* MashD or AafD, clamped to 1                               :1265-1266 */
    double pv, fdr;  /*                                                          :1272-1274 */
    double ci_m1, ci_m2, ci_d1, ci_d2; /*                                        :1277-1280 */
    uint32_t rs_u;   /* (unsigned)rs printed in the Shared_k column              :1269      */
    int skipped;     /* dist > dthreshold                                        :1267      */
} ko_metric;

/* output_ctrl arithmetic (command_dist.c:1251-1281). X=|ref|, Y=|qry|, s=shared, metric_sel 0=Jcd 1=Ctm */
void ko_output_ctrl(uint32_t X, uint32_t Y, uint32_t s, int kmerlen, int dim_rd_len,
                    int metric_sel, int correction, double dthreshold, uint64_t cmprsn_num,
                    ko_metric *m);

/* (Jaccard, MashD, containment, AafD) of n triples with rs = 0, host libm (command_dist.c:1262-1266) */
void ko_metrics_batch(const uint32_t *X, const uint32_t *Y, const uint32_t *S, size_t n, int kmerlen,
                      double *J, double *MD, double *Cc, double *AD);

/* one distance.out line exactly as output_ctrl prints it; returns length (0 if skipped) */
int ko_format_line(char *buf, size_t cap, const char *qname, const char *rname,
                   uint32_t X, uint32_t Y, uint32_t s, int kmerlen, int dim_rd_len,
                   int metric_sel, int pfield, int correction, double dthreshold,
                   uint64_t cmprsn_num);

/* whole distance.out (dist_print_nobin, command_dist.c:1161-1250) incl. header and -N top-n.
 * names are fixed 256-byte records like cofiles.stat. Returns 0 or KO_ERR_IO. */
int ko_dist_print(const char *path, const uint32_t *shared, int R, int Q,
                  const uint32_t *ref_sz, const uint32_t *qry_sz,
                  const char *refnames, const char *qrynames,
                  int kmerlen, int dim_rd_len, int metric_sel, int pfield, int correction,
                  double dthreshold, int n_max);

#ifdef __cplusplus
}
#endif
#endif
