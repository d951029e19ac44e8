/*
 * kssd_gpu.h -- C ABI of libkssd_gpu.so: the MI355X (gfx950) implementation of the kssd
 * sketch + distance hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (yhg926/public_kssd v1.2.21) has no FFI: its seams are C functions inside one
 * binary.  Each entry point below names the reference function it replaces.
 * The binding a kssd maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: one calling thread per kssd_gpu_ctx.  All work is enqueued on the hipStream_t passed
 * as `stream` (void*, NULL = the null stream).  Device-level calls (`*_device`) take DEVICE
 * pointers and do not synchronise; host-level calls take HOST pointers and return finished data.
 */
#ifndef KSSD_GPU_H
#define KSSD_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------------------------- */
#define KSSD_OK 0
#define KSSD_ERR_HIP (-1)         /* a HIP runtime call failed (kssd_gpu_strerror gives the text)      */
#define KSSD_ERR_PARAM (-2)       /* k/subk/drlevel outside what the reference accepts                 */
                                  /*   (command_shuffle.c:163-168, command_dist.c:217-236)             */
#define KSSD_ERR_CAPACITY (-3)    /* a genome holds more distinct k-mers than the reference's hash     */
                                  /*   admits: "the context space is too crowd" iseq2comem.c:262-263   */
#define KSSD_ERR_OVERFLOW (-4)    /* an output / staging buffer was too small (call again larger)      */
#define KSSD_ERR_UNSUPPORTED (-5) /* valid for the reference, not implemented on the device yet        */
#define KSSD_ERR_NOMEM (-6)
#define KSSD_ERR_NO_DEVICE (-7)   /* no usable gfx950 device: there is NO CPU fallback                 */
#define KSSD_ERR_INPUT (-8)       /* a FASTA header is not closed before the end of the file            */
                                  /*   ("fasta header not closed", iseq2comem.c:233)                    */

/* ---- geometry of a packed batch -------------------------------------------------------------- */
#define KSSD_CHUNK_BASES 4096u   /* positions per chunk; every genome starts on a chunk boundary      */
#define KSSD_CHUNK_WORDS 256u    /* u32 words of packed bases per chunk (16 bases / word)              */
#define KSSD_CHUNK_MASKW 128u    /* u32 words of validity mask per chunk (32 positions / word)         */
#define KSSD_PACK_SLACK_WORDS 8u /* readable words required past the last chunk of `packed` AND `mask`   */
/*
 * packed : position p lives in word p/16, bits [31-2(p%16)-1, 31-2(p%16)]  (first base = top bits),
 *          code A=0 C=1 G=2 T=3  (global_basic.c:64-71)
 * mask   : position p lives in word p/32, bit p%32; 1 = an A/C/G/T base, 0 = anything that resets the
 *          reference's run counter (iseq2comem.c:221-242): N, IUPAC, header, record boundary, padding
 */

/* sketch flags */
#define KSSD_SKETCH_FASTA 0u        /* fasta2co semantics: id 0 is never stored (iseq2comem.c:258)     */
#define KSSD_SKETCH_KEEP_ZERO 1u    /* fastq2co semantics: id 0 is kept (iseq2comem.c:335-337)         */
#define KSSD_SKETCH_UNIQ 2u         /* uniq_fasta2co (-u): drop ids seen more than once (:694-696)     */
#define KSSD_SKETCH_NO_CAPACITY 4u  /* do not raise KSSD_ERR_CAPACITY (fastq2co never does, :338)      */
#define KSSD_SKETCH_FIRST_POS 8u    /* also report every id's first position inside its genome: what the */
                                    /*   reference's hash-slot file order depends on (iseq2comem.c:254-268) */
#define KSSD_SKETCH_COUNTS 16u      /* report every id's number of occurrences, saturating at 65535, in the   */
                                    /*   second output array instead of positions: the abundance sketches of   */
                                    /*   -A (mt_shortreads2koc / write_fqkoc2files, iseq2comem.c:435-471,552-615) */

#define KSSD_SKETCH_BY_POS 32u      /* no dedup at all: EVERY sampled k-mer of a genome, in sequence order, repeats  */
                                    /*   and id 0 included; the second output array holds its position.  The stream  */
                                    /*   reads2mco writes for dist --byread (iseq2comem.c:156-176); the caller cuts it  */
                                    /*   into reads at the positions its tokeniser recorded for the '>' lines          */

typedef struct kssd_gpu_ctx kssd_gpu_ctx;

/* mirrors dim_shuffle_stat_t (command_shuffle.h:17-23) */
typedef struct kssd_shuf_hdr {
    int32_t id, k, subk, drlevel;
} kssd_shuf_hdr;

/* derived constants, mirrors seq2co_global_var_initial (iseq2comem.c:54-77) + get_hashsz */
typedef struct kssd_gpu_info {
    int32_t k, subk, drlevel, kmerlen /*2k*/, dim_rd_len /*2*drlevel*/;
    int32_t comp_num, comp_bits;
    uint32_t dim_end, hashsize, hashlimit;
    int32_t device, cu_count;
} kssd_gpu_info;

const char *kssd_gpu_strerror(int code);
/* text of the last HIP error seen by this thread ("" if none) */
const char *kssd_gpu_last_hip_error(void);

/*
 * Replaces read_dim_shuffle_file + seq2co_global_var_initial (command_shuffle.c:192-207,
 * iseq2comem.c:54-77).  `table` = the .shuf permutation, HOST int32[16^subk].  Only the dim_end
 * sub-contexts with table[x] < dim_end are kept (device tables are a few hundred KB).
 */
int kssd_gpu_create(kssd_gpu_ctx **out, const kssd_shuf_hdr *hdr, const int32_t *table, int device);
/* same, from the compact form: accepted[r] = the sub-context x with table[x] == r, r < dim_end */
int kssd_gpu_create_compact(kssd_gpu_ctx **out, const kssd_shuf_hdr *hdr, const uint32_t *accepted,
                            uint32_t n_accepted, int device);
/* a context that can only index and compute distances (no .shuf needed): kmerlen = 2k of the sketches,
 * as stored in cofiles.stat / mcofiles.stat (command_dist.c:727) */
int kssd_gpu_create_for_dist(kssd_gpu_ctx **out, int kmerlen, int device);
void kssd_gpu_destroy(kssd_gpu_ctx *ctx);
int kssd_gpu_get_info(const kssd_gpu_ctx *ctx, kssd_gpu_info *info);

/*
 * Sketch a packed batch: replaces fasta2co / fastq2co + wrt_co2cmpn_use_inn_subctx / write_fqco2file
 * for every genome of the batch (iseq2comem.c:188-356,499-551; driver loop command_dist.c:277-312).
 *   d_packed, d_mask : DEVICE, n_chunks chunks (+ KSSD_PACK_SLACK_WORDS readable words after each)
 *   h_chunk_off      : HOST  u64[n_genomes+1], genome g owns chunks [h_chunk_off[g], h_chunk_off[g+1])
 *   min_occ          : keep an id only if it occurs >= min_occ times (fastq -n, iseq2comem.c:336-345)
 *   d_out_off        : DEVICE u64[n_genomes+1] exclusive prefix of sketch sizes
 *   d_out_ids        : DEVICE u32[out_cap]; genome g's ids ascending and distinct (the reference's
 *                      sketch is this set in hash-slot order; see kssd_host_slot_order)
 * Nothing is synchronised; call kssd_gpu_sketch_status afterwards.
 */
int kssd_gpu_sketch_device(kssd_gpu_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                           const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags,
                           uint32_t min_occ, uint64_t *d_out_off, uint32_t *d_out_ids,
                           uint64_t out_cap, void *stream);
/*
 * The same call, split for callers that stream batches through several contexts (one per batch in flight, each on
 * its own stream): kssd_gpu_sketch_plan does the host part (validation, workspace sizing; no stream work),
 * kssd_gpu_sketch_phase enqueues one phase.  Phases of one plan go on one stream, in order PREP, SCAN, EXACT, FINISH;
 * the caller may put event waits between them so that the phases without LDS use (PREP, EXACT) run underneath another
 * context's SCAN, and the LDS users (FINISH) between two scans (bench.py).  kssd_gpu_sketch_device = the plan and
 * the four phases back to back.
 */
#define KSSD_PHASE_PREP 0   /* per-call state, chunk -> genome map                         */
#define KSSD_PHASE_SCAN 1   /* the scan kernel: every CU's LDS                             */
#define KSSD_PHASE_EXACT 2  /* exact evaluation of the candidates: no LDS, random HBM reads */
#define KSSD_PHASE_FINISH 3 /* per-genome dedup (LDS sort), CSR offsets, gather            */
#define KSSD_PHASE_REPASS 4 /* instead of PREP + SCAN: the next tuple pass over the candidates of the last scan (below) */
int kssd_gpu_sketch_plan(kssd_gpu_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                         const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                         uint64_t *d_out_off, uint32_t *d_out_ids, uint64_t out_cap);
int kssd_gpu_sketch_phase(kssd_gpu_ctx *ctx, int phase, void *stream);

/*
 * The summary level above the validity mask (part of the packed batch layout since round 6; optional): one 64-bit word per
 * chunk; a SET bit l says "the 64 positions [64 l, 64 l + 64) of the chunk are all bases" -- i.e. the reference's run counter
 * (`base`, iseq2comem.c:213-243) is never reset inside them --, a clear bit promises nothing (the scan then looks at the lane's
 * mask words: kssd_gpu_mask_summarise_device sets every bit that can be set, the device tokeniser leaves the few runs of 64
 * positions clear that two of its 16 KiB groups share).  With it the scan reads 8 bytes per chunk instead of the chunk's
 * 512 bytes of mask and fetches the two mask words of the lanes whose bit is clear only (a lane with an N, a genome's last
 * lanes, its padding).  Results are identical with and without.
 *   kssd_gpu_mask_summarise_device  writes d_summary[n_chunks] (DEVICE) from d_mask (aligned as for the sketch calls: 16 bytes) on `stream`: what whoever makes a batch
 *                                   resident calls once (the device tokeniser's callers get it with the mask);
 *   kssd_gpu_sketch_set_mask_summary  names the summary of the d_mask of the NEXT kssd_gpu_sketch_plan / _sketch_device call
 *                                   of this context (that one call only; NULL / nothing set: the scan streams the mask).
 * A batch sketched with fastq2co semantics (KSSD_SKETCH_KEEP_ZERO: read sets, where four lanes in ten hold a read's end) is still
 * scanned with the streamed mask; its exact-evaluation stage settles a candidate's validity by the summary words where they answer.
 */
int kssd_gpu_mask_summarise_device(kssd_gpu_ctx *ctx, const uint32_t *d_mask, uint64_t n_chunks, uint64_t *d_summary, void *stream);
int kssd_gpu_sketch_set_mask_summary(kssd_gpu_ctx *ctx, const uint64_t *d_summary);

/*
 * Synchronises `stream` and reports on the last kssd_gpu_sketch_device call:
 *   *total_ids  = ids the batch produced (valid even on KSSD_ERR_OVERFLOW: size to retry with)
 *   *bad_genome = first genome that raised KSSD_ERR_CAPACITY, else -1
 * returns KSSD_OK, KSSD_ERR_OVERFLOW (out_cap or the staging regions too small; the latter are
 * grown automatically, just call again), KSSD_ERR_CAPACITY, KSSD_ERR_UNSUPPORTED or KSSD_ERR_HIP.
 */
int kssd_gpu_sketch_status(kssd_gpu_ctx *ctx, uint64_t *total_ids, int64_t *bad_genome, void *stream);

/*
 * Telemetry of the last kssd_gpu_sketch_device call (synchronises `stream`): how many window positions passed
 * the stage-1 group filter and how many of those passed the Bloom test of the exact pattern (= the candidates
 * handed to the exact stage).  Any pointer may be NULL.
 */
int kssd_gpu_scan_stats(kssd_gpu_ctx *ctx, uint64_t *stage1, uint64_t *bloom, void *stream);

/*
 * Where kssd_gpu_sketch_device writes the first positions when called with KSSD_SKETCH_FIRST_POS (the counts with
 * KSSD_SKETCH_COUNTS, the positions of the stream with KSSD_SKETCH_BY_POS): DEVICE u32[out_cap], parallel to
 * d_out_ids (position of the id's first occurrence, counted from the genome's first chunk).  Genomes must be
 * shorter than 2^32 positions in these modes.
 */
int kssd_gpu_sketch_set_pos_output(kssd_gpu_ctx *ctx, uint32_t *d_out_pos);

/*
 * Process start-up, no counterpart in the reference: initialises the HIP runtime, the device's context and this
 * library's code object on `device` (0.1 - 0.2 s per process on the measurement box).  Callable from any thread; a
 * command that has host work to do first (reading sketch directories, opening inputs) runs it on a thread of its own so
 * that the first real call does not pay for it.  KSSD_ERR_NO_DEVICE without a usable device.
 */
int kssd_gpu_warm_up(int device);

/*
 * Tuning knob of the per-genome dedup: genomes whose staging region holds more than max_tuples tuples are sorted
 * in global memory instead of in one workgroup's LDS (default and upper limit: 32 768 four-byte keys; 0 = default).
 * The results do not depend on it; the parity tests use it to send small genomes down the large-genome path.
 */
int kssd_gpu_set_lds_sort_limit(kssd_gpu_ctx *ctx, uint32_t max_tuples);

/*
 * Parameter sets with k - drlevel = 9 (e.g. -k 12 -L 3): the reduced tuple has 36 bits; the reference spreads it over
 * 16^(k - drlevel - 7) = 256 component files of 28-bit ids (iseq2comem.c:63-64,527,542-543; hash table: 4 GiB per thread).
 * The device keeps its 32-bit ids and works in kssd_gpu_tuple_passes() = 16 passes over the candidates of ONE scan: pass s
 * keeps the tuples whose low four bits are s, the id it writes is tuple >> 4.  Component file of such an id:
 * ((id & 15) << 4) | s, stored id: id >> 4; the id 0 rule applies to pass 0 alone; the capacity rule (hashlimit of distinct
 * tuples per genome) is per pass on the device -- the caller adds the passes' sketch sizes up.
 *   kssd_gpu_set_tuple_pass(ctx, 0); kssd_gpu_sketch_plan(..out buffers of pass 0..); phases PREP, SCAN, EXACT, FINISH
 *   for s = 1 .. 15: kssd_gpu_set_tuple_pass(ctx, s); kssd_gpu_sketch_plan(..same batch, out buffers of pass s..);
 *                    phases REPASS, EXACT, FINISH
 * Every other parameter set has one pass and nothing changes (kssd_gpu_tuple_passes() = 1).
 */
uint32_t kssd_gpu_tuple_passes(const kssd_gpu_ctx *ctx);
int kssd_gpu_set_tuple_pass(kssd_gpu_ctx *ctx, uint32_t pass);
/* host level: the batch of the context's LAST host-level sketch call (kssd_gpu_sketch_batch[_pos], kssd_gpu_sketch_fast[aq]_text)
 * once more, without scanning it again: the passes 1 .. 15 after kssd_gpu_set_tuple_pass.  Same flags / min_occ / out_pos
 * as that call; results as there (kssd_gpu_free). */
int kssd_gpu_sketch_again(kssd_gpu_ctx *ctx, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                          uint32_t **out_pos, int64_t *bad_genome);

/*
 * Tuning knob of the scan: at most max_workgroups workgroups of 16 waves (0 = default, one per compute unit).  Every wave
 * owns one contiguous run of the batch's chunks, so fewer workgroups mean longer runs per wave.  The results do not
 * depend on it; the parity tests use it to give a wave more than 2 048 chunks at a size the oracle checks in seconds
 * (what a 34 GB read set or a batch of mammalian genomes does on the full grid).
 */
int kssd_gpu_set_scan_grid(kssd_gpu_ctx *ctx, uint32_t max_workgroups);

/* host-level convenience: HOST packed/mask in, malloc'd HOST CSR out (free with kssd_gpu_free) */
int kssd_gpu_sketch_batch(kssd_gpu_ctx *ctx, const uint32_t *packed, const uint32_t *mask,
                          const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags,
                          uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                          int64_t *bad_genome);
/* same, plus the first position of every id (KSSD_SKETCH_FIRST_POS is added to flags) or, with KSSD_SKETCH_COUNTS
 * in flags, its number of occurrences or, with KSSD_SKETCH_BY_POS, the position of every entry of the k-mer stream */
int kssd_gpu_sketch_batch_pos(kssd_gpu_ctx *ctx, const uint32_t *packed, const uint32_t *mask,
                              const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags,
                              uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids, uint32_t **out_pos,
                              int64_t *bad_genome);
void kssd_gpu_free(void *p);

/*
 * The FASTA tokeniser on the device: the byte rules of fasta2co (iseq2comem.c:213-242: bases, transparent line ends,
 * headers skipped to their line end, every other byte breaks the run) applied to raw file bytes in HBM.
 *   d_text      DEVICE bytes; file f occupies [h_text_off[f], h_text_off[f] + h_text_len[f]), offsets multiples of 16
 *   h_chunk_off HOST u64[n_files + 1]: genome f gets chunks [h_chunk_off[f], h_chunk_off[f+1]), at least
 *               ceil(h_text_len[f] / 4096) of them (an input of B bytes never emits more than B positions)
 *   d_packed / d_mask   DEVICE, the batch layout above (+ KSSD_PACK_SLACK_WORDS); zeroed and written here
 * Nothing is synchronised.  kssd_gpu_tokenise_status synchronises `stream` and returns KSSD_ERR_INPUT with the index of
 * the first file whose last header is not closed (the host tokeniser's KSSD_HOST_ERR_HEADER), else KSSD_OK;
 * h_positions (HOST u64[n_files], may be NULL) receives every file's positions (bases + run breaks).
 * The batch is bit for bit what libkssd_host.so's kssd_batch_fill_text writes for the same bytes.
 */
int kssd_gpu_tokenise_fasta_device(kssd_gpu_ctx *ctx, const uint8_t *d_text, const uint64_t *h_text_off, const uint64_t *h_text_len,
                                   uint32_t n_files, uint32_t *d_packed, uint32_t *d_mask, const uint64_t *h_chunk_off, void *stream);
int kssd_gpu_tokenise_status(kssd_gpu_ctx *ctx, int64_t *bad_file, uint64_t *h_positions, void *stream);
/* the same, and the mask's summary words (kssd_gpu_sketch_set_mask_summary) written beside the mask: d_summary = DEVICE
 * u64[h_chunk_off[n_files]].  kssd_gpu_sketch_fasta_text does this by itself and scans with them. */
int kssd_gpu_tokenise_fasta_device_summary(kssd_gpu_ctx *ctx, const uint8_t *d_text, const uint64_t *h_text_off, const uint64_t *h_text_len,
                                           uint32_t n_files, uint32_t *d_packed, uint32_t *d_mask, uint64_t *d_summary,
                                           const uint64_t *h_chunk_off, void *stream);
/*
 * host-level: FASTA texts in HOST memory (page-locked for full PCIe speed) -> sketches, tokenised on the device.  One
 * genome per file, layout as above; out_pos may be NULL (then no first positions / counts / stream positions are
 * returned), otherwise as kssd_gpu_sketch_batch_pos.  KSSD_ERR_INPUT: *bad_genome = the file with the unclosed header.
 */
int kssd_gpu_sketch_fasta_text(kssd_gpu_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_len,
                               uint32_t n_files, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                               uint32_t **out_pos, int64_t *bad_genome);
/*
 * The quality floor of the FASTQ calls below (fastq2co's -Q, iseq2comem.c:312: a base counts only if the byte in the same
 * column of the record's fourth line is >= min_quality; 0 = none, the default; at most 127).  Stays set on the context.
 * A file whose quality lines are shorter than their bases is handed back like the other cases only the reference's own
 * fgets() sequence reproduces.
 */
int kssd_gpu_set_fastq_quality(kssd_gpu_ctx *ctx, int min_quality);

/*
 * on != 0: the FASTQ calls below frame their input as mt_shortreads2koc does (dist -A, iseq2comem.c:552-615): records of
 * four lines, the second one scanned, no quality floor, line buffer of 4 096 bytes.  On complete files of ordinary reads
 * that is fastq2co's framing; what differs -- a file that does not end with the line end of a complete record, a line of
 * 4 000 bytes or more -- is handed back to the host tokeniser (kssd_batch_fill_text, kind 2).  The "lines" output of such
 * a call is 4 x the records scanned.  Stays set on the context.
 */
int kssd_gpu_set_fastq_reads(kssd_gpu_ctx *ctx, int on);

/*
 * The same two for FASTQ read sets as fastq2co reads them with -Q 0 (iseq2comem.c:274-330: records of four lines, only
 * the second one scanned, a read never continues the k-mer of the read in front of it, a final record its four lines
 * do not complete is not scanned).  An input the device cannot do exactly as the reference -- no complete record, a line
 * of 19 000 bytes or more (the reference's fgets() buffer splits lines at 19 999), a NUL or a byte >= 0x80 -- is handed
 * back: kssd_gpu_tokenise_fastq_status / kssd_gpu_sketch_fastq_text return KSSD_ERR_UNSUPPORTED with its index, and the
 * caller runs libkssd_host.so's kssd_batch_add_fastq for it.  h_lines (HOST u64[n_files], may be NULL) receives the
 * line count the reference reports per file (4 x complete records).  The quality floor and the dist -A framing: kssd_gpu_set_fastq_quality / kssd_gpu_set_fastq_reads above.
 */
int kssd_gpu_tokenise_fastq_device(kssd_gpu_ctx *ctx, const uint8_t *d_text, const uint64_t *h_text_off, const uint64_t *h_text_len,
                                   uint32_t n_files, uint32_t *d_packed, uint32_t *d_mask, const uint64_t *h_chunk_off, void *stream);
int kssd_gpu_tokenise_fastq_status(kssd_gpu_ctx *ctx, int64_t *bad_file, uint64_t *h_positions, uint64_t *h_lines, void *stream);
int kssd_gpu_sketch_fastq_text(kssd_gpu_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_len,
                               uint32_t n_files, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                               uint32_t **out_pos, uint64_t *h_lines, int64_t *bad_genome);
/*
 * dist --byread (reads2mco, iseq2comem.c:78-186): where the reads of FASTA file `file` of the batch this context tokenised
 * LAST (kssd_gpu_sketch_fasta_text / kssd_gpu_tokenise_fasta_device, nothing tokenised in between) begin in the file's
 * position stream -- at every '>' met outside a header the position the next base gets, exactly what libkssd_host.so's
 * kssd_batch_add_fasta_reads returns.  read_start: malloc'd HOST u64[*n_reads], ascending (kssd_gpu_free).  Together with
 * the KSSD_SKETCH_BY_POS stream of the same call this is all kssd_byread_write needs.
 */
int kssd_gpu_fasta_read_starts(kssd_gpu_ctx *ctx, uint32_t file, uint64_t **read_start, uint64_t *n_reads);
/*
 * Streaming a long input in without a host buffer of its size: reserve the context's device text buffer (reserving more
 * later keeps what has been put: an input of unknown size -- a gzip'ed file -- starts from an estimate), copy
 * slices in from page-locked host memory (asynchronous, on the context's own stream; the returned ticket >= 0 tells
 * kssd_gpu_text_wait which copy to wait for before the slice's memory is refilled -- at most the last 32 copies are
 * tracked one by one), then call kssd_gpu_sketch_fasta_text / _fastq_text with text == NULL: the offsets then address
 * the device buffer.  (The reference reads a file through a FILE* / zcat pipe line by line, iseq2comem.c:196-330; this is
 * what replaces that for inputs of many gigabytes.)
 */
int kssd_gpu_text_reserve(kssd_gpu_ctx *ctx, uint64_t bytes);
int64_t kssd_gpu_text_put(kssd_gpu_ctx *ctx, uint64_t dst_off, const void *src, uint64_t n);
int kssd_gpu_text_wait(kssd_gpu_ctx *ctx, int64_t ticket);



/*
 * Page-locked host memory (hipHostMalloc): a batch tokenised into it (kssd_batch_create_ex of the host library takes
 * these two as its allocator) goes to the device by DMA at PCIe speed, without the runtime's staging copy.  The
 * reference reads its inputs through popen("zcat") into a 64 KiB buffer (iseq2comem.c:196-208); this is that buffer's
 * counterpart.  kssd_gpu_host_alloc returns NULL when the memory cannot be had.
 */
void *kssd_gpu_host_alloc(size_t bytes);
void kssd_gpu_host_free(void *p);
/* Memory of the caller's own (malloc) page-locked in place, and released again (before it is freed): for what a command read into
 * ordinary memory while the runtime was still starting -- copies out of unregistered memory run at a fraction of the PCIe rate. */
int kssd_gpu_host_register(void *p, size_t bytes);
int kssd_gpu_host_unregister(void *p);

/*
 * Build the inverted index of the reference sketches on the device: replaces combco2mco
 * (co2mco.c:25-77).  CSR input, DEVICE pointers (as kssd_gpu_sketch_device writes them; any id order is
 * accepted).  max_ref_ids is only an upper bound of d_roff[n_ref] -- the TOTAL number of reference ids; it sizes the
 * index's arrays -- so no host synchronisation is needed between sketching and indexing.  A bound that turns out too
 * small is not followed behind those arrays: the index then holds the first max_ref_ids entries and
 * kssd_gpu_index_status returns KSSD_ERR_PARAM.  The index lives in ctx until the next call.
 */
int kssd_gpu_index_build_device(kssd_gpu_ctx *ctx, const uint64_t *d_roff, const uint32_t *d_rids,
                                uint32_t n_ref, uint64_t max_ref_ids, void *stream);
/*
 * Synchronises `stream` and reports on the last kssd_gpu_index_build_device -- the counterpart of kssd_gpu_sketch_status.
 * The ordinary build gives every bucket of the hash the same room (no counting pass: two launches, and the search finds a
 * bucket's slots from its number alone); ids that do not spread over the buckets -- crafted ones, a database of hundreds of
 * near-identical genomes -- overflow a bucket: KSSD_ERR_OVERFLOW, the index in place is not whole (kssd_gpu_dist_device
 * computes nothing on it), and every later build of this context counts first (the exact build: four launches, one
 * descriptor per bucket).  Call kssd_gpu_index_build_device again.  The host-level searches do all of this themselves.
 * kssd_gpu_index_set_exact(ctx, 1) asks for the exact build up front (a tuning / test knob: results do not depend on it).
 */
int kssd_gpu_index_status(kssd_gpu_ctx *ctx, void *stream);
int kssd_gpu_index_set_exact(kssd_gpu_ctx *ctx, int exact);

/*
 * Tuning knob for searches whose query rows mostly MISS the index (the foreign rows of the multi-GPU all-pairs
 * partition: every rank's sketches against the own index): the next kssd_gpu_index_build_device also builds a blocked
 * Bloom filter of the indexed ids (~2 MB per 1.2 M ids, L2 resident), and kssd_gpu_dist_device consults it before it
 * walks the table -- except for query rows [skip_row_begin, skip_row_end), which are expected to hit (the rank's own
 * rows).  The results do not depend on it.  Off by default: on rows that hit it only costs.
 */
int kssd_gpu_index_set_filter(kssd_gpu_ctx *ctx, int enable, uint32_t skip_row_begin, uint32_t skip_row_end);

/*
 * Shared-k-mer counts and distances for query rows [q_begin, q_end) against the indexed references:
 * replaces the hot loop of mco_cbdco_nobin_dist (command_dist.c:763-790) and the arithmetic of
 * output_ctrl (command_dist.c:1251-1266, no --correction).  All pointers DEVICE; the five outputs are
 * row-major [(q_end-q_begin) x n_ref], any of the four f64 planes may be NULL.
 *   J = s/(X+Y-s)  MashD = min(1, ln(1/(2J)+0.5)/kmerlen)  C = s/min(X,Y)  AafD = min(1, ln(1/C)/kmerlen)
 * with X=|ref|, Y=|qry|, s=shared.
 */
int kssd_gpu_dist_device(kssd_gpu_ctx *ctx, const uint64_t *d_qoff, const uint32_t *d_qids,
                         uint32_t n_qry, uint32_t q_begin, uint32_t q_end, uint32_t *d_shared,
                         double *d_jaccard, double *d_mashd, double *d_contain, double *d_aafd,
                         void *stream);

/*
 * The same for query rows that may be very long (a read set sketched as ONE genome holds hundreds of thousands of ids):
 * max_row_ids = an upper bound of the ids of the longest row in [q_begin, q_end); beyond 16 384 the row's ids are shared
 * by several workgroups (the rows are zeroed and summed with atomics, the workgroup that arrives last computes the
 * metrics).  The results are the same.
 */
int kssd_gpu_dist_device_long(kssd_gpu_ctx *ctx, const uint64_t *d_qoff, const uint32_t *d_qids, uint32_t n_qry,
                              uint32_t q_begin, uint32_t q_end, uint64_t max_row_ids, uint32_t *d_shared,
                              double *d_jaccard, double *d_mashd, double *d_contain, double *d_aafd, void *stream);

/*
 * The same rows with the block written TRANSPOSED: for an all-pairs run over several devices in which a device indexes only
 * the sketches it made and runs everybody's sketches as query rows.  What that device owns of the matrix
 * are the rows of its OWN genomes (one owner per output row, command_dist.c:774-785) -- the transpose of what the rows kernel
 * counts, and every metric of the path is symmetric in (X, Y).  d_work: DEVICE u32[(q_end - q_begin) x n_ref] scratch (the
 * counts row-major by query); the five outputs receive, for indexed sketch r and query q, element r * out_pitch + (q - q_begin)
 * (out_pitch >= q_end - q_begin, in elements; planes may be NULL).  The bits of a pair are those kssd_gpu_dist_device writes.
 */
int kssd_gpu_dist_device_transposed(kssd_gpu_ctx *ctx, const uint64_t *d_qoff, const uint32_t *d_qids, uint32_t n_qry,
                                    uint32_t q_begin, uint32_t q_end, uint32_t *d_work, uint64_t out_pitch,
                                    uint32_t *d_shared_t, double *d_jaccard_t, double *d_mashd_t, double *d_contain_t,
                                    double *d_aafd_t, void *stream);

/*
 * The two halves of kssd_gpu_dist_device_transposed on their own, for a caller that computes its query rows in several calls
 * (its own rows while the exchange is still under way, the others' behind it) and turns all of them around at once:
 *   kssd_gpu_dist_counts_device        shared counts only, row-major by query: row q of [q_begin, q_end) at
 *                                      d_counts[(q - q_begin) * n_ref ..] (kssd_gpu_dist_device without planes).
 *   kssd_gpu_transpose_metrics_device  d_counts (rows [q_begin, q_end) as above) -> the five outputs, element (indexed sketch r,
 *                                      query q) at r * out_pitch + (q - q_begin); |query| from d_qoff.
 */
int kssd_gpu_dist_counts_device(kssd_gpu_ctx *ctx, const uint64_t *d_qoff, const uint32_t *d_qids, uint32_t n_qry, uint32_t q_begin,
                                uint32_t q_end, uint32_t *d_counts, void *stream);
int kssd_gpu_transpose_metrics_device(kssd_gpu_ctx *ctx, const uint64_t *d_qoff, uint32_t n_qry, uint32_t q_begin, uint32_t q_end,
                                      const uint32_t *d_counts, uint64_t out_pitch, uint32_t *d_shared_t, double *d_jaccard_t,
                                      double *d_mashd_t, double *d_contain_t, double *d_aafd_t, void *stream);

/* host-level convenience: HOST CSR in, HOST matrices out (caller-allocated, Q x R; planes may be NULL) */
int kssd_gpu_dist(kssd_gpu_ctx *ctx, const uint64_t *roff, const uint32_t *rids, uint32_t n_ref,
                  const uint64_t *qoff, const uint32_t *qids, uint32_t n_qry, uint32_t *shared,
                  double *jaccard, double *mashd, double *contain, double *aafd);

/*
 * The search with the report's selection done on the device (dist_print_nobin + output_ctrl, command_dist.c:1196-1267):
 * returns, per query row, the references that CAN appear in distance.out under -M metric (0 Jaccard / 1 containment),
 * --correction, -D dthreshold and -N n_max -- with -N every pair with a positive metric (a metric of 0 never enters the
 * reference's best-N list, :1212-1227), otherwise every pair whose distance, computed on the device, does not exceed
 * the threshold by more than a margin that covers any difference to the host's libm.  The host applies the exact -N
 * ranking and -D test to these candidates and formats only them (kssd_distance_print_pairs of the host library): the
 * text is the dense report's byte for byte, the formatting work shrinks from Q x R lines to the pairs that matter.
 *   shared       HOST u32[Q x R] or NULL: the dense counts as well (sharedk_ct.dat, --keepskf)
 *   pair_off     malloc'd HOST u64[n_qry + 1]; pair_ref / pair_shared: malloc'd HOST u32[pair_off[n_qry]], references
 *                ascending inside a row (free all three with kssd_gpu_free)
 * dim_rd_len = 2 * drlevel as stored in cofiles.stat (only --correction uses it).
 */
int kssd_gpu_dist_select(kssd_gpu_ctx *ctx, const uint64_t *roff, const uint32_t *rids, uint32_t n_ref, const uint64_t *qoff,
                         const uint32_t *qids, uint32_t n_qry, int metric, int correction, int dim_rd_len, double dthreshold,
                         int n_max, uint32_t *shared, uint64_t **pair_off, uint32_t **pair_ref, uint32_t **pair_shared);

/*
 * The same search on several devices: the query rows are cut into n_devices contiguous blocks (the reference gives
 * every output row one owner thread, command_dist.c:774-785), every device of `devices` receives the whole reference
 * CSR, builds its own index and writes its block of rows into the caller's matrices.  No exchange between the devices
 * (the sketches come from the host).  One host thread per entry; an entry may name a device twice.  kmerlen = 2k.
 */
int kssd_gpu_dist_multi(const int *devices, int n_devices, int kmerlen, const uint64_t *roff, const uint32_t *rids,
                        uint32_t n_ref, const uint64_t *qoff, const uint32_t *qids, uint32_t n_qry, uint32_t *shared,
                        double *jaccard, double *mashd, double *contain, double *aafd);
/*
 * The unpacking side of the one exchange step of the multi-GPU path (one process per GPU, an all-gather of every rank's
 * sketches over RCCL): each of `world` ranks contributed a fixed-size unit -- n_per_unit + 1 offsets (u64, exclusive
 * prefix of its sketch sizes) and `cap` id slots (u32, the first off[n_per_unit] meaningful).  d_off_all
 * [world x (n_per_unit + 1)] and d_ids_all [world x cap] are the gathered units; the call writes the CSR of all
 * world x n_per_unit sketches: d_roff [world * n_per_unit + 1], d_rids [up to world x cap] (genome r * n_per_unit + g of
 * the result is genome g of rank r).  DEVICE pointers, nothing is synchronised.
 */
int kssd_gpu_concat_units_device(kssd_gpu_ctx *ctx, const uint64_t *d_off_all, const uint32_t *d_ids_all, uint32_t world,
                                 uint32_t n_per_unit, uint64_t cap, uint64_t *d_roff, uint32_t *d_rids, void *stream);
/*
 * The exchange itself inside ONE process that drives n devices (no counterpart in the reference: its threads share one
 * address space; here the reference sketches are produced on different devices and every device's rows need all of
 * them): an all-gather over RCCL -- xGMI between the GPUs of a node -- of every device's unit, then the unpacking above
 * on every device.  ctxs[i] lives on its own device (one rank per device); d_off_l[i] = u64[n_per_rank + 1] and
 * d_ids_l[i] = u32[unit_ids] are device i's unit, d_roff[i] (u64[n * n_per_rank + 1]) and d_rids[i] (u32[n * unit_ids])
 * receive the CSR of all n * n_per_rank sketches on device i; streams[i] (NULL: the null stream) orders the work of
 * device i.  Nothing is synchronised.  librccl.so is loaded by the first call (dlopen), the communicators are kept and
 * remade when the device list changes.  KSSD_ERR_HIP with the RCCL error text in kssd_gpu_last_hip_error().
 */
int kssd_gpu_allgather_sketches(kssd_gpu_ctx *const *ctxs, int n, const uint64_t *const *d_off_l, const uint32_t *const *d_ids_l,
                                uint32_t n_per_rank, uint64_t unit_ids, uint64_t *const *d_roff, uint32_t *const *d_rids,
                                void *const *streams);
/*
 * Stage I, the exchange and the search as ONE flow in which the sketches never leave the devices that made them: what the
 * reference's `dist` does when it runs stage I, stage II and the search back to back (dist_dispatch, command_dist.c:64-99;
 * mco_cbdco_nobin_dist :670-808, one owner thread per output row :774-785) -- there through files every thread can map,
 * here with devices as the owners of rows.
 *   kssd_gpu_resident_create    one object per device: `n_slots` genome slots.  Touches no device.
 *   kssd_gpu_resident_put       slots [first_slot, first_slot + n) := the n genomes of ctx's LAST host-level sketch call
 *                               (kssd_gpu_sketch_batch[_pos], kssd_gpu_sketch_fast[aq]_text, kssd_gpu_sketch_again; not a
 *                               KSSD_SKETCH_BY_POS stream): their ids as that call left them on the device, copied device
 *                               to device.  ctx lives on the object's device; several threads (the sketch workers of one
 *                               device) may put into one object, each slot once.
 *   kssd_gpu_resident_put_host  the same from host arrays (the modes whose keep rule the host replays: -u, fastq -n > 1)
 *   kssd_gpu_resident_sizes     sizes[n_slots]: the sketch size of every slot (0xFFFFFFFF: never put)
 *   kssd_gpu_resident_allpairs  all-pairs among the genomes of `sets` (global numbering: set 0's slots, then set 1's, ...;
 *                               every set but the last holds the same number of slots -- kssd_shard_plan of the host
 *                               library deals the inputs out that way): ONE RCCL all-gather of every device's packed
 *                               sketches (one device: its unit is unpacked in place, or -- with KSSD_EXCHANGE_ONE_RANK=1 in
 *                               the environment -- sent through a one-rank communicator); then every device indexes the
 *                               sketches IT made, runs everybody's as query rows and writes the transpose -- the rows of its
 *                               own genomes (kssd_gpu_dist_device_transposed: queries = references, every metric symmetric;
 *                               the index build stays constant per device).  KSSD_ALLPAIRS_FULL_INDEX=1: the full index on
 *                               every device and its own genomes as query rows instead (also taken when a sketch is a
 *                               read set of more than 16 384 ids).  Results into the caller's HOST matrices (N x N
 *                               row-major; `shared` may be a mapped sharedk_ct.dat, the f64 planes may be NULL).  One host
 *                               thread per device: it unpacks, indexes and computes behind the device's share of the
 *                               collective on the device's stream.  KSSD_ERR_PARAM when a device is named twice (RCCL: one
 *                               rank per device) or a slot was never put.  No kssd_gpu_resident_put* may be running on any
 *                               of the sets (join the workers first).  KSSD_EXCHANGE_FAKE_RANKS=n: development -- the sets
 *                               may share a device and the collective is replaced by device-to-device copies of the bytes
 *                               it delivers, everything else runs as on n devices.  KSSD_TIMING: one JSON line on stderr
 *                               (pack, communicator set-up, exchange, unpack + index + rows).
 */
typedef struct kssd_gpu_resident kssd_gpu_resident;
int kssd_gpu_resident_create(kssd_gpu_resident **out, int device, uint32_t n_slots);
void kssd_gpu_resident_destroy(kssd_gpu_resident *r);
int kssd_gpu_resident_put(kssd_gpu_resident *r, kssd_gpu_ctx *ctx, uint32_t first_slot, uint32_t n);
int kssd_gpu_resident_put_host(kssd_gpu_resident *r, uint32_t first_slot, uint32_t n, const uint64_t *off, const uint32_t *ids);
int kssd_gpu_resident_sizes(const kssd_gpu_resident *r, uint32_t *sizes);
int kssd_gpu_resident_allpairs(kssd_gpu_resident *const *sets, int n_sets, int kmerlen, uint32_t *shared, double *jaccard,
                               double *mashd, double *contain, double *aafd);
/* Loads librccl.so and sets the communicators of a device list up ahead of the first kssd_gpu_allgather_sketches /
 * kssd_gpu_resident_allpairs over that list (one to two seconds: a command runs it on a thread of its own while its devices
 * sketch).  KSSD_ERR_PARAM for a list with a repeat, KSSD_ERR_HIP with the RCCL text otherwise. */
int kssd_gpu_exchange_warm_up(const int *devices, int n);
/* which == 0: the file of the HIP runtime this library is bound to; 1: the RCCL the first exchange loaded ("" before it).
 * For the line a multi-GPU run prints about itself; the string is the calling thread's until its next call. */
const char *kssd_gpu_runtime_path(int which);
/* how many gfx950 devices this process sees (0 without any; never an error) */
int kssd_gpu_device_count(void);

/*
 * Set operations on sketches of ONE component (ids below 16^7): replace the 2^28-bit dictionary walks of
 * `kssd set` (command_set.c).  HOST pointers in, malloc'd HOST arrays out (free with kssd_gpu_free).
 *   kssd_gpu_set_union   -u (sketch_union, :226-288): the distinct ids of `ids`, ascending; uniq != 0: -q
 *                        (uniq_sketch_union, :376-443): only the ids that occur exactly once in `ids`
 *   kssd_gpu_set_filter  -s / -i (sketch_operate, :289-375): every sketch of the CSR without (keep_members = 0) or
 *                        restricted to (keep_members = 1) the ids of `pan`; the order inside a sketch is kept
 * A ctx from kssd_gpu_create_for_dist is enough.
 */
int kssd_gpu_set_union(kssd_gpu_ctx *ctx, const uint32_t *ids, uint64_t n, int uniq, uint32_t **out_ids, uint64_t *out_n);
int kssd_gpu_set_filter(kssd_gpu_ctx *ctx, const uint64_t *off, const uint32_t *ids, uint32_t n_genomes,
                        const uint32_t *pan, uint64_t n_pan, int keep_members, uint64_t **out_off, uint32_t **out_ids);

/*
 * Timing hook for bench.py: every launch of the dominant kernel of a path is bracketed by HIP events on
 * the caller's stream (a ring of the last 128 launches).  which: 0 = sketch scan, 1 = distance rows, 2 / 3 = the FASTA
 * tokeniser's two passes over the text (tok_summarise, tok_emit).
 * Waits for the recorded events, returns their average in milliseconds and how many launches that covers;
 * reset != 0 empties the ring.
 */
int kssd_gpu_kernel_time(kssd_gpu_ctx *ctx, int which, int reset, float *avg_ms, uint32_t *launches);
/* the same launches one by one: ms[0 .. min(*launches, cap)) receive their durations (for a minimum / maximum beside the mean) */
int kssd_gpu_kernel_times(kssd_gpu_ctx *ctx, int which, float *ms, uint32_t cap, uint32_t *launches);
/*
 * How many launches carry the events: every `every`-th one of each path from this call on (1: all of them, the default;
 * 0: none).  A bracketed dispatch does not overlap its neighbours in the stream -- it waits for the kernel in front to drain
 * and the kernel behind waits for its time stamp --, which costs a step of six short kernels several microseconds per bracket
 * (profiles/r04E_gaps.txt); a caller that streams steps measures a sample of its launches instead of all.
 */
int kssd_gpu_set_kernel_timing(kssd_gpu_ctx *ctx, uint32_t every);

#ifdef __cplusplus
}
#endif
#endif
