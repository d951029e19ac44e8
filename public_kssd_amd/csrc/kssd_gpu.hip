// kssd_gpu.hip -- gfx950 (MI355X / CDNA4) kernels and the C ABI of libkssd_gpu.so.
//
// Hot path of kssd: genome sketching (reference iseq2comem.c:188-356) and the
// shared-k-mer intersection + distances (co2mco.c:25-77, command_dist.c:763-790,1251-1266).
// Integer / indexing work: HBM streaming, LDS tables, wavefront ballot/scan.  No MFMA on purpose.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC kssd_gpu.hip -o libkssd_gpu.so
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/kssd_gpu.h"
#include "kssd_core.h"
#include "kssd_dev.h"

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local char g_hip_err[256] = "";

static int hip_fail(hipError_t e, const char *what, int line)
{
    snprintf(g_hip_err, sizeof g_hip_err, "%s (line %d): %s", what, line, hipGetErrorString(e));
    return KSSD_ERR_HIP;
}
#define HIPCK(x)                                                \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) return hip_fail(e_, #x, __LINE__); \
    } while (0)

extern "C" const char *kssd_gpu_last_hip_error(void) { return g_hip_err; }

extern "C" const char *kssd_gpu_strerror(int code)
{
    switch (code) {
    case KSSD_OK: return "ok";
    case KSSD_ERR_HIP: return g_hip_err[0] ? g_hip_err : "HIP runtime error";
    case KSSD_ERR_PARAM: return "k / subk / drlevel rejected (same limits as the reference)";
    case KSSD_ERR_CAPACITY: return "the context space is too crowd, try rerun the program using a larger -k";
    case KSSD_ERR_OVERFLOW: return "output or staging buffer too small; call again";
    case KSSD_ERR_UNSUPPORTED: return "parameters accepted by the reference but not by the device path";
    case KSSD_ERR_NOMEM: return "out of memory";
    case KSSD_ERR_NO_DEVICE: return "no gfx950 device available (there is no CPU fallback)";
    case KSSD_ERR_INPUT: return "malformed input: a FASTA header is not closed before the end of the file";
    default: return "unknown kssd_gpu error";
    }
}

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
#define SCAN_THREADS 1024
#define SCAN_WAVES (SCAN_THREADS / 64)
#define CBUF 256          // per-wave buffer of stage-1 candidates waiting for their Bloom round: 4-byte position entries
#define SKETCH_TRACK_FILL 0x80000000u  // internal flag: record the fullest staging region even without an overflow
#define DEDUP_THREADS 512
#define DEDUP_MAX_N 32768 // ids one workgroup can sort in LDS (128 KiB)
#define EV_RING 128
#define EV_KINDS 4  // 0 = sketch scan, 1 = distance rows, 2 = tok_summarise, 3 = tok_emit
#define SCAN_TAB_BYTES (KSSD_T1_BYTES + KSSD_BLOOM_WORDS * 4)  // stage-1 table + stage-1.5 Bloom filter, contiguous
#define SCAN_LDS_BYTES (SCAN_TAB_BYTES + SCAN_WAVES * CBUF * 4)

struct SketchStatus {
    unsigned long long total_ids;
    unsigned int region_overflow;  // some genome emitted more than its staging region holds
    unsigned int out_overflow;     // d_out_ids too small
    unsigned int max_need_q8;      // max over genomes of emitted / capacity, in 1/256 units
    unsigned int capacity_genome_p1;  // 1 + first genome over the reference's hash limit (0 = none)
    // telemetry of the last scan: positions that passed stage 1, that passed the Bloom test
    unsigned long long n_stage1, n_bloom;
    unsigned int cand_overflow;    // a shard of the candidate list was too small
    unsigned int cand_need;        // entries the fullest shard wanted
    unsigned int lb_timeout;       // a workgroup of the per-genome kernel gave up waiting for the kept counts in front of it (kssd_dedup.inc: dedup_direct_out)
    unsigned int ranges_skew;      // a large genome's keys do not spread over its id ranges (one id tens of thousands of times, crafted
                                   // ids): the call is repeated with the global-memory sort for large genomes
    unsigned int parts_skew;       // a genome sorted in parts has more keys in ONE id range than the LDS sort takes although its region held
                                   // them all (a tandem repeat: hundreds of occurrences of a handful of ids): the call is repeated with such
                                   // genomes on the global-memory path -- larger regions would not spread them
};

struct kssd_gpu_ctx {
    int device;
    int cu_count;
    hipStream_t own_stream;  // non-blocking stream of the host-level calls (contexts on different host threads do not meet on the null stream)
    bool dist_only;  // created by kssd_gpu_create_for_dist: no sketch tables
    KssdParams P;
    uint8_t *d_T1;
    KssdG *d_G;
    uint32_t *d_gfilt;      // bit filter of the accepted sub-contexts (behind d_G: kssd_gfilt_bit)
    // sketch workspace
    uint32_t *d_chunk_gid;
    size_t cap_chunks;
    uint64_t *d_chunk_off;  // n_genomes+1
    uint64_t *d_reg_off;    // n_genomes+1
    uint32_t *d_cursor;     // n_genomes
    uint32_t *d_kept;       // n_genomes
    size_t cap_chunk_off, cap_reg_off, cap_cursor, cap_kept;
    uint32_t *d_regions;    // staging regions: u32 tuples, or u64 (tuple << 32 | position) in first-position mode
    size_t cap_regions;
    uint32_t *d_out_pos;    // first-position output of the next KSSD_SKETCH_FIRST_POS call (caller's buffer)
    const uint64_t *d_next_summ = nullptr;  // the summary words of the NEXT planned batch's mask (kssd_gpu_sketch_set_mask_summary)
    bool summ_auto = false;        // KSSD_MASK_SUMMARY=1: the host-level calls summarise their resident batch themselves (tests, fuzzers: the
                                   // summary costs a pass over the mask, which a batch that is scanned once does not earn back)
    uint64_t *d_in_summ = nullptr; // ... into this
    size_t cap_in_summ = 0;
    bool in_summ_valid = false;    // d_in_summ holds the summary words of the resident batch already (the one-pass tokeniser wrote them)
    uint64_t *d_cand;       // candidate list of the last scan (one slice per scan wave)
    size_t cap_cand;
    uint32_t *d_cand_count;
    size_t cap_cand_count;
    unsigned long long *d_blk_info;  // per block of the last scan: where its candidates are (scan_blk_pack)
    size_t cap_blk_info;
    uint64_t last_cand_cap;
    uint64_t cand_floor;    // per-slice capacity an overflowed attempt asked for (kept for the retries)
    double cand_factor;
    SketchStatus *d_status;
    double region_factor;
    uint32_t last_n_genomes;
    int last_launch_rc;
    uint32_t lds_sort_limit;  // kssd_gpu_set_lds_sort_limit (0 = DEDUP_MAX_N)
    uint32_t scan_grid_limit; // kssd_gpu_set_scan_grid (0 = one workgroup per CU)
    std::vector<uint64_t> h_reg_off;
    std::vector<uint32_t> h_big;  // genomes of the last batch that take the global-memory dedup path
    std::vector<uint2> h_med, dev_med;  // ... that are sorted in LDS in parts (genome, log2 parts): planned / what d_med holds
    uint2 *d_med = nullptr;
    void *d_med_out = nullptr;
    uint32_t *d_med_cnt = nullptr;
    size_t cap_med = 0, cap_med_out = 0, cap_med_cnt = 0;
    uint32_t med_part_cap = 0;
    // the planned call (kssd_gpu_sketch_plan), executed phase by phase (kssd_gpu_sketch_phase)
    struct {
        bool valid;
        const uint32_t *d_packed, *d_mask;
        const uint64_t *d_summ;  // the mask's summary words, or NULL (kssd_gpu_sketch_set_mask_summary)
        uint32_t n_genomes, flags, min_occ, big_min, n_slices;
        bool with_pos;
        uint64_t n_chunks, max_cap, max_big, cand_cap, out_cap;
        int grid;
        uint64_t *d_out_off;
        uint32_t *d_out_ids;
    } plan;
    std::vector<uint64_t> h_chunk_off;      // the planned batch's chunk offsets
    std::vector<uint64_t> dev_chunk_off, dev_reg_off;  // what d_chunk_off / d_reg_off hold (copies are skipped when unchanged)
    struct {  // the last scan: candidate list and staging layout a further tuple pass has to find as they were
        bool valid = false, fused = false;
        const void *cand = nullptr, *blk_info = nullptr;
        uint64_t cand_cap = 0, n_chunks = 0;
        uint32_t n_slices = 0;
        std::vector<uint64_t> reg_off;
        std::vector<uint2> med;
        std::vector<uint32_t> big;
    } scanned;
    // device-side buffers of the host-level sketch call (kssd_gpu_sketch_batch): grow-only
    uint32_t *d_in_packed, *d_in_mask, *d_b_ids, *d_b_pos;
    uint64_t *d_b_off;
    size_t cap_in_packed, cap_in_mask, cap_b_ids, cap_b_pos, cap_b_off;
    uint32_t *d_big_alt;          // sort output | tile counts | 2 accumulators
    size_t cap_big_alt;
    // inverted index (dist)
    uint32_t n_ref;
    uint64_t n_ref_ids;
    uint32_t *d_ref_sz;     // n_ref sketch sizes
    uint32_t *d_hkeys;      // the bucketed table: 8-byte slots (IdxSlot8) of a capped build, 16-byte ones (IdxSlot) of a counted one
    uint32_t h_log2;        // log2 of the number of buckets
    uint32_t idx_cap = 0;         // capped build in place: every bucket's slots, a multiple of 64 (0: exact build, descriptors)
    uint32_t idx_serial = 0;      // number of the build in place
    uint32_t *d_idx_flag = nullptr;  // serial of the last capped build that met a bucket fuller than its run
    bool idx_exact = false;       // builds of this context count first (a capped build of it has overflowed, or the caller asked)
    uint32_t *d_bkt;        // per bucket: counters | starts | cursors | descriptors (kssd_dist.inc)
    size_t cap_bkt;
    // device tokeniser (kssd_tok.inc)
    unsigned long long *d_tok_tab, *d_tok_pos, *d_tok_sup;
    uint32_t *d_tok_sum;
    uint8_t *d_tok_state, *d_text;
    size_t cap_tok_tab, cap_tok_pos, cap_tok_sum, cap_tok_state, cap_text, cap_tok_sup;
    uint32_t tok_files;
    bool tok_fastq = false;
    unsigned long long *d_x_off = nullptr;  // kssd_gpu_allgather_sketches: the gathered units of every rank
    uint32_t *d_x_ids = nullptr;
    size_t cap_x_off = 0, cap_x_ids = 0;
    void *tok_args_saved = nullptr;  // the tokeniser's kernel arguments of the last call (malloc; kssd_gpu_fasta_read_starts)
    uint32_t *d_hdr_cnt = nullptr;
    unsigned long long *d_hdr_pre = nullptr, *d_hdr_out = nullptr;
    size_t cap_hdr_cnt = 0, cap_hdr_pre = 0, cap_hdr_out = 0;
    unsigned long long *d_lb = nullptr;  // per genome: the look-back word of the per-genome kernel's DIRECT output (kssd_dedup.inc: FuseArgs)
    size_t cap_lb = 0;
    uint32_t lb_serial = 0;
    bool resident_valid = false;  // the last sketch call was a host-level one: its batch is in d_in_packed / d_in_mask (kssd_gpu_sketch_again)
    bool results_valid = false;   // ... and it succeeded: d_b_off / d_b_ids hold ITS sketches (kssd_gpu_resident_put copies them)
    uint32_t ranges_off_calls = 0;  // successful calls the switch below still lasts for
    uint32_t parts_off_calls = 0;
    bool parts_off = false;  // a batch of this context has shown a part that overflows inside a region that did not: no PARTS launches for a while
    bool ranges_off = false; // a batch of this context has shown keys that do not spread over id ranges: large genomes take the global-memory sort
    int fastq_min_qual = 0;  // kssd_gpu_set_fastq_quality
    bool fastq_reads = false;  // kssd_gpu_set_fastq_reads
    int *d_tok_q = nullptr;  // per tile: newlines around it (quality floor)
    size_t cap_tok_q = 0;
    hipEvent_t text_ev[32];  // kssd_gpu_text_put: the last 32 copies
    uint64_t text_puts;
    std::vector<unsigned long long> h_tok_tab;
    uint32_t *d_filt;       // negative filter of the index (kssd_gpu_index_set_filter): one word per four slots
    size_t cap_filt;
    bool idx_filter, idx_has_filter;  // asked for / the index in place was built with it
    uint32_t nofilt_lo, nofilt_hi;
    uint32_t *d_arrive;     // long query rows: arrival counters of the workgroups that share a row
    size_t cap_arrive;
    uint32_t *d_sel_cnt, *d_sel_out;  // report selection (kssd_gpu_dist_select): per-row counts / starts, candidate pairs
    size_t cap_sel_cnt, cap_sel_out;
    size_t cap_hash;
    uint32_t *d_post;       // postings (genome indices) | entries partitioned by bucket: ids | genomes
    size_t cap_pairs;
    size_t cap_ref;
    // timing: ring of HIP event pairs around the dominant kernel of each path (0 = sketch scan, 1 = dist rows)
    hipEvent_t ev_a[EV_KINDS][EV_RING], ev_b[EV_KINDS][EV_RING];
    unsigned ev_n[EV_KINDS];
    unsigned ev_every[EV_KINDS], ev_launch[EV_KINDS];  // every ev_every-th launch is bracketed (0: none), counted by ev_launch (kssd_gpu_set_kernel_timing)
};

static int ctx_upload_tables(kssd_gpu_ctx *c, const std::vector<uint32_t> &accepted)
{
    KssdParams &P = c->P;  // the builder picks the cuckoo multipliers
    std::vector<uint8_t> T1;
    std::vector<uint32_t> bloom;
    std::vector<KssdG> G;
    if (!kssd_build_tables(P, accepted, KSSD_GW, T1, bloom, G)) return KSSD_ERR_PARAM;
    const size_t gn = G.size();
    HIPCK(hipMalloc(&c->d_T1, SCAN_TAB_BYTES));
    HIPCK(hipMalloc(&c->d_G, gn * sizeof(KssdG) + KSSD_GFILT_WORDS * 4));  // (the exact table, then its bit filter)
    {
        std::vector<uint32_t> gf(KSSD_GFILT_WORDS, 0u);
        for (uint32_t dim : accepted) {
            const uint32_t h = kssd_gfilt_bit(dim);
            gf[h >> 5] |= 1u << (h & 31u);
        }
        c->d_gfilt = reinterpret_cast<uint32_t *>(c->d_G + gn);
        HIPCK(hipMemcpy(c->d_gfilt, gf.data(), KSSD_GFILT_WORDS * 4, hipMemcpyHostToDevice));
    }
    HIPCK(hipMemcpy(c->d_T1, T1.data(), KSSD_T1_BYTES, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_T1 + KSSD_T1_BYTES, bloom.data(), KSSD_BLOOM_WORDS * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_G, G.data(), gn * sizeof(KssdG), hipMemcpyHostToDevice));
    return KSSD_OK;
}

static int ctx_new(kssd_gpu_ctx **out, const kssd_shuf_hdr *hdr, std::vector<uint32_t> &accepted, int device)
{
    if (!out || !hdr) return KSSD_ERR_PARAM;
    KssdParams P;
    int prc = kssd_params_init(&P, hdr->k, hdr->subk, hdr->drlevel);
    if (prc == -1) return KSSD_ERR_PARAM;
    if (prc == -2) return KSSD_ERR_UNSUPPORTED;
    if (accepted.size() != P.dim_end) return KSSD_ERR_PARAM;
    {   // accepted[] is a set of sub-contexts: every value below 16^subk, none twice (a compact form from a damaged cache file)
        std::vector<uint32_t> sorted(accepted);
        std::sort(sorted.begin(), sorted.end());
        if ((uint64_t)sorted.back() >= (1ull << (4 * P.subk))) return KSSD_ERR_PARAM;
        for (size_t i = 1; i < sorted.size(); i++)
            if (sorted[i] == sorted[i - 1]) return KSSD_ERR_PARAM;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KSSD_ERR_NO_DEVICE;
    HIPCK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, device));
    kssd_gpu_ctx *c = new (std::nothrow) kssd_gpu_ctx();
    if (!c) return KSSD_ERR_NOMEM;
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->P = P;
    c->region_factor = 2.0;
    c->cand_factor = 1.5;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return KSSD_ERR_HIP; }
    int rc = ctx_upload_tables(c, accepted);
    if (rc != KSSD_OK) { delete c; return rc; }
    if (hipMalloc(&c->d_status, sizeof(SketchStatus)) != hipSuccess) {
        delete c;
        return KSSD_ERR_NOMEM;
    }
    for (int w = 0; w < EV_KINDS; w++) c->ev_every[w] = 1;  // (the events themselves are made by the first launch that needs one: kernel_timed)
    {
        const char *e = getenv("KSSD_MASK_SUMMARY");
        c->summ_auto = e && *e && *e != '0';
    }
    *out = c;
    return KSSD_OK;
}

extern "C" int kssd_gpu_create(kssd_gpu_ctx **out, const kssd_shuf_hdr *hdr, const int32_t *table, int device)
{
    if (!hdr || !table) return KSSD_ERR_PARAM;
    KssdParams P;
    int prc = kssd_params_init(&P, hdr->k, hdr->subk, hdr->drlevel);
    if (prc == -1) return KSSD_ERR_PARAM;
    if (prc == -2) return KSSD_ERR_UNSUPPORTED;
    std::vector<uint32_t> accepted;
    if (!kssd_accepted_from_table(P, table, accepted)) return KSSD_ERR_PARAM;  // not a permutation
    return ctx_new(out, hdr, accepted, device);
}

extern "C" int kssd_gpu_create_compact(kssd_gpu_ctx **out, const kssd_shuf_hdr *hdr, const uint32_t *acc, uint32_t n,
                                       int device)
{
    if (!hdr || !acc) return KSSD_ERR_PARAM;
    std::vector<uint32_t> accepted(acc, acc + n);
    return ctx_new(out, hdr, accepted, device);
}

extern "C" int kssd_gpu_create_for_dist(kssd_gpu_ctx **out, int kmerlen, int device)
{
    if (!out || kmerlen < 2 || kmerlen > 30 || (kmerlen & 1)) return KSSD_ERR_PARAM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KSSD_ERR_NO_DEVICE;
    HIPCK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, device));
    kssd_gpu_ctx *c = new (std::nothrow) kssd_gpu_ctx();
    if (!c) return KSSD_ERR_NOMEM;
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    memset(&c->P, 0, sizeof c->P);
    c->P.k = kmerlen / 2;
    c->dist_only = true;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return KSSD_ERR_HIP; }
    if (hipMalloc(&c->d_status, sizeof(SketchStatus)) != hipSuccess) { delete c; return KSSD_ERR_NOMEM; }
    for (int w = 0; w < EV_KINDS; w++) c->ev_every[w] = 1;  // (the events themselves are made by the first launch that needs one: kernel_timed)
    {
        const char *e = getenv("KSSD_MASK_SUMMARY");
        c->summ_auto = e && *e && *e != '0';
    }
    *out = c;
    return KSSD_OK;
}

extern "C" void kssd_gpu_destroy(kssd_gpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    void *ptrs[] = {c->d_T1, c->d_G, c->d_chunk_gid, c->d_chunk_off, c->d_reg_off, c->d_cursor, c->d_kept,
                    c->d_regions, c->d_status, c->d_ref_sz, c->d_hkeys, c->d_post, c->d_cand, c->d_cand_count, c->d_blk_info, c->d_big_alt,
                    c->d_in_packed, c->d_in_mask, c->d_b_ids, c->d_b_pos, c->d_b_off, c->d_bkt, c->d_sel_cnt, c->d_sel_out, c->d_arrive, c->d_tok_tab, c->d_tok_pos, c->d_tok_sum, c->d_tok_state, c->d_text, c->d_filt, c->d_tok_sup, c->d_tok_q, c->d_x_off, c->d_x_ids, c->d_med, c->d_med_out, c->d_med_cnt, c->d_hdr_cnt, c->d_hdr_pre, c->d_hdr_out, c->d_idx_flag, c->d_lb, c->d_in_summ};
    for (void *p : ptrs)
        if (p) hipFree(p);
    free(c->tok_args_saved);
    for (hipEvent_t e : c->text_ev)
        if (e) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    for (int w = 0; w < EV_KINDS; w++)
        for (int i = 0; i < EV_RING; i++) {
            if (c->ev_a[w][i]) hipEventDestroy(c->ev_a[w][i]);
            if (c->ev_b[w][i]) hipEventDestroy(c->ev_b[w][i]);
        }
    delete c;
}

extern "C" int kssd_gpu_get_info(const kssd_gpu_ctx *c, kssd_gpu_info *o)
{
    if (!c || !o) return KSSD_ERR_PARAM;
    o->k = c->P.k; o->subk = c->P.subk; o->drlevel = c->P.drlevel;
    o->kmerlen = 2 * c->P.k; o->dim_rd_len = 2 * c->P.drlevel;
    o->comp_num = (int32_t)c->P.comp_num; o->comp_bits = c->P.comp_bits;
    o->dim_end = c->P.dim_end; o->hashsize = c->P.hashsize; o->hashlimit = c->P.hashlimit;
    o->device = c->device; o->cu_count = c->cu_count;
    return KSSD_OK;
}

extern "C" void kssd_gpu_free(void *p) { free(p); }

extern "C" int kssd_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) return 0;
    return n;
}

__global__ void warm_up_kernel(uint32_t *p) { if (p) *p = 1; }

extern "C" int kssd_gpu_warm_up(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KSSD_ERR_NO_DEVICE;
    HIPCK(hipSetDevice(device));
    HIPCK(hipFree(nullptr));  // the device's context
    hipLaunchKernelGGL(warm_up_kernel, dim3(1), dim3(64), 0, 0, (uint32_t *)nullptr);  // the library's code object onto the device
    HIPCK(hipGetLastError());
    HIPCK(hipDeviceSynchronize());
    return KSSD_OK;
}

template <typename T>
static int ensure(T **p, size_t *cap, size_t need, size_t slack = 0)
{
    if (*p && *cap >= need) return KSSD_OK;
    if (*p) hipFree(*p);
    *p = nullptr;
    size_t n = need + need / 4 + slack;
    if (hipMalloc(p, n * sizeof(T)) != hipSuccess) { *cap = 0; return KSSD_ERR_NOMEM; }
    *cap = n;
    return KSSD_OK;
}

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of a 64-bit wave mask below this lane
__device__ __forceinline__ uint32_t rank_in(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// make this wave's LDS writes visible to its other lanes (LDS ops of one wave complete in order;
// the fence stops the compiler from moving the accesses across)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += t;
    }
    return v;
}

// the same through DPP moves (row shifts inside the rows of 16 lanes, then the two row broadcasts): no LDS crossbar traffic,
// which a kernel that keeps the LDS pipe busy with table reads cannot afford (a ds_bpermute costs what a conflicted read costs)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// ---------------------------------------------------------------------------------------------------
// kernel 0: chunk -> genome map
// ---------------------------------------------------------------------------------------------------
// (it also zeroes the small per-call state: four separate memsets cost more than this whole kernel)
__global__ void chunk_gid_kernel(const uint64_t *__restrict__ chunk_off, uint32_t n_genomes, uint64_t n_chunks,
                                 uint32_t *__restrict__ chunk_gid, uint32_t *__restrict__ cursor, uint32_t *__restrict__ cand_count,
                                 uint32_t n_slices, uint32_t *__restrict__ status_words)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < sizeof(SketchStatus) / 4) status_words[c] = 0;
    if (c < n_genomes) cursor[c] = 0;
    if (c < n_slices) cand_count[c] = 0;
    if (c >= n_chunks) return;
    uint32_t lo = 0, hi = n_genomes;  // last g with chunk_off[g] <= c
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid;
        else hi = mid;
    }
    chunk_gid[c] = lo;
}

// the same map written genome by genome (one workgroup each, plain coalesced stores, no search): the cheaper way
// when genomes span many chunks.  The per-call state is zeroed by the first workgroups like above.
__global__ void chunk_gid_by_genome_kernel(const uint64_t *__restrict__ chunk_off, uint32_t n_genomes,
                                           uint32_t *__restrict__ chunk_gid, uint32_t *__restrict__ cursor, uint32_t *__restrict__ cand_count,
                                           uint32_t n_slices, uint32_t *__restrict__ status_words)
{
    // grid = (genomes or more, parts): workgroup (g, y) writes every gridDim.y-th run of 256 chunks of genome g
    const uint64_t t = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * gridDim.y * blockDim.x;
    if (t < sizeof(SketchStatus) / 4) status_words[t] = 0;
    for (uint64_t i = t; i < n_genomes; i += nt) cursor[i] = 0;
    for (uint64_t i = t; i < n_slices; i += nt) cand_count[i] = 0;
    if (blockIdx.x >= n_genomes) return;
    const uint32_t g = blockIdx.x;
    for (uint64_t c = chunk_off[g] + (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; c < chunk_off[g + 1]; c += (uint64_t)gridDim.y * blockDim.x)
        chunk_gid[c] = g;
}

#include "kssd_scan.inc"
#include "kssd_exact.inc"
#include "kssd_dedup.inc"
#include "kssd_big.inc"
// kernel 3: exclusive scan of the kept counts -> CSR offsets (single workgroup, n_genomes is small)
__global__ __launch_bounds__(1024) void sketch_offsets_kernel(const uint32_t *__restrict__ kept, uint32_t n,
                                                               unsigned long long *__restrict__ out_off,
                                                               unsigned long long out_cap, SketchStatus *st)
{
    __shared__ unsigned long long wsum[16];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = tid * per, e = (b + per < n) ? b + per : n;
    unsigned long long s = 0;
    for (uint32_t i = b; i < e; i++) s += kept[i];
    // block-wide exclusive scan of the per-thread sums: wave scan (64-bit through two 32-bit shuffles) + 16 wave totals
    unsigned long long incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)incl, d, 64), hi = __shfl_up((uint32_t)(incl >> 32), d, 64);
        if ((int)lane >= d) incl += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long off = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const unsigned long long t = wsum[w];
        if (w < wave) off += t;
        total += t;
    }
    if (tid == 0) {
        out_off[n] = total;
        st->total_ids = total;
        if (total > out_cap) st->out_overflow = 1;
    }
    unsigned long long run = off + incl - s;
    for (uint32_t i = b; i < e; i++) { out_off[i] = run; run += kept[i]; }
}

// kernel 4: gather every genome's kept ids into the dense CSR
// SELF: the workgroup adds up the kept counts in front of its genome itself (g values, coalesced, a block reduction) and
// writes the genome's CSR offset -- no offsets kernel in front for batches of up to a few thousand genomes (one launch of
// ~5 us less; at 10 000 genomes the 10 000 partial sums would read 200 MB, the scan kernel above stays)
template <typename K, bool SELF>
__global__ __launch_bounds__(256) void sketch_gather_kernel(const unsigned long long *__restrict__ reg_off,
                                                             const K *__restrict__ regions,
                                                             const uint32_t *__restrict__ kept,
                                                             unsigned long long *__restrict__ out_off,
                                                             uint32_t *__restrict__ out_ids, uint32_t *__restrict__ out_pos,
                                                             uint32_t n_genomes, unsigned long long out_cap, SketchStatus *st)
{
    const uint32_t g = blockIdx.x;
    const uint32_t n = kept[g];
    unsigned long long o0;
    if (SELF) {
        __shared__ unsigned long long s_part[4];
        unsigned long long sum = 0;
        for (uint32_t i = threadIdx.x; i < g; i += 256) sum += kept[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), d, 64);
            sum += ((unsigned long long)hi << 32) | lo;
        }
        if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = sum;
        __syncthreads();
        o0 = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (threadIdx.x == 0 && blockIdx.y == 0) {
            out_off[g] = o0;
            if (g + 1 == n_genomes) {
                out_off[n_genomes] = o0 + n;
                st->total_ids = o0 + n;
                if (o0 + n > out_cap) st->out_overflow = 1;
            }
        }
        if (o0 + n > out_cap) return;  // (the last genome's workgroup reports it; nothing is written beyond the caller's array)
    } else {
        if (st->out_overflow) return;
        o0 = out_off[g];
    }
    const unsigned long long r0 = reg_off[g];
    // gridDim.y > 1 when the batch holds a large genome (a read set's 262 570 ids copied by one workgroup: 154 us)
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        const K kv = regions[r0 + i];
        out_ids[o0 + i] = KeyOps<K>::id(kv);
        if (out_pos) out_pos[o0 + i] = KeyOps<K>::pos(kv);
    }
}

// ---------------------------------------------------------------------------------------------------
// sketch entry points
// ---------------------------------------------------------------------------------------------------
// the launch carries its own start / stop events (hipExtLaunchKernelGGL): they take the timestamps of the dispatch itself,
// so kssd_gpu_kernel_time reports the kernel's execution time like the profiler does, not the distance between two
// stream markers (which also holds the launch latency whenever the kernel in front is too short to hide it)
// is this launch of the path's dominant kernel (0 = scan, 1 = rows) one of the bracketed ones?
static bool kernel_timed(kssd_gpu_ctx *c, int which)
{
    const unsigned every = c->ev_every[which], n = c->ev_launch[which]++;
    if (every == 0 || n % every != 0) return false;
    const unsigned i = c->ev_n[which] % EV_RING;  // the ring's events are made when a launch first needs them (512 of them up front were
    if (!c->ev_a[which][i] && hipEventCreate(&c->ev_a[which][i]) != hipSuccess) return false;  // milliseconds of every context's creation)
    if (!c->ev_b[which][i] && hipEventCreate(&c->ev_b[which][i]) != hipSuccess) return false;
    return true;
}

template <int SUBK, int ABL = 0>
static int launch_scan(kssd_gpu_ctx *c, const ScanArgs &a, int grid, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (a.summ && ABL == 0) {  // the batch came with its summary words (kssd_gpu_sketch_set_mask_summary): the mask is not streamed
        if (ev_start) hipExtLaunchKernelGGL((sketch_scan_kernel<SUBK, 0, 1>), dim3(grid), dim3(SCAN_THREADS), 0, s, ev_start, ev_stop, 0, a);
        else hipLaunchKernelGGL((sketch_scan_kernel<SUBK, 0, 1>), dim3(grid), dim3(SCAN_THREADS), 0, s, a);
    } else {
        if (ev_start) hipExtLaunchKernelGGL((sketch_scan_kernel<SUBK, ABL>), dim3(grid), dim3(SCAN_THREADS), 0, s, ev_start, ev_stop, 0, a);
        else hipLaunchKernelGGL((sketch_scan_kernel<SUBK, ABL>), dim3(grid), dim3(SCAN_THREADS), 0, s, a);
    }
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

#ifdef KSSD_DEV
static unsigned long long *g_dev_dedup_times;
static unsigned long long *dev_dedup_times()
{
    if (!g_dev_dedup_times && getenv("KSSD_DEV_DEDUPTIME")) {
        hipMalloc(&g_dev_dedup_times, 65536 * 4 * 8);
        hipMemset(g_dev_dedup_times, 0, 65536 * 4 * 8);
    }
    return g_dev_dedup_times;
}
extern "C" int kssd_gpu_dev_deduptimes(unsigned long long *out, uint32_t n_genomes)
{
    if (!g_dev_dedup_times) return KSSD_ERR_PARAM;
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy(out, g_dev_dedup_times, (size_t)n_genomes * 4 * 8, hipMemcpyDeviceToHost));
    return KSSD_OK;
}
#endif

// per-genome dedup (LDS sort; large genomes: sorted by ranges of their keys, or -- keys that do not spread -- by the bitonic network of kssd_big.inc), CSR offsets, gather; K = key type
template <typename K>
static int finish_sketch(kssd_gpu_ctx *c, uint32_t n_genomes, uint32_t flags, uint32_t min_occ, uint32_t big_min, uint64_t max_cap,
                         uint64_t max_big, uint64_t *d_out_off, uint32_t *d_out_ids, uint32_t *d_out_pos, uint64_t out_cap,
                         hipStream_t s)
{
    int rc;
    K *regions = reinterpret_cast<K *>(c->d_regions);
    // The LDS key array of the per-genome workgroups: 5/8 of the largest staging region, rounded up to a power of two -- the
    // regions are sized at twice the expected emission (region_factor), the array at 1.25 x.  A genome that emits more than the
    // array holds is reported like one that overflows its region (the call is repeated with larger regions, hence a larger
    // array).  Sized by the regions themselves the array took 4 096 keys for bacterial genomes that emit 1 220: 56 KB of LDS
    // per workgroup, TWO workgroups per CU, and a batch of 1 000 genomes ran in two generations (the second started 35 us
    // late: profiles/r03G_per_genome_occupancy.txt).
    uint32_t np = 64;
    while (np < (uint32_t)((max_cap * 5 + 7) / 8)) np <<= 1;
    FuseArgs fx;
    memset(&fx, 0, sizeof fx);
    PartArgs px;
    memset(&px, 0, sizeof px);
    // the bucket sort's LDS arrays behind the key array (lds_bucket_sort): as many key slots as the key array has, up to
    // DEDUP_BSORT_MAX; by-position keys lead with the position, which does not spread over the id range: bitonic only
    bool bsort = !(c->plan.flags & KSSD_SKETCH_BY_POS);
#ifdef KSSD_DEV
    if (getenv("KSSD_DEV_NO_BUCKET_SORT")) bsort = false;  // (development A/B: the bitonic network for every genome)
#endif
    fx.id_bits = (uint32_t)(4 * (c->P.k - c->P.drlevel)) - c->P.pass_bits;
    fx.lds_keys = np;
    // DIRECT: no genome of the batch needs the parts / ranges / global-memory paths (their results arrive in the staging regions
    // and are gathered below) -- every workgroup of the per-genome launch writes its ids into the CSR itself
    const bool direct = c->h_big.empty() && c->h_med.empty() && n_genomes > 0 && n_genomes <= DEDUP_DIRECT_MAX;
    if (direct) {
        const size_t before = c->cap_lb;
        if ((rc = ensure(&c->d_lb, &c->cap_lb, (size_t)n_genomes)) != KSSD_OK) return rc;
        if (c->cap_lb != before) {  // (fresh words carry no launch's serial number ...)
            HIPCK(hipMemsetAsync(c->d_lb, 0, c->cap_lb * 8, s));
            c->lb_serial = 0;
        }
        if (++c->lb_serial == 0) {  // (... and after 2^32 launches the numbers start over on cleared words)
            HIPCK(hipMemsetAsync(c->d_lb, 0, c->cap_lb * 8, s));
            c->lb_serial = 1;
        }
        fx.lb = c->d_lb;
        fx.lb_serial = c->lb_serial;
        fx.n_genomes = n_genomes;
        fx.out_off = (unsigned long long *)d_out_off;
        fx.out_ids = d_out_ids;
        fx.out_pos = d_out_pos;
        fx.out_cap = (unsigned long long)out_cap;
    }
    auto bsort_slots = [&](uint32_t key_slots) -> uint32_t {  // what fits beside the key array and the kernel's static LDS
        if (!bsort) return 0u;
        const uint32_t bs = key_slots < DEDUP_BSORT_MAX ? key_slots : DEDUP_BSORT_MAX;
        return (size_t)key_slots * sizeof(K) + (size_t)bs * (sizeof(K) + 4) + DEDUP_BSORT_HEAVY * 4 <= (size_t)144 * 1024 ? bs : 0u;
    };
    fx.bsort_keys = bsort_slots(np);
    const size_t dlds = (size_t)np * sizeof(K) + (size_t)fx.bsort_keys * (sizeof(K) + 4) + (fx.bsort_keys ? DEDUP_BSORT_HEAVY * 4 : 0);
    if (c->h_big.empty() && c->h_med.empty()) {
        // no genome needs staged tuples: exact stage and per-genome sort in one kernel, straight from the candidate list
        const auto &pl = c->plan;
        fx.cand = reinterpret_cast<const unsigned long long *>(c->d_cand);  // (8-byte records: phase_scan's rec8)
        fx.blk_info = c->d_blk_info;
        fx.chunk_off = (const unsigned long long *)c->d_chunk_off;
        fx.packed = pl.d_packed;
        fx.mask = pl.d_mask;
        fx.G = c->d_G;
        fx.gfilt = c->d_gfilt;
        fx.carry = kssd_carry_ok(c->P) ? 1u : 0u;
        fx.by_pos = (pl.flags & KSSD_SKETCH_BY_POS) ? 1u : 0u;
        fx.lds_keys = np;
        fx.cand_count = c->d_cand_count;
        fx.n_slices = pl.n_chunks ? pl.n_slices : 0u;  // (no chunk, no scan: nothing to add up)
        fx.cand_cap = pl.cand_cap;
#ifdef KSSD_DEV
        fx.dev_times = n_genomes <= 65536 ? dev_dedup_times() : nullptr;
        fx.dev_split = getenv("KSSD_DEV_GATHERSPLIT") ? 1u : 0u;
#endif
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_dedup_kernel<K, DEDUP_FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dlds < 65536 ? 65536 : dlds)));
        hipLaunchKernelGGL((sketch_dedup_kernel<K, DEDUP_FUSED>), dim3(n_genomes), dim3(DEDUP_THREADS), dlds, s, c->P,
                           (const unsigned long long *)c->d_reg_off, (const uint32_t *)c->d_cursor, regions, c->d_kept,
                           flags, min_occ, big_min, c->d_status, fx, px);
    } else if (c->h_big.size() + c->h_med.size() < (size_t)n_genomes) {  // (a batch of nothing but large genomes -- one read set -- has no workgroup to send)
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_dedup_kernel<K, DEDUP_STAGED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dlds < 65536 ? 65536 : dlds)));
        hipLaunchKernelGGL((sketch_dedup_kernel<K, DEDUP_STAGED>), dim3(n_genomes), dim3(DEDUP_THREADS), dlds, s, c->P,
                           (const unsigned long long *)c->d_reg_off, (const uint32_t *)c->d_cursor, regions, c->d_kept,
                           flags, min_occ, big_min, c->d_status, fx, px);
    }
    if (!c->h_med.empty()) {
        // genomes between one LDS sort and sixteen: sorted in LDS in parts (ranges of the ids' top bits), two launches for all of them
        uint32_t lg_max = 1;
        for (const uint2 &m : c->h_med) lg_max = m.y > lg_max ? m.y : lg_max;
        px.list = c->d_med;
        px.out = c->d_med_out;
        px.cnt = c->d_med_cnt;
        px.part_cap = c->med_part_cap;
        px.id_bits = (uint32_t)(4 * (c->P.k - c->P.drlevel)) - c->P.pass_bits;
        fx.bsort_keys = bsort_slots(px.part_cap);
        const size_t plds = (size_t)px.part_cap * sizeof(K) + (size_t)fx.bsort_keys * (sizeof(K) + 4) + (fx.bsort_keys ? DEDUP_BSORT_HEAVY * 4 : 0);
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_dedup_kernel<K, DEDUP_PARTS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(plds < 65536 ? 65536 : plds)));
        hipLaunchKernelGGL((sketch_dedup_kernel<K, DEDUP_PARTS>), dim3((unsigned)c->h_med.size(), 1u << lg_max), dim3(DEDUP_THREADS), plds, s, c->P,
                           (const unsigned long long *)c->d_reg_off, (const uint32_t *)c->d_cursor, regions, c->d_kept,
                           flags, min_occ, big_min, c->d_status, fx, px);
        hipLaunchKernelGGL((dedup_parts_finish_kernel<K>), dim3((unsigned)c->h_med.size(), 1u << lg_max), dim3(256), 0, s, c->P,
                           (const unsigned long long *)c->d_reg_off, regions, c->d_kept, flags, c->d_status, px);
    }
    if (!c->h_big.empty()) {
        const size_t kw = sizeof(K) / 4;  // u32 words per key
        // RANGES (kernel 2c) unless an earlier attempt of this context has met keys that do not spread, or the genome is beyond
        // what 2^RNG_MAX_LOG2_BINS bins of twice the mean hold (hundreds of millions of staged keys)
        const bool ranges = !c->ranges_off && max_big <= ((uint64_t)(2 * RNG_BIN_MEAN) << RNG_MAX_LOG2_BINS);
        const size_t n_tiles_max = ranges ? (size_t)(max_big / RNG_T + 2) : (size_t)((max_big + BIG_TILE - 1) / BIG_TILE);
        const size_t rng_words = ranges ? 2 * n_tiles_max + 2 + 2 * ((size_t)1 << RNG_MAX_LOG2_BINS) + 2 + (size_t)256 * ((size_t)1 << RNG_MAX_LOG2_BINS) : 0;
        // keys (sort output | partitioned keys) | item / tile counts | 8 accumulator words | RANGES: item starts, bin totals, bin starts, rows of counts
        // (the global-memory sort works on the power of two from the largest region on: big_bitonic_sort)
        size_t sort_room = (size_t)max_big;
        if (!ranges) {
            sort_room = BIT_TILE;
            while (sort_room < (size_t)max_big) sort_room <<= 1;
        }
        if ((rc = ensure(&c->d_big_alt, &c->cap_big_alt, sort_room * kw + n_tiles_max + 8 + rng_words)) != KSSD_OK) return rc;
        for (uint32_t g : c->h_big) {
            const uint64_t r0 = c->h_reg_off[g], cap = c->h_reg_off[g + 1] - r0;
            K *region = regions + r0, *sorted = reinterpret_cast<K *>(c->d_big_alt);
            uint32_t *tile_cnt = c->d_big_alt + sort_room * kw, *accum = tile_cnt + n_tiles_max;
            const uint32_t *cur = c->d_cursor + g;
            if (ranges) {
                // the keys' leading field: the id, or (--byread) the position inside the genome
                uint32_t lead_bits = (uint32_t)(4 * (c->P.k - c->P.drlevel)) - c->P.pass_bits;
                if (c->plan.flags & KSSD_SKETCH_BY_POS) {
                    const uint64_t positions = (c->h_chunk_off[g + 1] - c->h_chunk_off[g]) * KSSD_CHUNK;
                    lead_bits = 1;
                    while (lead_bits < 32 && (1ull << lead_bits) < positions) lead_bits++;
                }
                uint32_t lg = 4;
                while (lg < RNG_MAX_LOG2_BINS && ((uint64_t)RNG_BIN_MEAN << lg) < cap) lg++;
                if (lg > lead_bits) lg = lead_bits;
                const uint32_t nb = 1u << lg, shift = lead_bits - lg;
                uint64_t rows64 = cap / 16384ull;  // a workgroup's share: at least 16 384 keys
                const uint32_t rows = (uint32_t)(rows64 < 1 ? 1 : rows64 > 256 ? 256 : rows64);
                const uint32_t n_items = (uint32_t)(cap / RNG_T + 2);
                uint32_t *item_s0 = accum + 8, *item_bin = item_s0 + n_tiles_max, *bin_tot = item_bin + n_tiles_max + 2,
                         *bin_start = bin_tot + ((size_t)1 << RNG_MAX_LOG2_BINS), *wg_hist = bin_start + ((size_t)1 << RNG_MAX_LOG2_BINS) + 2;
                hipLaunchKernelGGL((big_rng_hist_kernel<K>), dim3(rows), dim3(1024), (size_t)nb * 4, s, (const K *)region, (unsigned long long)cap, cur,
                                   c->d_kept + g, accum, c->d_status, nb, shift, wg_hist);
                hipLaunchKernelGGL(big_rng_colscan_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, (unsigned long long)cap, cur, nb, rows, wg_hist, bin_tot);
                if ((size_t)(nb + 1) * 4 > 65536)
                    HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(big_rng_binscan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)((nb + 1) * 4)));
                hipLaunchKernelGGL(big_rng_binscan_kernel, dim3(1), dim3(DEDUP_THREADS), (size_t)(nb + 1) * 4, s, (unsigned long long)cap, cur, nb,
                                   (const uint32_t *)bin_tot, bin_start, item_bin, n_items);
                hipLaunchKernelGGL((big_rng_scatter_kernel<K>), dim3(rows), dim3(1024), (size_t)nb * 4, s, (const K *)region, (unsigned long long)cap, cur,
                                   nb, shift, (const uint32_t *)wg_hist, (const uint32_t *)bin_start, sorted);
                PartArgs rx = px;
                rx.parted = sorted;
                rx.bin_start = bin_start;
                rx.item_cnt = tile_cnt;
                rx.item_s0 = item_s0;
                rx.item_bin = item_bin;
                rx.acc = accum;
                rx.rng_bins = nb;
                rx.rng_g = g;
                rx.part_cap = RNG_PART_CAP;
                rx.id_bits = lead_bits;
                FuseArgs rf = fx;
                rf.id_bits = lead_bits;
                rf.lds_keys = RNG_PART_CAP;
                rf.bsort_keys = bsort_slots(RNG_PART_CAP);
                if (rf.bsort_keys > RNG_BSORT_KEYS) rf.bsort_keys = RNG_BSORT_KEYS;
#ifdef KSSD_DEV
                rf.dev_times = dev_dedup_times();  // (per item here)
#endif
                const size_t rlds = (size_t)RNG_PART_CAP * sizeof(K) + (size_t)rf.bsort_keys * (sizeof(K) + 4) + (rf.bsort_keys ? DEDUP_BSORT_HEAVY * 4 : 0);
                HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_dedup_kernel<K, DEDUP_RANGES>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(rlds < 65536 ? 65536 : rlds)));
                hipLaunchKernelGGL((sketch_dedup_kernel<K, DEDUP_RANGES>), dim3(n_items), dim3(DEDUP_THREADS), rlds, s, c->P,
                                   (const unsigned long long *)c->d_reg_off, (const uint32_t *)c->d_cursor, regions, c->d_kept,
                                   flags, min_occ, big_min, c->d_status, rf, rx);
                hipLaunchKernelGGL(big_scan_kernel, dim3(1), dim3(1024), 0, s, tile_cnt, n_items, (const uint32_t *)accum, c->P.hashlimit,
                                   flags, g, (unsigned long long)cap, cur, c->d_kept + g, c->d_status);
                hipLaunchKernelGGL((big_rng_copy_kernel<K>), dim3(n_items), dim3(256), 0, s, (unsigned long long)cap, cur, (const K *)sorted,
                                   (const uint32_t *)tile_cnt, (const uint32_t *)item_s0, n_items, (const uint32_t *)(c->d_kept + g), region);
                continue;
            }
            const uint32_t n_tiles = (uint32_t)((cap + BIG_TILE - 1) / BIG_TILE);
            hipLaunchKernelGGL((big_pad_kernel<K>), dim3(1024), dim3(256), 0, s, region, (unsigned long long)cap, cur, c->d_kept + g, accum,
                               c->d_status);
            {
                unsigned long long np2 = BIT_TILE;
                while (np2 < cap) np2 <<= 1;
                if ((rc = big_bitonic_sort<K>(region, (unsigned long long)cap, sorted, np2, s)) != KSSD_OK) return rc;
            }
            hipLaunchKernelGGL((big_runs_kernel<K, false>), dim3(n_tiles), dim3(BIG_THREADS), 0, s, (const K *)sorted,
                               (unsigned long long)cap, cur, flags, min_occ, tile_cnt, accum, (K *)nullptr);
            hipLaunchKernelGGL(big_scan_kernel, dim3(1), dim3(1024), 0, s, tile_cnt, n_tiles, (const uint32_t *)accum, c->P.hashlimit,
                               flags, g, (unsigned long long)cap, cur, c->d_kept + g, c->d_status);
            hipLaunchKernelGGL((big_runs_kernel<K, true>), dim3(n_tiles), dim3(BIG_THREADS), 0, s, (const K *)sorted,
                               (unsigned long long)cap, cur, flags, min_occ, tile_cnt, accum, region);
        }
    }
    if (direct) return KSSD_OK;  // (the per-genome launch has written the CSR)
    const dim3 ggrid(n_genomes, c->h_big.empty() ? (c->h_med.empty() ? 1u : 16u) : (n_genomes < 64u ? 256u : 16u));
    if (n_genomes <= 4096) {
        hipLaunchKernelGGL((sketch_gather_kernel<K, true>), ggrid, dim3(256), 0, s, (const unsigned long long *)c->d_reg_off, (const K *)regions,
                           (const uint32_t *)c->d_kept, (unsigned long long *)d_out_off, d_out_ids, d_out_pos, n_genomes,
                           (unsigned long long)out_cap, c->d_status);
    } else {
        hipLaunchKernelGGL(sketch_offsets_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t *)c->d_kept, n_genomes,
                           (unsigned long long *)d_out_off, (unsigned long long)out_cap, c->d_status);
        hipLaunchKernelGGL((sketch_gather_kernel<K, false>), ggrid, dim3(256), 0, s, (const unsigned long long *)c->d_reg_off, (const K *)regions,
                           (const uint32_t *)c->d_kept, (unsigned long long *)d_out_off, d_out_ids, d_out_pos, n_genomes,
                           (unsigned long long)out_cap, c->d_status);
    }
    return KSSD_OK;
}

// A sketch call = a plan (host: validation, workspace sizes, staging layout) and four phases on a stream:
//   PREP    per-call state, chunk -> genome map          (no LDS, small)
//   SCAN    the scan kernel                              (every CU's LDS)
//   EXACT   stage 2 on the candidates                    (no LDS, bound by random HBM reads)
//   FINISH  per-genome dedup, CSR offsets, gather        (LDS sort)
// kssd_gpu_sketch_device runs them back to back.  A caller that streams batches through several contexts puts its
// own event waits between the phases, so that the LDS-free phases of one batch run underneath the scan of another
// (bench.py); the results do not depend on how the phases are interleaved with other contexts' work.
extern "C" int kssd_gpu_sketch_set_mask_summary(kssd_gpu_ctx *c, const uint64_t *d_summary)
{
    if (!c) return KSSD_ERR_PARAM;
    c->d_next_summ = d_summary;
    return KSSD_OK;
}

extern "C" int kssd_gpu_mask_summarise_device(kssd_gpu_ctx *c, const uint32_t *d_mask, uint64_t n_chunks, uint64_t *d_summary, void *stream)
{
    if (!c || !d_mask || !d_summary) return KSSD_ERR_PARAM;
    if (n_chunks == 0) return KSSD_OK;
    if (n_chunks > 0xFFFFFFFFull * 4) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    hipLaunchKernelGGL(mask_summarise_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_mask,
                       (unsigned long long)n_chunks, reinterpret_cast<unsigned long long *>(d_summary));
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

extern "C" int kssd_gpu_sketch_plan(kssd_gpu_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask,
                                    const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                    uint64_t *d_out_off, uint32_t *d_out_ids, uint64_t out_cap)
{
    if (!c) return KSSD_ERR_PARAM;
    c->plan.valid = false;
    c->plan.d_summ = c->d_next_summ;  // (one plan's: a summary of another mask would be wrong bits, not a slower scan)
    c->d_next_summ = nullptr;
    if (d_packed != c->d_in_packed) c->resident_valid = false;
    if (c->dist_only || !h_chunk_off || !d_out_off || (!d_out_ids && out_cap)) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    c->last_launch_rc = KSSD_OK;
    c->last_n_genomes = n_genomes;
    if (min_occ < 1) min_occ = 1;
    const bool with_pos = (flags & (KSSD_SKETCH_FIRST_POS | KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS)) != 0;  // 64-bit keys, second output array
    {
        const uint32_t m = flags & (KSSD_SKETCH_FIRST_POS | KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS);
        if (m & (m - 1)) return KSSD_ERR_PARAM;  // one of them per call
    }
    if (flags & KSSD_SKETCH_BY_POS) {
        // the keys are (position, tuple): every key is its own run, so the dedup kernels keep everything -- as long as no
        // keep rule interferes (reads2mco has none: no table, no capacity limit, id 0 written like any other)
        if (flags & KSSD_SKETCH_UNIQ) return KSSD_ERR_PARAM;
        flags |= KSSD_SKETCH_KEEP_ZERO | KSSD_SKETCH_NO_CAPACITY;
        min_occ = 1;
    }
    if (c->P.pass) flags |= KSSD_SKETCH_KEEP_ZERO;  // id 0 of pass s is the tuple s, not the tuple 0 the reference never stores
    if (with_pos && !c->d_out_pos) return KSSD_ERR_PARAM;  // kssd_gpu_sketch_set_pos_output first
    const uint64_t n_chunks = h_chunk_off[n_genomes];
    auto &pl = c->plan;
    pl.d_packed = d_packed; pl.d_mask = d_mask; pl.n_genomes = n_genomes; pl.flags = flags; pl.min_occ = min_occ;
    pl.with_pos = with_pos; pl.n_chunks = n_chunks; pl.d_out_off = d_out_off; pl.d_out_ids = d_out_ids; pl.out_cap = out_cap;
    if (n_genomes == 0) {
        pl.valid = true;
        return KSSD_OK;
    }
    // staging regions: expected emissions = positions * dim_end / 16^subk, times a safety factor
    const double rate = (double)c->P.dim_end / (double)(1ull << (4 * c->P.subk));
    c->h_reg_off.resize((size_t)n_genomes + 1);
    c->h_big.clear();
    c->h_med.clear();
    uint64_t med_part = 0;
    uint32_t big_min = c->lds_sort_limit && c->lds_sort_limit < DEDUP_MAX_N ? c->lds_sort_limit : DEDUP_MAX_N;
    if (with_pos && big_min > DEDUP_MAX_N / 2) big_min = DEDUP_MAX_N / 2;  // 8-byte keys: half as many fit the LDS sort
    uint64_t acc = 0, max_cap = 0, max_big = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
        if (h_chunk_off[g + 1] < h_chunk_off[g]) return KSSD_ERR_PARAM;
        const uint64_t pos = (h_chunk_off[g + 1] - h_chunk_off[g]) * KSSD_CHUNK;
        if (with_pos && pos >= (1ull << 32)) { c->last_launch_rc = KSSD_ERR_UNSUPPORTED; return KSSD_ERR_UNSUPPORTED; }
        uint64_t cap = (uint64_t)((double)pos * rate * c->region_factor) + 256;
        const bool all_positions = cap > pos;
        if (all_positions) cap = pos;  // a genome cannot emit more tuples than it has positions
        // A region that is all of the genome's positions cannot grow any further: after an overflow (the factor has moved) such a
        // genome must not go back to a path whose room depends on how its ids SPREAD -- a tandem repeat of 190 kb at a dense
        // parameter set stages 38 000 occurrences of a handful of ids, all in one of sixteen parts of 16 384 keys, and every
        // repetition asked for "2.36 x" of a region that was at its limit (profiles/r05E_fuzz_tandem_repeat.txt)
        const bool no_parts = (all_positions && c->region_factor > 2.0) || c->parts_off;
        if (cap > big_min) {
            // a genome expected to fit the LDS sort keeps that path -- unless an earlier attempt has shown that this
            // batch emits far more than the sampling rate predicts (low-complexity sequence): then the factor decides
            if (c->region_factor <= 2.0 && (uint64_t)((double)pos * rate * 1.25) + 64 <= big_min) cap = big_min;
            else if (!(flags & KSSD_SKETCH_BY_POS) && cap <= ((uint64_t)big_min << DEDUP_MAX_PARTS_LOG2) && !no_parts) {
                // sorted in LDS in parts: ranges of the ids' top bits, narrow enough for the register sort where that is possible
                // (~4 096 expected tuples per part) and never wider than the LDS array
                uint32_t lg = 1;
                while (lg < DEDUP_MAX_PARTS_LOG2 && ((cap >> lg) > 8192 || (cap >> lg) > big_min)) lg++;
                c->h_med.push_back(make_uint2(g, lg));
                if ((cap >> lg) + 1 > med_part) med_part = (cap >> lg) + 1;
            } else {
                if (cap >= (1ull << 31)) { c->last_launch_rc = KSSD_ERR_UNSUPPORTED; return KSSD_ERR_UNSUPPORTED; }
                c->h_big.push_back(g);  // global-memory sort path
            }
        }
        c->h_reg_off[g] = acc;
        acc += cap;
        const bool is_med = !c->h_med.empty() && c->h_med.back().x == g;
        // (the per-genome workgroup's key array is 5/8 of the largest region, finish_sketch: a region that is all of its genome's
        // positions asks for an array that holds them all -- there is no larger region to ask for behind it)
        const uint64_t lds_cap = all_positions ? cap * 8 / 5 : cap;  // (rounded DOWN: 5/8 of it rounded up is `cap` again, never cap + 1 -- which doubled the array of a 16 384-position genome past the LDS)
        if (cap <= big_min && lds_cap > max_cap) max_cap = lds_cap;
        if (cap > big_min && !is_med && cap > max_big) max_big = cap;
    }
    c->h_reg_off[n_genomes] = acc;
    c->h_chunk_off.assign(h_chunk_off, h_chunk_off + n_genomes + 1);
    int rc;
    if (!c->h_med.empty()) {
        uint32_t pc = 1024;
        while (pc < med_part) pc <<= 1;
        if (pc > big_min) pc = big_min;
        c->med_part_cap = pc;
        const size_t nm = c->h_med.size();
        {
            const size_t before = c->cap_med;
            if ((rc = ensure(&c->d_med, &c->cap_med, nm)) != KSSD_OK) return rc;
            if (c->cap_med != before) c->dev_med.clear();
        }
        if ((rc = ensure(&c->d_med_cnt, &c->cap_med_cnt, nm << (DEDUP_MAX_PARTS_LOG2 + 2))) != KSSD_OK) return rc;
        const size_t out_bytes = (nm << DEDUP_MAX_PARTS_LOG2) * (size_t)pc * (with_pos ? 8 : 4);
        if (out_bytes > c->cap_med_out) {
            if (c->d_med_out) hipFree(c->d_med_out);
            c->d_med_out = nullptr;
            c->cap_med_out = 0;
            if (hipMalloc(&c->d_med_out, out_bytes + out_bytes / 4) != hipSuccess) return KSSD_ERR_NOMEM;
            c->cap_med_out = out_bytes + out_bytes / 4;
        }
    }
    if ((rc = ensure(&c->d_chunk_gid, &c->cap_chunks, (size_t)n_chunks + 1)) != KSSD_OK) return rc;
    {
        const size_t before = c->cap_chunk_off;
        if ((rc = ensure(&c->d_chunk_off, &c->cap_chunk_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
        if (c->cap_chunk_off != before) c->dev_chunk_off.clear();
        const size_t before2 = c->cap_reg_off;
        if ((rc = ensure(&c->d_reg_off, &c->cap_reg_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
        if (c->cap_reg_off != before2) c->dev_reg_off.clear();
    }
    if ((rc = ensure(&c->d_cursor, &c->cap_cursor, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_kept, &c->cap_kept, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_regions, &c->cap_regions, ((size_t)acc + 1) * (with_pos ? 2 : 1))) != KSSD_OK) return rc;
    // candidate list between the scan and the exact stage: patterns of S (both strands) + Bloom false positives
    // (one private slice per wave of the scan grid)
    const uint64_t want_blocks = (n_chunks + SCAN_WAVES - 1) / SCAN_WAVES;
    const uint64_t max_blocks = c->scan_grid_limit ? c->scan_grid_limit : (uint64_t)c->cu_count;
    const int grid = (int)(want_blocks < max_blocks ? want_blocks : max_blocks);
    const uint32_t n_slices = (uint32_t)(grid > 0 ? grid : 1) * SCAN_WAVES;
    uint64_t cand_cap = (uint64_t)((double)n_chunks * KSSD_CHUNK * (2.0 * rate + 0.0005) * c->cand_factor / n_slices) + 256;
    if (cand_cap < c->cand_floor) cand_cap = c->cand_floor;  // what the fullest slice of an overflowed attempt wanted
    if ((rc = ensure(&c->d_cand, &c->cap_cand, (size_t)cand_cap * n_slices * 2)) != KSSD_OK) return rc;  // 16-byte records
    if ((rc = ensure(&c->d_cand_count, &c->cap_cand_count, 2 * (size_t)n_slices)) != KSSD_OK) return rc;  // per slice: candidates listed | positions past stage 1
    if ((rc = ensure(&c->d_blk_info, &c->cap_blk_info, (size_t)((n_chunks + SCAN_BLOCK - 1) / SCAN_BLOCK) + 1)) != KSSD_OK) return rc;
    c->last_cand_cap = cand_cap;
    pl.big_min = big_min; pl.max_cap = max_cap; pl.max_big = max_big; pl.cand_cap = cand_cap; pl.n_slices = n_slices; pl.grid = grid;
    pl.valid = true;
    return KSSD_OK;
}

static int phase_prep(kssd_gpu_ctx *c, hipStream_t s)
{
    const auto &pl = c->plan;
    if (pl.n_genomes == 0) {
        HIPCK(hipMemsetAsync(c->d_status, 0, sizeof(SketchStatus), s));
        HIPCK(hipMemsetAsync(pl.d_out_off, 0, sizeof(uint64_t), s));
        return KSSD_OK;
    }
    const size_t nb = ((size_t)pl.n_genomes + 1) * 8;
    // the layout tables travel only when they differ from what the device holds (a stream of equally shaped batches
    // pays for them once)
    if (c->dev_chunk_off != c->h_chunk_off) {
        HIPCK(hipMemcpyAsync(c->d_chunk_off, c->h_chunk_off.data(), nb, hipMemcpyHostToDevice, s));
        c->dev_chunk_off = c->h_chunk_off;
    }
    if (c->dev_reg_off != c->h_reg_off) {
        HIPCK(hipMemcpyAsync(c->d_reg_off, c->h_reg_off.data(), nb, hipMemcpyHostToDevice, s));
        c->dev_reg_off = c->h_reg_off;
    }
    if (!c->h_med.empty()) {
        bool same = c->dev_med.size() == c->h_med.size();
        for (size_t i = 0; same && i < c->h_med.size(); i++) same = c->dev_med[i].x == c->h_med[i].x && c->dev_med[i].y == c->h_med[i].y;
        if (!same) {
            c->dev_med = c->h_med;  // (the copy reads dev_med: it stays as it is until the next plan that differs)
            HIPCK(hipMemcpyAsync(c->d_med, c->dev_med.data(), c->dev_med.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
        }
    }
    if (c->h_big.empty() && c->h_med.empty()) {
        // every genome goes through the fused per-genome kernel: nobody reads the chunk -> genome map or the cursors, the scan
        // writes every slice's counts and zeroes the status words itself -- nothing to launch (a batch without a chunk has no
        // scan: the status is cleared here)
        if (pl.n_chunks == 0) HIPCK(hipMemsetAsync(c->d_status, 0, sizeof(SketchStatus), s));
        return KSSD_OK;
    }
    uint64_t init_n = pl.n_chunks > pl.n_genomes ? pl.n_chunks : pl.n_genomes;
    if (init_n < pl.n_slices) init_n = pl.n_slices;
    if (init_n < sizeof(SketchStatus) / 4) init_n = sizeof(SketchStatus) / 4;
    if (pl.n_chunks >= 64ull * pl.n_genomes) {  // long genomes: fill the map genome by genome
        // enough workgroups per genome that a single huge one (a read set, a chromosome) is not written by one of them
        uint64_t parts = (pl.n_chunks / pl.n_genomes + 16 * 256 - 1) / (16 * 256);
        if (parts > 1024) parts = 1024;
        // (the grid's x is the genomes and nothing more: a read set is ONE genome, and 63 of 64 workgroups of the launch used to find
        // that they had none -- 58 000 empty workgroups, 19 of the kernel's 25 us at configs[3]; the per-call state is zeroed in loops)
        hipLaunchKernelGGL(chunk_gid_by_genome_kernel, dim3(pl.n_genomes, (unsigned)(parts ? parts : 1)), dim3(256), 0, s,
                           (const uint64_t *)c->d_chunk_off, pl.n_genomes, c->d_chunk_gid, c->d_cursor, c->d_cand_count, pl.n_slices,
                           reinterpret_cast<uint32_t *>(c->d_status));
        HIPCK(hipGetLastError());
        return KSSD_OK;
    }
    hipLaunchKernelGGL(chunk_gid_kernel, dim3((unsigned)((init_n + 255) / 256)), dim3(256), 0, s,
                       (const uint64_t *)c->d_chunk_off, pl.n_genomes, pl.n_chunks, c->d_chunk_gid, c->d_cursor, c->d_cand_count,
                       pl.n_slices, reinterpret_cast<uint32_t *>(c->d_status));
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

#ifdef KSSD_DEV
static unsigned long long *g_dev_times;
// development build only (scanbench): the per-wave time stamps of the last scan, 3 per wave
extern "C" int kssd_gpu_dev_wavetimes(unsigned long long *out, uint32_t n_waves)
{
    if (!g_dev_times) return KSSD_ERR_PARAM;
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy(out, g_dev_times, (size_t)n_waves * 3 * 8, hipMemcpyDeviceToHost));
    return KSSD_OK;
}
#endif

static int phase_scan(kssd_gpu_ctx *c, hipStream_t s)
{
    const auto &pl = c->plan;
    if (pl.n_genomes == 0 || pl.n_chunks == 0) return KSSD_OK;
    int rc;
    ScanArgs a;
    a.packed = pl.d_packed; a.mask = pl.d_mask; a.n_chunks = pl.n_chunks; a.tab = c->d_T1;
    // (fastq2co semantics = a read set: four lanes in ten hold a read's end, their mask loads are issued in every chunk all the same and
    // cost the scan more than the stream they replace -- 1.52 against 1.47 ms at configs[3], profiles/r06c_*; the exact-evaluation kernel
    // asks the summary words either way)
    a.summ = (pl.flags & KSSD_SKETCH_KEEP_ZERO) ? nullptr : reinterpret_cast<const uint32_t *>(pl.d_summ);
    a.cand = reinterpret_cast<ulonglong2 *>(c->d_cand); a.cand_cap = pl.cand_cap; a.cand_count = c->d_cand_count;
    a.rec8 = (c->h_big.empty() && c->h_med.empty()) ? 1u : 0u;  // (= scanned.fused below: the FINISH phase evaluates the candidates itself)
    a.stage1_count = c->d_cand_count + pl.n_slices;
    a.blk_info = c->d_blk_info;
    a.status = c->d_status;
#ifdef KSSD_DEV
    {
        static unsigned long long *d_times = nullptr;
        if (!d_times && getenv("KSSD_DEV_WAVETIME")) hipMalloc(&d_times, 4096 * 16 * 3 * 8);
        a.dev_times = d_times;
        g_dev_times = d_times;
    }
#endif
    const int grid = pl.grid;
    const unsigned evi = c->ev_n[0] % EV_RING;
    const bool timed = kernel_timed(c, 0);
    hipEvent_t e0 = timed ? c->ev_a[0][evi] : nullptr, e1 = timed ? c->ev_b[0][evi] : nullptr;
    switch (c->P.subk) {
    case 2: rc = launch_scan<2>(c, a, grid, s, e0, e1); break;
    case 3: rc = launch_scan<3>(c, a, grid, s, e0, e1); break;
    case 4: rc = launch_scan<4>(c, a, grid, s, e0, e1); break;
    case 5: rc = launch_scan<5>(c, a, grid, s, e0, e1); break;
    case 6: {
#ifdef KSSD_DEV  // development build only (libkssd_gpu_dev.so, `make tools`): ablated scans for profiles/scanbench
        static const int abl = getenv("KSSD_DEV_ABLATE") ? atoi(getenv("KSSD_DEV_ABLATE")) : 0;
        if (abl == 1) { rc = launch_scan<6, 1>(c, a, grid, s, e0, e1); break; }
        if (abl == 2) { rc = launch_scan<6, 2>(c, a, grid, s, e0, e1); break; }
        if (abl == 3) { rc = launch_scan<6, 3>(c, a, grid, s, e0, e1); break; }
#endif
        rc = launch_scan<6>(c, a, grid, s, e0, e1);
        break;
    }
    case 7: rc = launch_scan<7>(c, a, grid, s, e0, e1); break;
    default: rc = KSSD_ERR_UNSUPPORTED;
    }
    if (rc != KSSD_OK) return rc;
    if (timed) c->ev_n[0]++;
    c->scanned.valid = true;  // what a further tuple pass (KSSD_PHASE_REPASS) must find unchanged
    c->scanned.cand = c->d_cand;
    c->scanned.blk_info = c->d_blk_info;
    c->scanned.cand_cap = pl.cand_cap;
    c->scanned.n_slices = pl.n_slices;
    c->scanned.n_chunks = pl.n_chunks;
    c->scanned.fused = c->h_big.empty() && c->h_med.empty();
    c->scanned.reg_off = c->h_reg_off;
    c->scanned.med = c->h_med;
    c->scanned.big = c->h_big;
    return KSSD_OK;
}

// does the plan in place describe the batch, the candidate list and the staging layout the last scan ran with?
static bool repass_matches_scan(const kssd_gpu_ctx *c)
{
    const auto &pl = c->plan;
    const auto &sc = c->scanned;
    if (!pl.valid || !sc.valid || sc.cand != c->d_cand || sc.blk_info != c->d_blk_info || sc.cand_cap != pl.cand_cap ||
        sc.n_slices != pl.n_slices || sc.n_chunks != pl.n_chunks || sc.reg_off != c->h_reg_off || sc.big != c->h_big ||
        sc.med.size() != c->h_med.size())
        return false;
    for (size_t i = 0; i < sc.med.size(); i++)
        if (sc.med[i].x != c->h_med[i].x || sc.med[i].y != c->h_med[i].y) return false;
    return true;
}

static int phase_exact(kssd_gpu_ctx *c, hipStream_t s)
{
    const auto &pl = c->plan;
    if (pl.n_genomes == 0 || pl.n_chunks == 0) return KSSD_OK;
    if (c->h_big.empty() && c->h_med.empty()) return KSSD_OK;  // the FINISH phase evaluates the candidates itself (sketch_dedup_kernel<K, DEDUP_FUSED>)
    ExactArgs x;
    x.packed = pl.d_packed; x.mask = pl.d_mask; x.chunk_gid = c->d_chunk_gid;
    x.summ = reinterpret_cast<const unsigned long long *>(pl.d_summ);
    x.chunk_off = (const unsigned long long *)c->d_chunk_off; x.G = c->d_G; x.gfilt = c->d_gfilt;
    x.cand = reinterpret_cast<const ulonglong2 *>(c->d_cand); x.cand_cap = pl.cand_cap; x.cand_count = c->d_cand_count;
    x.carry = kssd_carry_ok(c->P) ? 1u : 0u;
    x.n_slices = pl.n_slices;
    x.reg_off = (const unsigned long long *)c->d_reg_off; x.cursor = c->d_cursor; x.regions = c->d_regions;
    x.by_pos = (pl.flags & KSSD_SKETCH_BY_POS) ? 1u : 0u;
    x.one_genome = pl.n_genomes == 1 ? 1u : 0u;
    x.cand_cap_end = (unsigned long long)pl.n_chunks * KSSD_CHUNK;
#ifdef KSSD_DEV
    x.dev_no_atomic = getenv("KSSD_DEV_EXACT_NO_ATOMIC") ? (uint32_t)atoi(getenv("KSSD_DEV_EXACT_NO_ATOMIC")) : 0u;  // (1: no cursor atomic, 2: no exact-table read, 3: no staging, 4: nothing)
#endif
    x.status = c->d_status;
    const dim3 grid((unsigned)((pl.cand_cap + EXACT_THREADS * EXACT_PER - 1) / (EXACT_THREADS * EXACT_PER)), pl.n_slices);
    if (pl.with_pos) hipLaunchKernelGGL((sketch_exact_kernel<unsigned long long>), grid, dim3(EXACT_THREADS), 0, s, c->P, x);
    else hipLaunchKernelGGL((sketch_exact_kernel<uint32_t>), grid, dim3(EXACT_THREADS), 0, s, c->P, x);
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

static int phase_finish(kssd_gpu_ctx *c, hipStream_t s)
{
    const auto &pl = c->plan;
    if (pl.n_genomes == 0) return KSSD_OK;
    uint32_t flags = pl.flags;
    if (c->region_factor > 2.0) flags |= SKETCH_TRACK_FILL;
    // by-position keys carry the position in the upper half: the gather's "id" is the position, its "position" the tuple
    uint32_t *ids_to = (flags & KSSD_SKETCH_BY_POS) ? c->d_out_pos : pl.d_out_ids;
    uint32_t *pos_to = (flags & KSSD_SKETCH_BY_POS) ? pl.d_out_ids : c->d_out_pos;
    int rc = pl.with_pos ? finish_sketch<unsigned long long>(c, pl.n_genomes, flags, pl.min_occ, pl.big_min, pl.max_cap, pl.max_big,
                                                             pl.d_out_off, ids_to, pos_to, pl.out_cap, s)
                         : finish_sketch<uint32_t>(c, pl.n_genomes, flags, pl.min_occ, pl.big_min, pl.max_cap, pl.max_big, pl.d_out_off,
                                                   pl.d_out_ids, nullptr, pl.out_cap, s);
    if (rc != KSSD_OK) return rc;
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

// the per-call state a further tuple pass starts from: cursors and the status words of the stages behind the scan
__global__ void repass_reset_kernel(uint32_t *__restrict__ cursor, uint32_t n_genomes, SketchStatus *st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_genomes) cursor[i] = 0;
    if (i < sizeof(SketchStatus) / 4) reinterpret_cast<uint32_t *>(st)[i] = 0;  // (the scan's totals are added up again by the stage that follows)
}

extern "C" int kssd_gpu_sketch_phase(kssd_gpu_ctx *c, int phase, void *stream)
{
    if (!c || !c->plan.valid) return KSSD_ERR_PARAM;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipSetDevice(c->device));
    switch (phase) {
    case KSSD_PHASE_PREP: return phase_prep(c, s);
    case KSSD_PHASE_SCAN: return phase_scan(c, s);
    case KSSD_PHASE_EXACT: return phase_exact(c, s);
    case KSSD_PHASE_FINISH: return phase_finish(c, s);
    case KSSD_PHASE_REPASS:
        if (!repass_matches_scan(c)) return KSSD_ERR_PARAM;  // the plan is not the scanned batch's any more: PREP + SCAN instead
        if (c->plan.n_genomes)
            hipLaunchKernelGGL(repass_reset_kernel, dim3((c->plan.n_genomes + 255) / 256), dim3(256), 0, s, c->d_cursor, c->plan.n_genomes, c->d_status);
        HIPCK(hipGetLastError());
        return KSSD_OK;
    default: return KSSD_ERR_PARAM;
    }
}

extern "C" int kssd_gpu_sketch_device(kssd_gpu_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask,
                                      const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                      uint64_t *d_out_off, uint32_t *d_out_ids, uint64_t out_cap, void *stream)
{
    int rc = kssd_gpu_sketch_plan(c, d_packed, d_mask, h_chunk_off, n_genomes, flags, min_occ, d_out_off, d_out_ids, out_cap);
    for (int ph = KSSD_PHASE_PREP; rc == KSSD_OK && ph <= KSSD_PHASE_FINISH; ph++) rc = kssd_gpu_sketch_phase(c, ph, stream);
    return rc;
}

extern "C" int kssd_gpu_sketch_set_pos_output(kssd_gpu_ctx *c, uint32_t *d_out_pos)
{
    if (!c) return KSSD_ERR_PARAM;
    c->d_out_pos = d_out_pos;
    return KSSD_OK;
}

extern "C" int kssd_gpu_set_lds_sort_limit(kssd_gpu_ctx *c, uint32_t max_tuples)
{
    if (!c) return KSSD_ERR_PARAM;
    c->lds_sort_limit = max_tuples;
    return KSSD_OK;
}

// k - drlevel = 9 (36-bit tuples): which of the 2^4 passes the following sketch calls make (kssd_core.h KssdParams::pass_bits)
extern "C" uint32_t kssd_gpu_tuple_passes(const kssd_gpu_ctx *c) { return c ? 1u << c->P.pass_bits : 0u; }
extern "C" int kssd_gpu_set_tuple_pass(kssd_gpu_ctx *c, uint32_t pass)
{
    if (!c || pass >= (1u << c->P.pass_bits)) return KSSD_ERR_PARAM;
    c->P.pass = pass;
    return KSSD_OK;
}

extern "C" int kssd_gpu_set_scan_grid(kssd_gpu_ctx *c, uint32_t max_workgroups)
{
    if (!c) return KSSD_ERR_PARAM;
    c->scan_grid_limit = max_workgroups;
    return KSSD_OK;
}

extern "C" int kssd_gpu_sketch_status(kssd_gpu_ctx *c, uint64_t *total_ids, int64_t *bad_genome, void *stream)
{
    if (!c) return KSSD_ERR_PARAM;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipSetDevice(c->device));
    if (bad_genome) *bad_genome = -1;
    if (total_ids) *total_ids = 0;
    if (c->last_launch_rc != KSSD_OK) return c->last_launch_rc;
    SketchStatus st;
    HIPCK(hipMemcpyAsync(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    if (total_ids) *total_ids = st.total_ids;
#ifdef KSSD_DEV
    if (getenv("KSSD_DEV_TRACE"))
        fprintf(stderr, "[kssd_gpu] status: ids %llu, stage1 %llu, bloom %llu, cand_overflow %u (need %u of %llu), region_overflow %u "
                        "(need x%.2f, factor %.2f), out_overflow %u, capacity %u\n",
                (unsigned long long)st.total_ids, (unsigned long long)st.n_stage1, (unsigned long long)st.n_bloom, st.cand_overflow,
                st.cand_need, (unsigned long long)c->last_cand_cap, st.region_overflow, st.max_need_q8 / 256.0, c->region_factor,
                st.out_overflow, st.capacity_genome_p1);
#endif
    if (st.lb_timeout) {
        // the per-genome kernel's look-back relies on workgroups being dispatched in the order of their numbers; a workgroup that waited
        // ~a second for a genome in front of it says so instead of hanging the launch
        snprintf(g_hip_err, sizeof g_hip_err, "sketch_dedup_kernel: a workgroup's look-back timed out (workgroups not dispatched in order?)");
        return KSSD_ERR_HIP;
    }
    if (st.cand_overflow) {
        c->cand_floor = (uint64_t)st.cand_need + st.cand_need / 4 + 64;
        return KSSD_ERR_OVERFLOW;
    }
    if (st.parts_skew) {
        // (without this switch the factor had to grow until the region left the parts' range: x 1.39 x 1.25 per attempt from 23 to 770 for
        // a 16 kb repeat under a test's LDS limit of 256 keys -- more attempts than callers make, profiles/r06long_fuzz_sketch.txt)
        c->parts_off = true;
        c->parts_off_calls = 4;
        if (!st.region_overflow && !st.ranges_skew) return KSSD_ERR_OVERFLOW;
    }
    if (st.ranges_skew && !st.region_overflow) {
        c->ranges_off = true;
        c->ranges_off_calls = 4;  // the repeated call and the further passes callers make over the same batch (occurrences, abundances)
        return KSSD_ERR_OVERFLOW;
    }
    if (st.region_overflow) {
        double need = (double)st.max_need_q8 / 256.0;  // emitted / capacity of the worst genome
        if (need > 4096.0) need = 4096.0;              // (no genome emits more than one tuple per position; a region that is all positions grows no further)
        c->region_factor *= (need > 1.0 ? need : 1.0) * 1.25;
        if (c->region_factor > 1e9) c->region_factor = 1e9;
        return KSSD_ERR_OVERFLOW;
    }
    if (st.out_overflow) return KSSD_ERR_OVERFLOW;
    // (36-bit tuples: a pass stages a sixteenth of what the regions are sized for, and the passes 1 .. 15 of a batch run on the
    // layout tables of its scan -- nothing about the layout may move between them: no shrinking, and only pass 0 counts a call)
    if (c->region_factor > 2.0 && st.max_need_q8 < 64 && c->P.pass_bits == 0) {
        // the regions were grown for a batch that emitted far more than the sampling rate predicts; this batch filled
        // its fullest region to less than a quarter: shrink towards the default again
        const double f = c->region_factor * ((double)st.max_need_q8 / 256.0) * 2.0;
        c->region_factor = f > 2.0 ? f : 2.0;
    }
    if (c->ranges_off && c->ranges_off_calls && c->P.pass == 0 && --c->ranges_off_calls == 0) c->ranges_off = false;  // later batches try the ranges again
    if (c->parts_off && c->parts_off_calls && c->P.pass == 0 && --c->parts_off_calls == 0) c->parts_off = false;
    if (st.capacity_genome_p1) {
        if (bad_genome) *bad_genome = (int64_t)(0xFFFFFFFFu - st.capacity_genome_p1);
        return KSSD_ERR_CAPACITY;
    }
    return KSSD_OK;
}

extern "C" int kssd_gpu_scan_stats(kssd_gpu_ctx *c, uint64_t *stage1, uint64_t *bloom, void *stream)
{
    if (!c) return KSSD_ERR_PARAM;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipSetDevice(c->device));
    SketchStatus st;
    HIPCK(hipMemcpyAsync(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    if (stage1) *stage1 = st.n_stage1;
    if (bloom) *bloom = st.n_bloom;
    return KSSD_OK;
}

// the batch is in the context's input buffers (d_in_packed / d_in_mask): sketch it and bring the CSR to the host
// again = true: the batch of the last call once more (a further tuple pass): its scan's candidates are still there
static int sketch_resident_impl(kssd_gpu_ctx *c, const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                uint64_t **out_off, uint32_t **out_ids, uint32_t **out_pos, int64_t *bad_genome, bool again = false)
{
    hipStream_t s = c->own_stream;
    const uint64_t n_chunks = chunk_off[n_genomes];
    int rc;
    c->results_valid = false;  // (a call that fails leaves nothing a later kssd_gpu_resident_put may copy)
    if ((rc = ensure(&c->d_b_off, &c->cap_b_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if (out_pos && !(flags & (KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS))) flags |= KSSD_SKETCH_FIRST_POS;
    if (!out_pos) flags &= ~(KSSD_SKETCH_FIRST_POS | KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS);
    const double rate = (double)c->P.dim_end / (double)(1ull << (4 * c->P.subk));
    uint64_t out_cap = (uint64_t)((double)n_chunks * KSSD_CHUNK * rate * 1.5) + 1024;
    uint64_t total = 0;
    const bool have_summ = n_chunks && (c->in_summ_valid || c->summ_auto);
    if (have_summ && !c->in_summ_valid) {
        if ((rc = ensure(&c->d_in_summ, &c->cap_in_summ, (size_t)n_chunks)) != KSSD_OK) return rc;
        if ((rc = kssd_gpu_mask_summarise_device(c, c->d_in_mask, n_chunks, c->d_in_summ, s)) != KSSD_OK) return rc;
    }
    for (int attempt = 0; attempt < 12; attempt++) {
        if (have_summ) c->d_next_summ = c->d_in_summ;
        if ((rc = ensure(&c->d_b_ids, &c->cap_b_ids, (size_t)out_cap)) != KSSD_OK) break;
        if (out_pos) {
            if ((rc = ensure(&c->d_b_pos, &c->cap_b_pos, (size_t)out_cap)) != KSSD_OK) break;
            c->d_out_pos = c->d_b_pos;
        }
        if (again && attempt == 0) {  // (a retry after an overflow scans again: its workspaces may have moved)
            rc = kssd_gpu_sketch_plan(c, c->d_in_packed, c->d_in_mask, chunk_off, n_genomes, flags, min_occ, c->d_b_off, c->d_b_ids, out_cap);
            // the pass runs on the candidate list and the layout tables of the batch's scan: a plan that differs from that
            // scan's (the staging layout has moved since) scans again
            const bool same = rc == KSSD_OK && repass_matches_scan(c);
            for (int ph : {same ? KSSD_PHASE_REPASS : KSSD_PHASE_PREP, same ? -1 : KSSD_PHASE_SCAN, KSSD_PHASE_EXACT, KSSD_PHASE_FINISH})
                if (rc == KSSD_OK && ph >= 0) rc = kssd_gpu_sketch_phase(c, ph, s);
        } else {
            rc = kssd_gpu_sketch_device(c, c->d_in_packed, c->d_in_mask, chunk_off, n_genomes, flags, min_occ, c->d_b_off, c->d_b_ids,
                                        out_cap, s);
        }
        if (rc != KSSD_OK) break;
        rc = kssd_gpu_sketch_status(c, &total, bad_genome, s);
        if (rc != KSSD_ERR_OVERFLOW) break;
        if (total > out_cap) out_cap = total + 1024;
    }
    c->d_out_pos = nullptr;
    if (rc != KSSD_OK) return rc;
    uint64_t *h_off = (uint64_t *)malloc(((size_t)n_genomes + 1) * 8);
    uint32_t *h_ids = (uint32_t *)malloc((size_t)(total ? total : 1) * 4);
    uint32_t *h_pos = out_pos ? (uint32_t *)malloc((size_t)(total ? total : 1) * 4) : nullptr;
    if (!h_off || !h_ids || (out_pos && !h_pos)) { free(h_off); free(h_ids); free(h_pos); return KSSD_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(h_off, c->d_b_off, ((size_t)n_genomes + 1) * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && total) e = hipMemcpyAsync(h_ids, c->d_b_ids, (size_t)total * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && total && out_pos) e = hipMemcpyAsync(h_pos, c->d_b_pos, (size_t)total * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { free(h_off); free(h_ids); free(h_pos); return hip_fail(e, "sketch results to host", __LINE__); }
    *out_off = h_off;
    *out_ids = h_ids;
    if (out_pos) *out_pos = h_pos;
    c->resident_valid = true;
    c->results_valid = true;
    return KSSD_OK;
}

// The batch of the context's LAST host-level sketch call (kssd_gpu_sketch_batch[_pos], kssd_gpu_sketch_fast[aq]_text) once
// more without scanning it again -- for the tuple passes 1 .. 15 of k - drlevel = 9 after kssd_gpu_set_tuple_pass (same flags
// and min_occ as that call; out_pos NULL like there)
extern "C" int kssd_gpu_sketch_again(kssd_gpu_ctx *c, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids, uint32_t **out_pos,
                                     int64_t *bad_genome)
{
    if (!c || !out_off || !out_ids || c->h_chunk_off.size() != (size_t)c->last_n_genomes + 1 || !c->resident_valid) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    *out_off = nullptr;
    *out_ids = nullptr;
    if (out_pos) *out_pos = nullptr;
    if (bad_genome) *bad_genome = -1;
    const std::vector<uint64_t> co = c->h_chunk_off;  // (the plan rewrites the context's copy)
    return sketch_resident_impl(c, co.data(), c->last_n_genomes, flags, min_occ, out_off, out_ids, out_pos, bad_genome, true);
}



// Host-level sketch call.  The device-side input and output buffers belong to the context and only ever grow, the
// work runs on the context's own stream, and a caller that keeps its batch in page-locked memory (kssd_gpu_host_alloc)
// gets plain DMA transfers: a sequence of calls then costs transfers and kernels, not allocations.
static int sketch_batch_impl(kssd_gpu_ctx *c, const uint32_t *packed, const uint32_t *mask, const uint64_t *chunk_off,
                             uint32_t n_genomes, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                             uint32_t **out_pos, int64_t *bad_genome)
{
    if (!c || !chunk_off || !out_off || !out_ids) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    *out_off = nullptr;
    *out_ids = nullptr;
    if (out_pos) *out_pos = nullptr;
    hipStream_t s = c->own_stream;
    const uint64_t n_chunks = chunk_off[n_genomes];
    const size_t pw = (size_t)n_chunks * KSSD_CHUNK_WORDS, mw = (size_t)n_chunks * KSSD_CHUNK_MASKW;
    int rc;
    if ((rc = ensure(&c->d_in_packed, &c->cap_in_packed, pw + KSSD_PACK_SLACK_WORDS)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_in_mask, &c->cap_in_mask, mw + KSSD_PACK_SLACK_WORDS)) != KSSD_OK) return rc;
    if (n_chunks) {
        HIPCK(hipMemcpyAsync(c->d_in_packed, packed, pw * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(c->d_in_mask, mask, mw * 4, hipMemcpyHostToDevice, s));
    }
    HIPCK(hipMemsetAsync(c->d_in_packed + pw, 0, KSSD_PACK_SLACK_WORDS * 4, s));  // the slack behind the last chunk: no bases
    HIPCK(hipMemsetAsync(c->d_in_mask + mw, 0, KSSD_PACK_SLACK_WORDS * 4, s));
    c->in_summ_valid = false;  // (a batch packed on the host comes without summary words)
    return sketch_resident_impl(c, chunk_off, n_genomes, flags, min_occ, out_off, out_ids, out_pos, bad_genome);
}

// page-locked host memory for batches that travel by DMA (hipHostMalloc); NULL when it cannot be had
extern "C" void *kssd_gpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

// memory of the caller's own (malloc) made page-locked in place / released again: what a command does with the inputs it read while
// the runtime was still starting -- hipHostMalloc was not to be had yet, and a copy to the device out of ordinary memory goes
// through the runtime's staging buffers at a fraction of the PCIe rate (profiles/r05i: 7 GB/s against 55)
extern "C" int kssd_gpu_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return KSSD_ERR_PARAM;
    HIPCK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return KSSD_OK;
}

extern "C" int kssd_gpu_host_unregister(void *p)
{
    if (!p) return KSSD_ERR_PARAM;
    HIPCK(hipHostUnregister(p));
    return KSSD_OK;
}

extern "C" void kssd_gpu_host_free(void *p)
{
    if (p) hipHostFree(p);
}

extern "C" int kssd_gpu_sketch_batch(kssd_gpu_ctx *c, const uint32_t *packed, const uint32_t *mask,
                                     const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                     uint64_t **out_off, uint32_t **out_ids, int64_t *bad_genome)
{
    return sketch_batch_impl(c, packed, mask, chunk_off, n_genomes, flags, min_occ, out_off, out_ids, nullptr, bad_genome);
}

extern "C" int kssd_gpu_sketch_batch_pos(kssd_gpu_ctx *c, const uint32_t *packed, const uint32_t *mask,
                                         const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                         uint64_t **out_off, uint32_t **out_ids, uint32_t **out_pos, int64_t *bad_genome)
{
    if (!out_pos) return KSSD_ERR_PARAM;
    return sketch_batch_impl(c, packed, mask, chunk_off, n_genomes, flags, min_occ, out_off, out_ids, out_pos, bad_genome);
}

extern "C" int kssd_gpu_set_kernel_timing(kssd_gpu_ctx *c, uint32_t every)
{
    if (!c) return KSSD_ERR_PARAM;
    for (int w = 0; w < EV_KINDS; w++) { c->ev_every[w] = every; c->ev_launch[w] = 0; }
    return KSSD_OK;
}

extern "C" int kssd_gpu_kernel_time(kssd_gpu_ctx *c, int which, int reset, float *avg_ms, uint32_t *launches)
{
    if (!c || which < 0 || which >= EV_KINDS) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    const unsigned n = c->ev_n[which] < EV_RING ? c->ev_n[which] : EV_RING;
    double sum = 0;
    for (unsigned i = 0; i < n; i++) {
        float ms = 0;
        HIPCK(hipEventSynchronize(c->ev_b[which][i]));
        HIPCK(hipEventElapsedTime(&ms, c->ev_a[which][i], c->ev_b[which][i]));
        sum += ms;
    }
    if (avg_ms) *avg_ms = n ? (float)(sum / n) : 0.f;
    if (launches) *launches = n;
    if (reset) c->ev_n[which] = 0;
    return KSSD_OK;
}

// the durations one by one (ms[0 .. min(*launches, cap)): the ring's launches, oldest first): for the minimum / maximum a line
// prints beside the mean
extern "C" int kssd_gpu_kernel_times(kssd_gpu_ctx *c, int which, float *ms, uint32_t cap, uint32_t *launches)
{
    if (!c || which < 0 || which >= EV_KINDS || (!ms && cap)) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    const unsigned n = c->ev_n[which] < EV_RING ? c->ev_n[which] : EV_RING;
    for (unsigned i = 0; i < n && i < cap; i++) {
        HIPCK(hipEventSynchronize(c->ev_b[which][i]));
        HIPCK(hipEventElapsedTime(&ms[i], c->ev_a[which][i], c->ev_b[which][i]));
    }
    if (launches) *launches = n;
    return KSSD_OK;
}

// ---------------------------------------------------------------------------------------------------
// the FASTA tokeniser on the device lives in kssd_tok.inc
// ---------------------------------------------------------------------------------------------------
#include "kssd_tok.inc"

// host-level: FASTA texts (HOST bytes, ideally page-locked) in, sketches out -- text to the device, tokenised there
static int sketch_text_impl(kssd_gpu_ctx *c, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_len, uint32_t n_files,
                            uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids, uint32_t **out_pos, int64_t *bad_genome,
                            bool fq, uint64_t *h_lines)
{
    if (!c || !out_off || !out_ids || (n_files && (!text_off || !text_len))) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    *out_off = nullptr;
    *out_ids = nullptr;
    if (out_pos) *out_pos = nullptr;
    if (bad_genome) *bad_genome = -1;
    hipStream_t s = c->own_stream;
    std::vector<uint64_t> chunk_off((size_t)n_files + 1, 0);
    uint64_t text_end = 0;
    for (uint32_t f = 0; f < n_files; f++) {
        chunk_off[f + 1] = chunk_off[f] + (text_len[f] + KSSD_CHUNK - 1) / KSSD_CHUNK;
        if (text_off[f] + text_len[f] > text_end) text_end = text_off[f] + text_len[f];
    }
    const uint64_t n_chunks = chunk_off[n_files];
    int rc;
    if (text) {
        if ((rc = ensure(&c->d_text, &c->cap_text, (size_t)text_end + 64)) != KSSD_OK) return rc;
    } else if (text_end && (!c->d_text || c->cap_text < (size_t)text_end)) {
        return KSSD_ERR_PARAM;  // text == NULL: the bytes are in the context's buffer already (kssd_gpu_text_put)
    }
    if ((rc = ensure(&c->d_in_packed, &c->cap_in_packed, (size_t)n_chunks * KSSD_CHUNK_WORDS + KSSD_PACK_SLACK_WORDS)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_in_mask, &c->cap_in_mask, (size_t)n_chunks * KSSD_CHUNK_MASKW + KSSD_PACK_SLACK_WORDS)) != KSSD_OK) return rc;
    if (text_end && text) HIPCK(hipMemcpyAsync(c->d_text, text, (size_t)text_end, hipMemcpyHostToDevice, s));
    // FASTA: the one-pass tokeniser writes the mask's summary words with the mask, and the scan of this batch reads them instead of
    // streaming the mask (KSSD_TOK_NO_SUMMARY=1: without, A/B)
    static const bool tok_summ = !getenv("KSSD_TOK_NO_SUMMARY") && !getenv("KSSD_TOK_TWO_PASS");
    c->in_summ_valid = false;
    uint64_t *d_summ = nullptr;
    if (!fq && tok_summ && n_chunks) {
        if ((rc = ensure(&c->d_in_summ, &c->cap_in_summ, (size_t)n_chunks)) != KSSD_OK) return rc;
        d_summ = c->d_in_summ;
    }
    rc = tokenise_device_impl(c, c->d_text, text_off, text_len, n_files, c->d_in_packed, c->d_in_mask, chunk_off.data(), s, fq, d_summ);
    if (rc != KSSD_OK) return rc;
    c->in_summ_valid = d_summ != nullptr;
    rc = tokenise_status_impl(c, bad_genome, nullptr, fq ? h_lines : nullptr, s);
    if (rc != KSSD_OK) return rc;
    return sketch_resident_impl(c, chunk_off.data(), n_files, flags, min_occ, out_off, out_ids, out_pos, bad_genome);
}

extern "C" int kssd_gpu_sketch_fasta_text(kssd_gpu_ctx *c, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_len,
                                          uint32_t n_files, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                                          uint32_t **out_pos, int64_t *bad_genome)
{
    return sketch_text_impl(c, text, text_off, text_len, n_files, flags, min_occ, out_off, out_ids, out_pos, bad_genome, false, nullptr);
}

// The same for FASTQ read sets, one per input (fastq2co with -Q 0).  h_lines (nullable, n_files entries) receives the
// line count the reference reports per file.  KSSD_ERR_UNSUPPORTED with the input's index in bad_genome when an input needs
// the host tokeniser (see kssd_gpu_tokenise_fastq_device); nothing has been sketched then.
extern "C" int kssd_gpu_sketch_fastq_text(kssd_gpu_ctx *c, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_len,
                                          uint32_t n_files, uint32_t flags, uint32_t min_occ, uint64_t **out_off, uint32_t **out_ids,
                                          uint32_t **out_pos, uint64_t *h_lines, int64_t *bad_genome)
{
    return sketch_text_impl(c, text, text_off, text_len, n_files, flags, min_occ, out_off, out_ids, out_pos, bad_genome, true, h_lines);
}

// Streaming a long input in: the context's device text buffer is reserved once, slices are copied in from (page-locked)
// host memory asynchronously on the context's stream, and kssd_gpu_sketch_fast[aq]_text is called with text == NULL.
// The host side needs no buffer of the file's size: a few slices that are refilled as soon as their copy has left.
// (growing keeps what has been put so far: an input whose size is not known beforehand -- a gzip'ed file -- reserves an
// estimate and asks for more when the estimate runs out)
extern "C" int kssd_gpu_text_reserve(kssd_gpu_ctx *c, uint64_t bytes)
{
    if (!c) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->own_stream));
    const size_t need = (size_t)bytes + 64;
    if (c->d_text && c->cap_text >= need) return KSSD_OK;
    uint8_t *grown = nullptr;
    const size_t n = need + need / 4;
    if (hipMalloc(&grown, n) != hipSuccess) return KSSD_ERR_NOMEM;
    if (c->d_text) {
        if (hipMemcpy(grown, c->d_text, c->cap_text, hipMemcpyDeviceToDevice) != hipSuccess) { hipFree(grown); return KSSD_ERR_HIP; }
        hipFree(c->d_text);
    }
    c->d_text = grown;
    c->cap_text = n;
    return KSSD_OK;
}

// returns the copy's ticket (>= 0) for kssd_gpu_text_wait, or an error (< 0)
extern "C" int64_t kssd_gpu_text_put(kssd_gpu_ctx *c, uint64_t dst_off, const void *src, uint64_t n)
{
    if (!c || !src || !c->d_text || dst_off + n > c->cap_text) return KSSD_ERR_PARAM;
    if (hipSetDevice(c->device) != hipSuccess) return KSSD_ERR_HIP;
    const uint64_t t = c->text_puts;
    hipEvent_t &e = c->text_ev[t & 31u];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return KSSD_ERR_HIP;
    if (n && hipMemcpyAsync(c->d_text + dst_off, src, (size_t)n, hipMemcpyHostToDevice, c->own_stream) != hipSuccess) return KSSD_ERR_HIP;
    if (hipEventRecord(e, c->own_stream) != hipSuccess) return KSSD_ERR_HIP;
    c->text_puts = t + 1;
    return (int64_t)t;
}

// blocks until the copy with this ticket has read its source (for a ticket older than the last 32 it waits for the oldest
// of those: the copies of a stream run in order)
extern "C" int kssd_gpu_text_wait(kssd_gpu_ctx *c, int64_t ticket)
{
    if (!c || ticket < 0 || (uint64_t)ticket >= c->text_puts) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    const uint64_t newest = c->text_puts - 1;
    const uint64_t t = newest - (uint64_t)ticket >= 32 ? newest - 31 : (uint64_t)ticket;  // its slot was reused: wait for the oldest kept
    HIPCK(hipEventSynchronize(c->text_ev[t & 31u]));
    return KSSD_OK;
}

// ---------------------------------------------------------------------------------------------------
// distance path (index build + row kernel) lives in kssd_dist.inc
// ---------------------------------------------------------------------------------------------------
#include "kssd_dist.inc"

// ---------------------------------------------------------------------------------------------------
// set operations on sketches (kssd set) live in kssd_set.inc
// ---------------------------------------------------------------------------------------------------
#include "kssd_set.inc"
#include "kssd_xchg.inc"
#include "kssd_resident.inc"
