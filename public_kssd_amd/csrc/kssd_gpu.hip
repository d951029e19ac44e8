// kssd_gpu.hip -- gfx950 (MI355X / CDNA4) kernels and the C ABI of libkssd_gpu.so.
//
// Hot path of kssd: genome sketching (reference iseq2comem.c:188-356) and the
// shared-k-mer intersection + distances (co2mco.c:25-77, command_dist.c:763-790,1251-1266).
// Integer / indexing work: HBM streaming, LDS tables, wavefront ballot/scan.  No MFMA on purpose.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC kssd_gpu.hip -o libkssd_gpu.so
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>
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
    unsigned int ranges_skew;      // a large genome's keys do not spread over its id ranges (one id tens of thousands of times, crafted
                                   // ids): the call is repeated with the global-memory sort for large genomes
};

struct kssd_gpu_ctx {
    int device;
    int cu_count;
    hipStream_t own_stream;  // non-blocking stream of the host-level calls (contexts on different host threads do not meet on the null stream)
    bool dist_only;  // created by kssd_gpu_create_for_dist: no sketch tables
    KssdParams P;
    uint8_t *d_T1;
    KssdG *d_G;
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
    void *d_big_tmp;              // rocPRIM temporary storage
    size_t cap_big_tmp;
    // inverted index (dist)
    uint32_t n_ref;
    uint64_t n_ref_ids;
    uint32_t *d_ref_sz;     // n_ref sketch sizes
    uint32_t *d_hkeys;      // the bucketed table: 16-byte slots (IdxSlot)
    uint32_t h_log2;        // log2 of the number of buckets
    uint32_t idx_lg_cap = 0;      // capped build in place: log2 of every bucket's slots (0: exact build, descriptors)
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
    bool resident_valid = false;  // the last sketch call was a host-level one: its batch is in d_in_packed / d_in_mask (kssd_gpu_sketch_again)
    uint32_t ranges_off_calls = 0;  // successful calls the switch below still lasts for
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
    hipEvent_t ev_a[2][EV_RING], ev_b[2][EV_RING];
    unsigned ev_n[2];
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
    HIPCK(hipMalloc(&c->d_G, gn * sizeof(KssdG)));
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
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < EV_RING; i++) {
            hipEventCreate(&c->ev_a[w][i]);
            hipEventCreate(&c->ev_b[w][i]);
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
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < EV_RING; i++) {
            hipEventCreate(&c->ev_a[w][i]);
            hipEventCreate(&c->ev_b[w][i]);
        }
    *out = c;
    return KSSD_OK;
}

extern "C" void kssd_gpu_destroy(kssd_gpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    void *ptrs[] = {c->d_T1, c->d_G, c->d_chunk_gid, c->d_chunk_off, c->d_reg_off, c->d_cursor, c->d_kept,
                    c->d_regions, c->d_status, c->d_ref_sz, c->d_hkeys, c->d_post, c->d_cand, c->d_cand_count, c->d_blk_info, c->d_big_alt, c->d_big_tmp,
                    c->d_in_packed, c->d_in_mask, c->d_b_ids, c->d_b_pos, c->d_b_off, c->d_bkt, c->d_sel_cnt, c->d_sel_out, c->d_arrive, c->d_tok_tab, c->d_tok_pos, c->d_tok_sum, c->d_tok_state, c->d_text, c->d_filt, c->d_tok_sup, c->d_tok_q, c->d_x_off, c->d_x_ids, c->d_med, c->d_med_out, c->d_med_cnt, c->d_hdr_cnt, c->d_hdr_pre, c->d_hdr_out, c->d_idx_flag};
    for (void *p : ptrs)
        if (p) hipFree(p);
    free(c->tok_args_saved);
    for (hipEvent_t e : c->text_ev)
        if (e) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < EV_RING; i++) {
            hipEventDestroy(c->ev_a[w][i]);
            hipEventDestroy(c->ev_b[w][i]);
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

// ---------------------------------------------------------------------------------------------------
// kernel 1: the scan.  One wave per 4096-position chunk iteration, one lane per 64 positions.
//   HBM -> registers: 16 B of packed bases + 4 B halo + 8 B of mask per lane, coalesced, two chunks ahead
//   stage 1    27 LDS byte reads per lane decide 64 positions (group-filter table, 128 KiB of LDS, kssd_core.h)
//   stage 1.5  the ~0.8 % surviving positions: pattern cut out of the lane's registers, Bloom test in LDS
//   output     the ~0.06 % that pass (global position, u64) are compacted by ballot into a per-wave LDS buffer
//              and appended to a candidate list in HBM: 8 B per candidate, ~1 % on top of the streamed bytes
// The exact evaluation (stage 2) is kernel 1b: it needs scattered reads and 64-bit arithmetic that would
// cost this kernel registers, SGPRs and issue slots in its hot loop for work done on 1 position in 1 700.
// ---------------------------------------------------------------------------------------------------

struct ScanArgs {
    const uint32_t *packed;
    const uint32_t *mask;
    unsigned long long n_chunks;
    const uint8_t *tab;             // stage-1 table followed by the Bloom filter (SCAN_TAB_BYTES)
    ulonglong2 *cand;               // (waves of the grid) * cand_cap records {global position, carried k-mer bits}
    unsigned long long cand_cap;    // per wave
    uint32_t *cand_count;           // per wave: candidates it wanted to store (> cand_cap: overflow, reported by the stage behind the scan)
    uint32_t *stage1_count;         // per wave: positions that passed stage 1 (telemetry)
    unsigned long long *blk_info;   // per block of SCAN_BLOCK chunks: where its candidates sit in the list (scan_blk_pack)
    SketchStatus *status;
    KSSD_DEV_FIELD(unsigned long long *dev_times)  // per wave {first instruction, tables in LDS, last chunk done} (shader cycles)
};

// The scan's unit of work: a block of SCAN_BLOCK consecutive chunks.  A workgroup owns one contiguous run of the batch's
// blocks, its 16 waves take them in turn (wave w: blocks w, w + 16, ... of the run).  A wave lists the survivors of a block
// contiguously in its own slice of the candidate list and leaves, per block, where (the per-genome kernel that evaluates
// them reads exactly the blocks its genome's chunks lie in):
//   bits 0-23 the number of records, bits 24-63 the index of the first one in the whole list
// The buffer of stage-1 candidates is emptied at the end of every block, so an entry names its chunk relative to the block.
// Measured and not kept (profiles/r03d_*, r03e_*): the same blocks handed out by a queue.  With equal shares the fastest wave of
// EVERY workgroup is through at 70 % of the time its slowest one takes -- the waves of a CU do not get equal shares of its
// LDS and issue slots -- while the workgroups' means agree to 2 %.  But a CU's throughput does not depend on which of its
// waves get it: with a queue head per workgroup all waves finish together and the launch takes as long as before (0.621
// against 0.598 ms); one head for the whole grid saturates (305 000 atomics on one address: 3.6 ms).
#define SCAN_BLOCK 4
__host__ __device__ __forceinline__ unsigned long long scan_blk_pack(unsigned long long first, uint32_t n) { return (first << 24) | n; }
__host__ __device__ __forceinline__ uint32_t scan_blk_count(unsigned long long v) { return (uint32_t)(v & 0xFFFFFFull); }
__host__ __device__ __forceinline__ unsigned long long scan_blk_first(unsigned long long v) { return v >> 24; }

// one chunk of the lane's share of the stream: 64 positions + halo, and their validity bits
struct ChunkRegs {
    uint32_t W[5];
    uint32_t M[2];
};

__device__ __forceinline__ void load_chunk(const ScanArgs &a, unsigned long long c, uint32_t lane, ChunkRegs &r)
{
    const uint32_t *pp = a.packed + c * 256 + lane * 4;
    const uint4 v = *reinterpret_cast<const uint4 *>(pp);
    r.W[0] = v.x; r.W[1] = v.y; r.W[2] = v.z; r.W[3] = v.w;
    r.W[4] = pp[4];
    // (the validity words are read once: streamed past the L2 lines of the packed words, which the Bloom rounds come back to;
    // -0.8 % of the kernel's time, profiles/r03w)
    const uint32_t *mp = a.mask + c * 128 + lane * 2;
    r.M[0] = __builtin_nontemporal_load(mp);
    r.M[1] = __builtin_nontemporal_load(mp + 1);
}

// Stage-1 candidates are buffered as positions only (4 bytes in LDS); their bases are fetched when the Bloom round comes, one
// lane per buffered candidate, from the packed stream itself -- a 12-byte gather per lane on the vector-memory path, which
// the scan leaves idle (3 coalesced loads per chunk), mostly L2 hits (the wave read those lines a few chunks ago; FETCH_SIZE
// says a third of them come from HBM again).  Until round 3 the scanning lane cut them out of its registers inside the
// candidate loop: 26 VALU instructions per pass at ~15 % lane efficiency (3.4 passes per chunk for ~34 candidates), more than
// stage 1 itself.  A round is issued (entries read, gathers sent) one step before it is completed (Bloom test, survivors
// stored as {global position, k-mer payload | valid << 63}, plain 16-byte stores), so nothing waits for the gather.
//   entry: [11:0] position inside its chunk (lane << 6 | b)   [22:12] chunk - block's first chunk
//          [23] 1 = all 2k bases are known to be valid (the lane's and both neighbours' 64 positions are bases)
struct ScanRound {
    uint32_t e, w0, w1, w2, n;  // n (wave-uniform): entries of the round, 0 = none pending
    unsigned long long c0;      // first chunk of the block the entries belong to (wave-uniform)
};
__device__ __forceinline__ void scan_round_issue(const ScanArgs &a, unsigned long long blk_c0, const uint32_t *cbuf, uint32_t first, uint32_t n,
                                                 uint32_t lane, ScanRound &r)
{
    r.n = n;
    r.c0 = blk_c0;
    r.e = 0;
    r.w0 = r.w1 = r.w2 = 0;
    if (lane < n) {
        r.e = cbuf[first + lane];
        const unsigned long long p = ((blk_c0 + ((r.e >> 12) & 2047u)) << 12) | (r.e & 4095u);  // global position of the sub-context
        const unsigned long long i = p >> 4;                                                     // its packed word
        const uint32_t *pp = a.packed + (i ? i - 1 : 0);  // the word in front (the 4 bases before p may lie there), the word, the next
        r.w0 = pp[0];
        r.w1 = pp[1];
        r.w2 = pp[2];
    }
}
template <int SUBK, int ABL>
__device__ __forceinline__ uint32_t scan_round_complete(const ScanArgs &a, const uint32_t *bloom, unsigned long long wid,
                                                        uint32_t stored, uint32_t lane, ScanRound &r, uint32_t &abl_acc)
{
    bool pass = false;
    uint32_t top32 = 0, front = 0;
    unsigned long long p = 0;
    if (lane < r.n) {
        p = ((r.c0 + ((r.e >> 12) & 2047u)) << 12) | (r.e & 4095u);
        const bool at0 = (p >> 4) == 0;  // no word in front of the batch's first one
        kssd_carry_from_words(at0 ? 0u : r.w0, at0 ? r.w0 : r.w1, at0 ? r.w1 : r.w2, (uint32_t)p & 15u, top32, front);
        const uint32_t h = kssd_bloom_hash(top32 >> (32 - 4 * SUBK));
        const uint32_t bits = kssd_bloom_bits(h);
        pass = (bloom[kssd_bloom_word(h)] & bits) == bits;
    }
    const uint64_t bal = __ballot(pass);
    if (pass) {
        const unsigned long long at = (unsigned long long)stored + rank_in(bal);
        if (ABL != 0) abl_acc ^= r.e;
        else if (at < a.cand_cap)
            a.cand[wid * a.cand_cap + at] = make_ulonglong2(p, kssd_carry_payload(top32, front) | ((unsigned long long)((r.e >> 23) & 1u) << 63));
    }
    r.n = 0;
    return (uint32_t)__builtin_popcountll(bal);
}

// ABL != 0: development-only ablations for profiling (1 = loads only, 2 = + stage 1, 3 = + stage 1.5 without
// the candidate list); never used by the product path.
//
// Software pipeline of one wave over its chunks (c = the chunk whose candidates are being tested):
//   HBM   chunks c+2 and c+3 are being read into registers while chunk c is worked on
//   LDS   the table reads of alignment A of chunk c+1 are in flight during the merge / Bloom / push work of
//         chunk c, those of alignment B across the loop edge: at any time one batch of <= 15 reads is
//         outstanding behind the one being waited for, which is what s_waitcnt lgkmcnt can express
template <int SUBK, int ABL = 0>
__global__ __launch_bounds__(SCAN_THREADS) void sketch_scan_kernel(ScanArgs a)
{
    typedef KssdGrp<SUBK, KSSD_GW> Gp;
    uint32_t abl_acc = 0;
    // static LDS: the table sits at LDS address 0, so its byte reads need no base add
    __shared__ __attribute__((aligned(16))) unsigned char smem[SCAN_LDS_BYTES];
    uint8_t *T1 = smem;
    const uint32_t *bloom = reinterpret_cast<const uint32_t *>(smem + KSSD_T1_BYTES);
    // readfirstlane: the compiler cannot know that threadIdx.x >> 6 is wave-uniform; without it the chunk range, the
    // loop counter and every counter derived from a ballot live in VGPRs and the loops run on exec masks
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = lane_id();
    // the call's status words start at zero: every kernel that reports into them runs behind this one on the stream, and the
    // scan itself reports per wave (cand_count / stage1_count, added up by the stage that follows) -- no reset launch in front
    if (blockIdx.x == 0 && threadIdx.x < sizeof(SketchStatus) / 4) reinterpret_cast<uint32_t *>(a.status)[threadIdx.x] = 0;
    uint32_t *pbuf = reinterpret_cast<uint32_t *>(smem + SCAN_TAB_BYTES) + wave * CBUF;  // buffered stage-1 candidates (positions, see ScanRound)
    uint32_t cn = 0, stored = 0;  // buffered stage-1 candidates / listed stage-1.5 survivors (wave-uniform)

    // work distribution: the workgroup's run of blocks, its waves take them in turn (see SCAN_BLOCK)
    const uint32_t wid = blockIdx.x * SCAN_WAVES + wave;
    const unsigned long long clast = a.n_chunks - 1;  // reads past the end of the batch are clamped, their results unused
    const uint32_t n_blocks_all = (uint32_t)((a.n_chunks + SCAN_BLOCK - 1) / SCAN_BLOCK);
    const uint32_t wg_per = (n_blocks_all + gridDim.x - 1) / gridDim.x;
    const uint32_t wg_first = blockIdx.x * wg_per;
    const uint32_t n_blocks = wg_first >= n_blocks_all ? 0u : (n_blocks_all - wg_first < wg_per ? n_blocks_all - wg_first : wg_per);  // of this workgroup
    uint32_t n_rounded = 0;  // telemetry: stage-1 candidates that went through a Bloom round (in the end: all of them)
    // rounds in flight (issued, not completed): one per step; two behind a block's last step, whose survivors -- and the block's
    // blk_info entry -- are completed by the first step of the wave's next block (or after its last one)
    ScanRound pendA, pendB;
    pendA.n = pendB.n = 0;
    pendA.e = pendA.w0 = pendA.w1 = pendA.w2 = pendB.e = pendB.w0 = pendB.w1 = pendB.w2 = 0;
    pendA.c0 = pendB.c0 = 0;
    uint32_t defer_blk = 0xFFFFFFFFu, defer_first = 0;  // the block whose blk_info entry waits for those rounds (wave-uniform)
    uint32_t cur_first = 0;  // where the survivors of the block being worked on begin in the slice: behind the last survivor of the block before
    auto finish_pending = [&]() {
        if (pendA.n) stored += scan_round_complete<SUBK, ABL>(a, bloom, wid, stored, lane, pendA, abl_acc);
        if (pendB.n) stored += scan_round_complete<SUBK, ABL>(a, bloom, wid, stored, lane, pendB, abl_acc);
        if (defer_blk != 0xFFFFFFFFu) {
            if (ABL == 0 && lane == 0) {  // where the block's survivors are (clamped to what the slice holds: an overflow is reported below)
                const unsigned long long cap = a.cand_cap;
                const unsigned long long f = defer_first < cap ? defer_first : cap, e = stored < cap ? stored : cap;
                a.blk_info[defer_blk] = scan_blk_pack((unsigned long long)wid * cap + f, (uint32_t)(e - f));
            }
            defer_blk = 0xFFFFFFFFu;
            cur_first = stored;
        }
    };
    KSSD_DEV_STAMP_CYC(dev_t0);
    auto take = [&](uint32_t prev) -> uint32_t { return prev >= n_blocks ? prev : prev + SCAN_WAVES; };  // the wave's next block of the run
    auto chunk_at = [&](unsigned long long c) -> unsigned long long { return c < clast ? c : clast; };
    uint32_t b_cur = wave, b_nxt = take(b_cur);  // indices into the workgroup's run

    // the wave's first three chunks are requested before the tables are copied into LDS: the HBM latency of the
    // first reads overlaps the 144 KiB copy instead of following it
    ChunkRegs r0, r1, r2, r3;
    uint32_t raw[Gp::NMAX];
    uint32_t alo, ahi;
    {
        const unsigned long long c0 = (unsigned long long)(wg_first + b_cur) * SCAN_BLOCK;
        load_chunk(a, chunk_at(c0), lane, r0);
        load_chunk(a, chunk_at(c0 + 1), lane, r1);
        load_chunk(a, chunk_at(c0 + 2), lane, r2);
    }
    {   // 144 KiB = nine 16-byte pieces per thread: all nine requested before the first one is stored
        static_assert(SCAN_TAB_BYTES % (SCAN_THREADS * 16) == 0, "the table copy is written for whole rounds");
        constexpr int TAB_ROUNDS = SCAN_TAB_BYTES / (SCAN_THREADS * 16);
        uint4 t[TAB_ROUNDS];
#pragma unroll
        for (int r = 0; r < TAB_ROUNDS; r++) t[r] = *reinterpret_cast<const uint4 *>(a.tab + (size_t)(r * SCAN_THREADS + threadIdx.x) * 16);
#pragma unroll
        for (int r = 0; r < TAB_ROUNDS; r++) *reinterpret_cast<uint4 *>(smem + (size_t)(r * SCAN_THREADS + threadIdx.x) * 16) = t[r];
    }
    __syncthreads();
    KSSD_DEV_STAMP_CYC(dev_t1);
    if (b_cur >= n_blocks) {  // a wave without a block (a batch smaller than the grid): nothing listed
        if (lane == 0) { a.cand_count[wid] = 0; a.stage1_count[wid] = 0; }
        return;
    }
    // whether a lane's neighbours' 64 positions are all bases (ballots of this and the previous chunk; lane 63's right neighbour
    // is not looked at, and neither is lane 0's left one in the first chunk of a block: their candidates -- ~2 % -- let the
    // exact stage read the mask)
    unsigned long long blk_c0 = (unsigned long long)(wg_first + b_cur) * SCAN_BLOCK;  // first chunk of the block being worked on
    uint64_t vb_prev = 0;

    // prologue: the block's first chunk through both alignments
    kssd_grp_issue<SUBK, KSSD_GW, 0>(r0.W, T1, raw);
    kssd_grp_merge<SUBK, KSSD_GW, 0>(raw, alo, ahi);
    kssd_grp_issue<SUBK, KSSD_GW, 1>(r0.W, T1, raw);  // alignment B of the first chunk in flight

    // one chunk.  The four register sets rotate by name (a block is four steps, written out): copying one
    // set into another would make every iteration wait for the reads it has just issued.
    //   c = the chunk of this step, crel = its index in the block, c_far = the chunk three steps ahead in the wave's
    //   sequence (the next block's chunks once this block's are requested), last = the block's last step
    auto step = [&](const ChunkRegs &cur_r, ChunkRegs &nxt_r, ChunkRegs &far, const unsigned long long c, const uint32_t crel,
                    const unsigned long long c_far, const bool last) {
        // state: cur = chunk c, nxt = the chunk after it (requested two steps ago), the one after that in flight, far = free,
        //        raw = alignment-B reads of chunk c (in flight), alo/ahi = alignment A of chunk c
        // Wave priority: the part of an iteration that feeds the memory and LDS pipes runs at high priority (3 while the
        // loads and table reads are issued, 2 for the merges and the Bloom round), the candidate loop -- a long run of
        // VALU work that nothing waits for -- at priority 0, so that a SIMD's issue slots go first to the waves that
        // keep HBM and LDS busy.  Measured on the full batch, same box: 0.555 ms without priorities, 0.544 with only the
        // table-read issue raised, 0.529 with everything but the loop at 2, 0.526 as it is here.
        __builtin_amdgcn_s_setprio(3);
        load_chunk(a, chunk_at(c_far), lane, far);
        const ChunkRegs &cur = cur_r, &nxt = nxt_r;
        const uint64_t vb = __ballot((cur.M[0] & cur.M[1]) == 0xFFFFFFFFu);
        uint32_t rawa[Gp::NMAX];
        if (ABL != 1) kssd_grp_issue<SUBK, KSSD_GW, 0>(nxt.W, T1, rawa);  // alignment A of the next chunk goes in flight
        if (ABL == 1) {
            abl_acc ^= cur.W[0] ^ cur.W[1] ^ cur.W[2] ^ cur.W[3] ^ cur.W[4] ^ cur.M[0] ^ cur.M[1];
        } else {
            uint32_t blo, bhi;
            __builtin_amdgcn_s_setprio(2);
            kssd_grp_merge<SUBK, KSSD_GW, 1>(raw, blo, bhi);  // waits for the B reads of chunk c only
            const uint32_t real = c < a.n_chunks ? 0xFFFFFFFFu : 0u;  // (the batch's last block may be short: its missing chunks hold nothing)
            uint32_t cl = alo & blo & cur.M[0] & real;  // the window start itself must be a base: kills padding / N stretches early
            uint32_t ch = ahi & bhi & cur.M[1] & real;
            __builtin_amdgcn_s_setprio(0);
            if (ABL == 2) {
                abl_acc ^= cl ^ ch;
            } else {
                const uint64_t kvm = vb & ((vb << 1) | (vb_prev >> 63)) & (vb >> 1);  // lane and both neighbours all bases
                const uint32_t ebase = (crel << 12) | (lane << 6) | ((uint32_t)((kvm >> lane) & 1ull) << 23);
                // The lanes' candidate bits become one dense list of positions: a prefix sum over the lanes' counts (DPP moves),
                // then every lane writes its own ~0.5 positions -- a loop over its set bits with nothing but a find-first-bit, a
                // clear and a 4-byte LDS write per pass, no ballot, no rank, no extraction.
                // (No vector-memory instruction may sit inside such a loop: one on ANY path through it makes the compiler wait
                // for all outstanding loads -- the prefetched chunks -- at the loop's head, every pass.)
                unsigned long long m64 = ((unsigned long long)ch << 32) | cl;
                const uint32_t mine = (uint32_t)__builtin_popcountll(m64);
                const uint32_t incl = wave_incl_scan_dpp(mine);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (cn + total > CBUF) {  // dense parameter sets only: rounds completed at once make room (the rounds in flight first: blocks stay in order)
                    finish_pending();
                    wave_lds_sync();
                    while (cn >= 64 && cn + total > CBUF) {
                        ScanRound tmp;
                        scan_round_issue(a, blk_c0, pbuf, cn - 64, 64, lane, tmp);
                        stored += scan_round_complete<SUBK, ABL>(a, bloom, wid, stored, lane, tmp, abl_acc);
                        n_rounded += 64;
                        cn -= 64;
                    }
                    wave_lds_sync();
                }
                uint32_t w = cn + incl - mine;
                if (cn + total <= CBUF) {
                    while (m64) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m64);
                        m64 &= m64 - 1ull;
                        pbuf[w++] = ebase | b;
                    }
                    cn += total;
                } else {
                    // more candidates in ONE chunk than the buffer holds (a stretch where nearly every position passes stage 1):
                    // lane after lane, 64 positions at a time
                    for (uint32_t l = 0; l < 64; l++) {
                        const unsigned long long ml = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ch, (int)l) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readlane((int)cl, (int)l);
                        if (ml == 0) continue;
                        const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)ebase, (int)l);
                        if ((ml >> lane) & 1ull) pbuf[cn + (uint32_t)__builtin_popcountll(ml & ((1ull << lane) - 1ull))] = eb | lane;
                        cn += (uint32_t)__builtin_popcountll(ml);
                        if (cn + 64 > CBUF) {
                            wave_lds_sync();
                            while (cn >= 64) {
                                ScanRound tmp;
                                scan_round_issue(a, blk_c0, pbuf, cn - 64, 64, lane, tmp);
                                stored += scan_round_complete<SUBK, ABL>(a, bloom, wid, stored, lane, tmp, abl_acc);
                                n_rounded += 64;
                                cn -= 64;
                            }
                            wave_lds_sync();
                        }
                    }
                }
            }
            vb_prev = vb;
            __builtin_amdgcn_s_setprio(2);
            // the next chunk: alignment A is in; nothing is outstanding in LDS now, which is the cheap moment for a
            // stage-1.5 round; then alignment B goes in flight across the loop edge.  A block's last step empties the
            // buffer: the block's survivors then sit in one run of the wave's slice (blk_info), and an entry never waits
            // longer than a block (it names its chunk relative to the block's first one)
            kssd_grp_merge<SUBK, KSSD_GW, 0>(rawa, alo, ahi);
            if (ABL != 2) {
                // the rounds issued one step ago are completed (their gathers have long arrived) -- with them, behind a block's
                // last step, the block's blk_info entry -- and the next round is issued.  A block's last step issues everything
                // that is left (two rounds in flight; more than 128 entries: the rest at once), so that the buffer is empty when
                // the next block begins and a block's survivors are one run of the slice
                finish_pending();
                if (!last) {
                    if (cn >= 64) {  // (a round in every step whatever it holds -- younger gathers, more L2 hits -- was slower: profiles/r03x)
                        const uint32_t n = cn < 64 ? cn : 64;
                        wave_lds_sync();
                        scan_round_issue(a, blk_c0, pbuf, cn - n, n, lane, pendA);
                        n_rounded += n;
                        cn -= n;
                    }
                } else if (cn) {
                    wave_lds_sync();
                    if (cn > 128) {  // (dense parameter sets)
                        do {
                            ScanRound tmp;
                            scan_round_issue(a, blk_c0, pbuf, cn - 64, 64, lane, tmp);
                            stored += scan_round_complete<SUBK, ABL>(a, bloom, wid, stored, lane, tmp, abl_acc);
                            n_rounded += 64;
                            cn -= 64;
                        } while (cn > 128);
                    }
                    uint32_t n = cn < 64 ? cn : 64;
                    scan_round_issue(a, blk_c0, pbuf, cn - n, n, lane, pendA);
                    n_rounded += n;
                    cn -= n;
                    if (cn) {
                        scan_round_issue(a, blk_c0, pbuf, 0, cn, lane, pendB);
                        n_rounded += cn;
                        cn = 0;
                    }
                }
            }
            kssd_grp_issue<SUBK, KSSD_GW, 1>(nxt.W, T1, raw);
        }
    };
    static_assert(SCAN_BLOCK == 4, "a block is the four written-out steps below");
    for (;;) {
        const unsigned long long c = blk_c0;
        const bool more = b_nxt < n_blocks;
        // chunks of the block after this one (none: reads clamped to the last chunk of the batch, never looked at)
        const unsigned long long nb = more ? (unsigned long long)(wg_first + b_nxt) * SCAN_BLOCK : clast;
        const uint32_t b_nn = take(b_nxt);
        step(r0, r1, r3, c, 0u, c + 3, false);
        step(r1, r2, r0, c + 1, 1u, nb, false);
        step(r2, r3, r1, c + 2, 2u, nb + 1, false);
        step(r3, r0, r2, c + 3, 3u, nb + 2, true);
        defer_blk = wg_first + b_cur;  // its last rounds are in flight: finish_pending() of the next step (or below) writes the entry
        defer_first = cur_first;
        if (!more) {
            finish_pending();
            break;
        }
        b_cur = b_nxt;
        b_nxt = b_nn;
        blk_c0 = nb;
        vb_prev = 0;
    }
    KSSD_DEV_DO(if (a.dev_times && lane == 0) {
        a.dev_times[wid * 3] = dev_t0;
        a.dev_times[wid * 3 + 1] = dev_t1;
        a.dev_times[wid * 3 + 2] = __builtin_readcyclecounter();
    })
    if (lane == 0) {
        a.cand_count[wid] = stored;  // (more than cand_cap: the slice was too small -- scan_totals reports it, the call is repeated larger)
        a.stage1_count[wid] = n_rounded;
    }
    if (ABL != 0) {
        for (int i = 0; i < Gp::NMAX; i++) abl_acc ^= raw[i];
        if (abl_acc == 0x9e3779b9u) a.cand[0] = make_ulonglong2(abl_acc, 0);  // keeps the ablated work alive
    }
}

// ---------------------------------------------------------------------------------------------------
// kernel 1b: stage 2, the exact evaluation of the candidates (kssd_stage2 in kssd_core.h: validity of all
// 2k bases, canonical strand, sub-context -> rank through the cuckoo table, reduced tuple), one lane per
// candidate at full occupancy.  Survivors are appended to their genome's staging region with one returning
// atomic per wave and genome.
// ---------------------------------------------------------------------------------------------------
// Staged tuples are sorted per genome.  Plain mode: the key is the reduced tuple (u32).  First-position mode
// (KSSD_SKETCH_FIRST_POS): key = tuple << 32 | position inside the genome, so that the first entry of a run of equal
// tuples carries the tuple's first occurrence -- what the host needs to replay the reference's hash insertions in
// sequence order and leave combco.* byte-identical even where two ids of a genome probe the same slot.
// what the waves of the scan left per slice, added up into the call's status by ONE workgroup of the stage behind the scan
// (plain stores: no other kernel writes these four words)
__device__ __forceinline__ void scan_totals(const uint32_t *__restrict__ cand_count, uint32_t n_slices, unsigned long long cand_cap,
                                            SketchStatus *st, unsigned long long *s_acc /* LDS: 3 words */)
{
    const uint32_t *stage1_count = cand_count + n_slices;
    if (threadIdx.x == 0) s_acc[0] = s_acc[1] = s_acc[2] = 0;
    __syncthreads();
    unsigned long long bloom = 0, stage1 = 0, need = 0;
    for (uint32_t w = threadIdx.x; w < n_slices; w += blockDim.x) {
        const uint32_t n = cand_count[w];
        bloom += n;
        stage1 += stage1_count[w];
        need = n > need ? n : need;
    }
    atomicAdd(&s_acc[0], bloom);
    atomicAdd(&s_acc[1], stage1);
    atomicMax(&s_acc[2], need);
    __syncthreads();
    if (threadIdx.x == 0) {
        st->n_bloom = s_acc[0];
        st->n_stage1 = s_acc[1];
        if (s_acc[2] > cand_cap) {
            st->cand_overflow = 1u;
            st->cand_need = (uint32_t)s_acc[2];
        }
    }
}

template <typename K> struct KeyOps;
template <> struct KeyOps<uint32_t> {
    static __device__ __forceinline__ uint32_t id(uint32_t k) { return k; }
    static __device__ __forceinline__ uint32_t pos(uint32_t) { return 0u; }
    static __device__ __forceinline__ uint32_t make(uint32_t dr, uint32_t) { return dr; }
    static constexpr uint32_t pad() { return 0xFFFFFFFFu; }
};
template <> struct KeyOps<unsigned long long> {
    static __device__ __forceinline__ uint32_t id(unsigned long long k) { return (uint32_t)(k >> 32); }
    static __device__ __forceinline__ uint32_t pos(unsigned long long k) { return (uint32_t)k; }
    static __device__ __forceinline__ unsigned long long make(uint32_t dr, uint32_t p) { return ((unsigned long long)dr << 32) | p; }
    static constexpr unsigned long long pad() { return 0xFFFFFFFFFFFFFFFFull; }
};

struct ExactArgs {
    const uint32_t *packed;
    const uint32_t *mask;
    const uint32_t *chunk_gid;
    const unsigned long long *chunk_off;  // per genome, in chunks
    const KssdG *G;
    const ulonglong2 *cand;  // {global position of the sub-context, carried k-mer bits | known-valid << 63}
    unsigned long long cand_cap;
    const uint32_t *cand_count;
    uint32_t n_slices;
    uint32_t carry;  // kssd_carry_ok: the payload holds the whole k-mer, the packed stream is not read again
    const unsigned long long *reg_off;
    uint32_t *cursor;
    void *regions;  // uint32_t[] or, in first-position mode, unsigned long long[]
    uint32_t by_pos;  // KSSD_SKETCH_BY_POS: the 64-bit key is position << 32 | tuple, so that the sort leaves sequence order
    SketchStatus *status;
    KSSD_DEV_FIELD(uint32_t dev_no_atomic)  // A/B (wrong results): the workgroup's room is not reserved at the genome's cursor
};

#define EXACT_THREADS 256  // four waves share one reservation of staging room (see below); 1 024 threads: fewer atomics, and 450 us instead of 210 -- too few chains in flight
#define EXACT_PER 4  // candidates per thread: the kernel is a chain of dependent memory round trips (record -> genome of the chunk ->
                     // cuckoo slots -> cursor -> staging region), so every thread keeps four chains in flight
template <typename K>
__global__ __launch_bounds__(EXACT_THREADS) void sketch_exact_kernel(KssdParams P, ExactArgs x)
{
    // grid = (blocks per slice, slices): block (bx, w) takes candidates [1024 bx, 1024 bx + 1024) of scan wave w,
    // thread t the candidates 1024 bx + 256 j + t
    const uint32_t lane = lane_id();
    const uint32_t w = blockIdx.y;
    const uint32_t want = x.cand_count[w];
    if (blockIdx.x == 0 && blockIdx.y == 0) {  // (workgroup-uniform)
        __shared__ unsigned long long s_acc[3];
        scan_totals(x.cand_count, x.n_slices, x.cand_cap, x.status, s_acc);
    }
    const uint32_t n = want < x.cand_cap ? want : (uint32_t)x.cand_cap;
    const uint32_t i0 = blockIdx.x * ((uint32_t)EXACT_THREADS * EXACT_PER) + threadIdx.x;
    if (blockIdx.x * ((uint32_t)EXACT_THREADS * EXACT_PER) >= n) return;  // whole block past the end of the slice
    // Same arithmetic as kssd_stage2 (kssd_core.h).  The candidate record carries the k-mer's bases (the scan's Bloom rounds cut
    // them out of the packed words: kssd_carry_from_words) and whether all of them are known to be bases, so this stage is a streaming
    // read of 16-byte records plus two probes of the L2-resident cuckoo table; the packed stream is read again only for
    // parameter sets whose k-mer is longer than the 20 carried bases, the mask only for the ~2 % of the candidates near an
    // invalid position.  Out-of-genome k-mers are rejected at the end; their reads stay inside the batch (position clamped
    // at 0, slack words after the last chunk).
    bool ok[EXACT_PER];
    uint32_t dr[EXACT_PER], gid[EXACT_PER], gpos[EXACT_PER], dim[EXACT_PER];
    ulonglong2 cd[EXACT_PER];
    uint64_t u[EXACT_PER];
    bool valid[EXACT_PER];
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {
        const uint32_t i = i0 + (uint32_t)EXACT_THREADS * j;
        ok[j] = i < n;
        cd[j] = ok[j] ? x.cand[(unsigned long long)w * x.cand_cap + i] : make_ulonglong2(0ull, 0ull);
    }
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) gid[j] = ok[j] ? x.chunk_gid[cd[j].x >> 12] : 0u;
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {
        const long long b0 = (long long)cd[j].x - P.out;
        const unsigned long long b0c = b0 < 0 ? 0ull : (unsigned long long)b0;
        const bool known = (cd[j].y >> 63) != 0;
        valid[j] = true;
        u[j] = 0;
        dim[j] = 0;
        if (ok[j]) {
            if (x.carry) {
                kssd_s2_canon(P, kssd_carry_fwd(P, cd[j].y & 0xFFFFFFFFFFull), u[j], dim[j]);
                if (!known) {
                    const uint32_t *mp = x.mask + (b0c >> 5);
                    const uint64_t m64 = (uint64_t)mp[0] | ((uint64_t)mp[1] << 32), need = (1ull << P.nb) - 1ull;
                    valid[j] = ((m64 >> (b0c & 31ull)) & need) == need;
                }
            } else {
                const uint32_t *pp = x.packed + (b0c >> 4), *mp = x.mask + (b0c >> 5);
                const uint32_t p0 = __builtin_nontemporal_load(pp), p1 = __builtin_nontemporal_load(pp + 1), p2 = __builtin_nontemporal_load(pp + 2);
                uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
                if (!known) { m0 = mp[0]; m1 = mp[1]; }
                valid[j] = kssd_s2_decode(P, p0, p1, p2, m0, m1, (uint32_t)b0c, u[j], dim[j]);
            }
        }
    }
    const KssdGBucket *GB = reinterpret_cast<const KssdGBucket *>(x.G);
    KssdGBucket gb[EXACT_PER];  // one 16-byte read per candidate, all in flight together (kssd_core.h: the exact table)
    unsigned long long glo[EXACT_PER], ghi[EXACT_PER];
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {
        gb[j] = GB[kssd_g_slot(dim[j], P.g_mul[0], P.g_log2)];
        glo[j] = x.chunk_off[gid[j]] * KSSD_CHUNK;
        ghi[j] = x.chunk_off[gid[j] + 1] * KSSD_CHUNK;
    }
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {
        const long long s = (long long)cd[j].x, b0 = s - P.out;
        uint32_t rank = 0;
        int hit = kssd_g_match(gb[j], dim[j], rank);
        if (hit == 2) hit = kssd_g_match(GB[kssd_g_slot(dim[j], P.g_mul[1], P.g_log2)], dim[j], rank) == 1 ? 1 : 0;  // (1.4 % of the buckets)
        bool mine;
        dr[j] = kssd_s2_tuple(P, u[j], rank, mine);
        ok[j] = ok[j] && valid[j] && hit == 1 && mine && b0 >= (long long)glo[j] && b0 + P.nb <= (long long)ghi[j];
        gpos[j] = (uint32_t)(s - (long long)glo[j]);  // first-position mode: genomes are < 2^32 positions there (checked on the host)
    }
    // Survivors go to their genome's staging region.  Candidates arrive in stream order, so a wave's survivors almost
    // always belong to ONE genome, and so do the workgroup's: then ONE returning atomic reserves room for all of them.
    // (One per wave was enough for genome-sized sketches; a read set is one genome, and 34 000 waves queueing at one
    // address took 0.45 ms for 8.6 M candidates.)
    uint64_t bal[EXACT_PER];
    uint32_t total = 0, g0 = 0;
    bool have = false, same = true;
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {
        bal[j] = __ballot(ok[j]);
        if (bal[j]) {
            const uint32_t g = __builtin_amdgcn_readlane(gid[j], __builtin_ctzll(bal[j]));
            if (!have) { g0 = g; have = true; }
            same = same && (__ballot(ok[j] && gid[j] != g0) == 0);
            total += (uint32_t)__builtin_popcountll(bal[j]);
        }
    }
    __shared__ uint32_t s_kind[EXACT_THREADS / 64], s_g[EXACT_THREADS / 64], s_tot[EXACT_THREADS / 64], s_base;
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) { s_kind[wave] = have ? (same ? 1u : 2u) : 0u; s_g[wave] = g0; s_tot[wave] = total; }
    __syncthreads();
    bool wg_same = true;
    uint32_t wg_g = 0xFFFFFFFFu, wg_before = 0, wg_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < EXACT_THREADS / 64; k++) {
        const uint32_t kind = s_kind[k];
        if (kind == 2u) wg_same = false;
        if (kind == 1u) {
            if (wg_g == 0xFFFFFFFFu) wg_g = s_g[k];
            else if (s_g[k] != wg_g) wg_same = false;
            if (k < wave) wg_before += s_tot[k];
            wg_total += s_tot[k];
        }
    }
    if (wg_same && wg_total) {  // (workgroup-uniform)
        KSSD_DEV_DO(if (x.dev_no_atomic) { if (threadIdx.x == 0) s_base = (blockIdx.y * gridDim.x + blockIdx.x) * 400u; } else)
        if (threadIdx.x == 0) s_base = atomicAdd(&x.cursor[wg_g], wg_total);
        __syncthreads();
    }
    if (!have) return;
    if (same) {
        uint32_t at = 0;
        if (wg_same) {
            at = s_base + wg_before;
        } else {
            if (lane == 0) at = atomicAdd(&x.cursor[g0], total);
            at = __builtin_amdgcn_readfirstlane(at);
        }
        const unsigned long long r0 = x.reg_off[g0], cap = x.reg_off[g0 + 1] - r0;
#pragma unroll
        for (int j = 0; j < EXACT_PER; j++) {
            if (ok[j]) {
                const unsigned long long pos = (unsigned long long)at + rank_in(bal[j]);
                if (pos < cap) reinterpret_cast<K *>(x.regions)[r0 + pos] = x.by_pos ? KeyOps<K>::make(gpos[j], dr[j]) : KeyOps<K>::make(dr[j], gpos[j]);
            }
            at += (uint32_t)__builtin_popcountll(bal[j]);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < EXACT_PER; j++) {  // a genome boundary inside the wave's candidates: group by genome
        uint64_t todo = bal[j];
        while (todo) {
            const uint32_t leader = __builtin_ctzll(todo);
            const uint32_t g = __builtin_amdgcn_readlane(gid[j], leader);
            const bool mine = ok[j] && gid[j] == g;
            const uint64_t grp = __ballot(mine);
            uint32_t at = 0;
            if (lane == leader) at = atomicAdd(&x.cursor[g], (uint32_t)__builtin_popcountll(grp));
            at = __builtin_amdgcn_readlane(at, leader);
            if (mine) {
                const unsigned long long r0 = x.reg_off[g], cap = x.reg_off[g + 1] - r0;
                const unsigned long long pos = (unsigned long long)at + rank_in(grp);
                if (pos < cap) reinterpret_cast<K *>(x.regions)[r0 + pos] = x.by_pos ? KeyOps<K>::make(gpos[j], dr[j]) : KeyOps<K>::make(dr[j], gpos[j]);
            }
            todo &= ~grp;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// kernel 2: per-genome dedup.  One workgroup per genome: LDS bitonic sort of the staged tuples, run
// detection, the reference's keep rules, ballot/scan compaction back into the head of the region.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum /*>=DEDUP_THREADS/64+1*/, uint32_t &total)
{
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < DEDUP_THREADS / 64; w++) {
        uint32_t s = wsum[w];
        if (w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + incl - v;
}

// Bitonic sort of np = EPT * DEDUP_THREADS keys with the keys in registers: thread t holds elements t*EPT .. t*EPT+EPT-1.
// A pass whose partner distance j is below EPT stays inside a thread, one below 64*EPT inside a wave (one shuffle per
// key), and only the few passes beyond that go through LDS -- 6 of the 66 passes of a 2 048-key sort, where the plain
// LDS version moves every key through LDS (and a block barrier) in all 66.  Leaves the sorted keys in a[0 .. np).
__device__ __forceinline__ uint32_t shfl_xor_key(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ unsigned long long shfl_xor_key(unsigned long long v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

template <typename K, int EPT>
__device__ __forceinline__ void sort_in_registers(K *a, const K *src /* may be a itself (fused path): no __restrict__ */, uint32_t n, uint32_t tid)
{
    constexpr uint32_t np = EPT * DEDUP_THREADS;
    K x[EPT];
#pragma unroll
    for (int r = 0; r < EPT; r++) {
        const uint32_t e = tid * EPT + r;
        x[r] = e < n ? src[e] : KeyOps<K>::pad();
    }
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j < (uint32_t)EPT) {  // partner in this thread
#pragma unroll
                for (int r = 0; r < EPT; r++) {
                    if ((r & j) == 0) {
                        const uint32_t e = tid * EPT + r;
                        const bool asc = (e & k) == 0;
                        const K lo = x[r], hi = x[r | j];
                        if ((lo > hi) == asc) { x[r] = hi; x[r | j] = lo; }
                    }
                }
            } else {
                // j >= EPT (and k > j): whether this thread keeps the smaller or the larger key of a pair is the same for
                // all its keys
                const uint32_t e0 = tid * EPT;
                const bool take_min = ((e0 & j) == 0) == ((e0 & k) == 0);
                if (j < 64u * EPT) {  // partner in this wave
                    const int m = (int)(j / EPT);
#pragma unroll
                    for (int r = 0; r < EPT; r++) {
                        const K y = shfl_xor_key(x[r], m);
                        const K lo = x[r] < y ? x[r] : y, hi = x[r] < y ? y : x[r];
                        x[r] = take_min ? lo : hi;
                    }
                } else {  // partner in another wave: through LDS
#pragma unroll
                    for (int r = 0; r < EPT; r++) a[e0 + r] = x[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EPT; r++) {
                        const K y = a[(e0 ^ j) + r];
                        const K lo = x[r] < y ? x[r] : y, hi = x[r] < y ? y : x[r];
                        x[r] = take_min ? lo : hi;
                    }
                    __syncthreads();
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < EPT; r++) a[tid * EPT + r] = x[r];
    __syncthreads();
}

// Sort of one genome's keys by bucket counting in LDS, for the sizes the per-genome kernel mostly sees (up to
// DEDUP_BSORT_MAX keys).  The bitonic network in registers moves every key through the LDS crossbar once per pass -- ~45
// `ds_bpermute` per key at 2 048 keys, and a `ds_bpermute` costs what a conflicted LDS read costs (profiles/r03b_valu_probe.txt):
// with four workgroups per CU the sort took 45 000 of the kernel's 118 000 cycles.  The ids are reduced tuples -- the
// outermost bases of the canonical k-mer in their top bits -- so they spread over the id range: as many buckets as key
// slots (one LDS atomic per key to count, one to place), a workgroup scan of the counters, and an insertion sort of
// every bucket by the thread that owns it (most hold none or one; the canonical strand makes low prefixes more likely,
// a bucket of a dozen is rare).  Equal ids (repeats) meet in one bucket.  A batch whose ids do not spread -- low-complexity
// sequence, crafted input -- shows up as a bucket above DEDUP_BSORT_BUCKET keys: the function says no and the bitonic sort
// runs as before.  Returns true with the sorted keys in b[0 .. n).
#define DEDUP_BSORT_MAX 4096u     // key slots (= buckets) at most
#define DEDUP_BSORT_BUCKET 24u    // a fuller bucket sends the genome to the bitonic sort
// The ids lie in [id_base, id_base + 2^id_bits) (roughly: beyond it they share the last bucket).  max_bucket: a fuller bucket
// sends the keys to the bitonic sort up front (DEDUP_BSORT_BUCKET for a genome's keys; none for an item of a large genome,
// whose ids repeat by the read set's coverage -- equal keys cost an insertion sort nothing); a thread that has MOVED
// DEDUP_BSORT_MOVES keys in its buckets gives up for the workgroup either way (src is untouched).
#define DEDUP_BSORT_MOVES 256u
#define DEDUP_BSORT_HEAVY 256u    // full buckets (more than 16 keys) a workgroup lists at most
template <typename K>
__device__ __forceinline__ bool lds_bucket_sort(const K *src, K *b, uint32_t *cnt, uint32_t n, uint32_t nb /*pow2*/, uint32_t id_base, uint32_t id_bits,
                                                uint32_t max_bucket, uint32_t tid, uint32_t *wsum)
{
    uint32_t lg = 0;
    while ((1u << lg) < nb) lg++;
    const uint32_t shift = id_bits > lg ? id_bits - lg : 0u, last = nb - 1u;
    for (uint32_t i = tid; i < nb; i += DEDUP_THREADS) cnt[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += DEDUP_THREADS) {
        const uint32_t bk = (KeyOps<K>::id(src[i]) - id_base) >> shift;
        atomicAdd(&cnt[bk < last ? bk : last], 1u);
    }
    __syncthreads();
    // exclusive scan of the counters: every thread owns nb / DEDUP_THREADS consecutive ones (or none)
    const uint32_t per = nb >= DEDUP_THREADS ? nb / DEDUP_THREADS : 1u, b0 = tid * per;
    uint32_t sum = 0, mx = 0;
    if (b0 < nb)
        for (uint32_t k = 0; k < per; k++) { const uint32_t c = cnt[b0 + k]; sum += c; mx = c > mx ? c : mx; }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, wsum, total);
    // the fullest bucket decides (workgroup-uniform): packed into the scan's scratch words by a second tiny reduction
    uint32_t mx_all;
    {
        const uint32_t lane = lane_id(), wave = tid >> 6;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, 64); mx = o > mx ? o : mx; }
        if (lane == 0) wsum[wave] = mx;
        __syncthreads();
        mx_all = 0;
#pragma unroll
        for (uint32_t w = 0; w < DEDUP_THREADS / 64; w++) mx_all = wsum[w] > mx_all ? wsum[w] : mx_all;
        __syncthreads();
    }
    if (mx_all > max_bucket) return false;
    if (b0 < nb)
        for (uint32_t k = 0; k < per; k++) { const uint32_t c = cnt[b0 + k]; cnt[b0 + k] = run; run += c; }  // the bucket's first slot
    __syncthreads();
    for (uint32_t i = tid; i < n; i += DEDUP_THREADS) {
        const K kv = src[i];
        const uint32_t bk = (KeyOps<K>::id(kv) - id_base) >> shift;
        b[atomicAdd(&cnt[bk < last ? bk : last], 1u)] = kv;
    }
    __syncthreads();
    // small buckets: an insertion sort by the thread that owns the bucket.  A full bucket is nearly always a few ids many
    // times over (a read set's coverage): those go on a list and a whole wave takes each (below)
    uint32_t moves = 0;
    if (tid == 0) wsum[0] = 0;  // the list's length; the list itself: the scan's scratch words do not hold it -> the counters' tail
    __syncthreads();
    uint32_t *heavy = cnt + nb;  // [DEDUP_BSORT_HEAVY] bucket numbers (the caller's counter array has this room)
    for (uint32_t bk = tid; bk < nb; bk += DEDUP_THREADS) {  // (after the scatter cnt[bk] is the bucket's end = the next bucket's first slot)
        const uint32_t s0 = bk ? cnt[bk - 1] : 0u, e0 = cnt[bk];
        if (e0 - s0 > 16u) {
            const uint32_t at = atomicAdd(&wsum[0], 1u);
            if (at < DEDUP_BSORT_HEAVY) heavy[at] = bk;
            continue;
        }
        for (uint32_t i = s0 + 1; i < e0 && moves < DEDUP_BSORT_MOVES; i++) {
            const K x = b[i];
            uint32_t j = i;
            while (j > s0 && b[j - 1] > x) { b[j] = b[j - 1]; j--; moves++; }
            b[j] = x;
        }
    }
    __syncthreads();
    const uint32_t n_heavy = wsum[0];
    bool fail = moves >= DEDUP_BSORT_MOVES || n_heavy > DEDUP_BSORT_HEAVY;
    if (!fail && n_heavy) {
        // a wave per full bucket: its distinct ids (up to eight), how often each is there and the smallest key of each, then the
        // bucket rewritten as runs in id order -- every run as copies of its SMALLEST key, which is all the keep rules look at
        // (the run's first entry: the tuple's first position; the run's length)
        const uint32_t lane = lane_id(), wave = tid >> 6;
        for (uint32_t hi = wave; hi < n_heavy; hi += DEDUP_THREADS / 64) {
            const uint32_t bk = heavy[hi];
            const uint32_t s0 = bk ? cnt[bk - 1] : 0u, e0 = cnt[bk];
            K mk[8];
            uint32_t c[8], nd = 0;
            bool many = false;
#pragma unroll
            for (uint32_t d = 0; d < 8; d++) { mk[d] = KeyOps<K>::pad(); c[d] = 0; }
            for (uint32_t base = s0; base < e0 && !many; base += 64) {
                const bool valid = base + lane < e0;
                const K x = valid ? b[base + lane] : KeyOps<K>::pad();
                uint64_t pending = __ballot(valid);
                while (pending && !many) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(pending);
                    const uint32_t idl = (uint32_t)__builtin_amdgcn_readlane((int)KeyOps<K>::id(x), (int)l);
                    const bool same = valid && KeyOps<K>::id(x) == idl;
                    const uint64_t sb = __ballot(same);
                    K mn = same ? x : KeyOps<K>::pad();
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) { const K o = shfl_xor_key(mn, m); mn = o < mn ? o : mn; }
                    const uint32_t add = (uint32_t)__builtin_popcountll(sb);
                    bool found = false;
#pragma unroll
                    for (uint32_t d = 0; d < 8; d++)
                        if (!found && d < nd && KeyOps<K>::id(mk[d]) == idl) { c[d] += add; mk[d] = mn < mk[d] ? mn : mk[d]; found = true; }
                    if (!found) {
                        if (nd == 8) many = true;
#pragma unroll
                        for (uint32_t d = 0; d < 8; d++)
                            if (d == nd) { mk[d] = mn; c[d] = add; }
                        nd++;
                    }
                    pending &= ~sb;
                }
            }
            if (many) { fail = true; continue; }
#pragma unroll
            for (uint32_t r = 0; r < 7; r++)  // the (up to eight) runs by key; unused entries hold the pad key and stay behind
#pragma unroll
                for (uint32_t d = 0; d + 1 < 8; d++)
                    if (mk[d + 1] < mk[d]) { const K tk = mk[d]; mk[d] = mk[d + 1]; mk[d + 1] = tk; const uint32_t tc = c[d]; c[d] = c[d + 1]; c[d + 1] = tc; }
            uint32_t p = s0;
#pragma unroll
            for (uint32_t d = 0; d < 8; d++) {
                for (uint32_t t = lane; t < c[d]; t += 64) b[p + t] = mk[d];
                p += c[d];
            }
        }
    }
    return __syncthreads_or(fail) == 0;
}

// FUSED: the workgroup takes its genome's candidates straight from the scan's candidate list (the slices of the scan
// waves whose chunk runs overlap the genome), evaluates them (stage 2, as sketch_exact_kernel does) and collects the
// survivors in LDS, where the sort needs them anyway: no staging region is written and read back, no cursor atomics, no
// chunk -> genome map, one kernel less.  Used whenever no genome of the batch needs the global-memory path.
struct FuseArgs {
    const ulonglong2 *cand;
    const unsigned long long *blk_info;  // per block of SCAN_BLOCK chunks: first record and record count (scan_blk_pack)
    const unsigned long long *chunk_off;
    const uint32_t *packed, *mask;
    const KssdG *G;
    uint32_t carry, by_pos, lds_keys;  // lds_keys: keys the dynamic LDS array holds
    const uint32_t *cand_count;        // FUSED: the scan's per-slice counts (workgroup 0 adds them up into the status: scan_totals)
    uint32_t n_slices;
    unsigned long long cand_cap;
    uint32_t id_bits, bsort_keys;      // ids are below 2^id_bits (roughly); bsort_keys: key slots of the bucket sort's LDS arrays (0: none)
    KSSD_DEV_FIELD(unsigned long long *dev_times)  // per workgroup {start, keys in LDS, sorted, done}
    KSSD_DEV_FIELD(uint32_t dev_split)             // KSSD_DEV_GATHERSPLIT: the four stamps are start, block table in, first round done, keys in LDS
};
#define FUSE_PER 4  // candidates a thread evaluates at a time (six spill at 64 VGPRs)

// PARTS: a genome whose staged tuples do not fit one workgroup's LDS sort, but whose ids split by their top bits into 2, 4,
// 8 or 16 ranges that do (a 3 Gb record at -s 7 -l 5 stages ~45 000 tuples; chromosomes).  Workgroup (m, p) of the launch
// takes the tuples of medium genome m whose id lies in range p out of the genome's staging region into LDS, sorts them and
// applies the keep rules like any small genome; the ranges are disjoint and ordered, so their results, one behind the other,
// are the genome's sorted sketch (dedup_parts_finish_kernel puts them back into the region and adds up the counts).
// Two launches for all such genomes of a batch, where the global-memory path takes seven per genome.
#define DEDUP_MAX_PARTS_LOG2 4
struct PartArgs {
    const uint2 *list;   // per medium genome: genome index, log2 of its number of parts
    void *out;           // [n_medium][1 << DEDUP_MAX_PARTS_LOG2][part_cap] kept keys per part
    uint32_t *cnt;       // [n_medium][1 << DEDUP_MAX_PARTS_LOG2][4]: kept, distinct, occurrences of id 0, overflowed
    uint32_t part_cap;   // keys one part may hold (the LDS array)
    uint32_t id_bits;    // ids are below 2^id_bits (a little above it where the rank is ADDED over the outer bases: clamped)
    // RANGES (one large genome per launch, see big_rng_* below): the genome's keys partitioned by ranges of their leading
    // field; workgroup k sorts the whole bins that begin inside [k RNG_T, (k+1) RNG_T) of the partitioned array, in place
    const void *parted;        // partitioned keys (the sort's kept keys go back to the head of the item's span)
    const uint32_t *bin_start; // [rng_bins + 1]
    uint32_t *item_cnt;        // [items]: kept keys of the item
    uint32_t *item_s0;         // [items]: where the item's span begins
    const uint32_t *item_bin;  // [items + 1]: the item's first bin
    uint32_t *acc;             // [2]: the genome's distinct tuples, occurrences of id 0
    uint32_t rng_bins, rng_g;
};
enum { DEDUP_STAGED = 0, DEDUP_FUSED = 1, DEDUP_PARTS = 2, DEDUP_RANGES = 3 };
#define RNG_T 1024u             // span of the partitioned array an item begins in
#define RNG_BIN_MEAN 512u      // keys per bin aimed at
#define RNG_MAX_LOG2_BINS 14u
#define RNG_BSORT_KEYS 2048u    // bucket-sort arrays of an item (items of up to 2 048 keys, nearly all of them): 33 KB of LDS per workgroup,
                                // four per CU, instead of 49 KB and three (configs[3] step -0.7 %); larger items take the bitonic sort
#define RNG_PART_CAP 4096u      // keys an item may hold (its LDS array): RNG_T + the largest bin, i.e. a bin six times the mean

template <typename K, int MODE>
__global__ __launch_bounds__(DEDUP_THREADS, 8) void sketch_dedup_kernel(KssdParams P, const unsigned long long *__restrict__ reg_off,
                                                                      const uint32_t *__restrict__ cursor,
                                                                      K *__restrict__ regions, uint32_t *__restrict__ kept,
                                                                      uint32_t flags, uint32_t min_occ, uint32_t big_min,
                                                                      SketchStatus *st, FuseArgs fx, PartArgs px)
{
    constexpr bool FUSED = MODE == DEDUP_FUSED;
    KSSD_DEV_STAMP(dev_t0);
    KSSD_DEV_VAR(dev_t1);
    KSSD_DEV_VAR(dev_t2);
    KSSD_DEV_VAR(dev_tA);
    KSSD_DEV_VAR(dev_tB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K *a = reinterpret_cast<K *>(smem);
    uint32_t g = blockIdx.x, part = 0, lg_parts = 0;
    if (MODE == DEDUP_RANGES) {
        g = px.rng_g;
    } else if (MODE == DEDUP_PARTS) {
        const uint2 md = px.list[blockIdx.x];
        g = md.x;
        lg_parts = md.y;
        part = blockIdx.y;
        if (part >> lg_parts) return;  // this genome has fewer parts than the launch is wide
    } else if (reg_off[blockIdx.x + 1] - reg_off[blockIdx.x] > big_min) {
        return;  // too large for one workgroup's LDS: the parts launch or the global-memory path below
    }
    __shared__ uint32_t wsum[DEDUP_THREADS / 64 + 1];
    __shared__ uint32_t s_distinct, s_zero_occ;
    const uint32_t tid = threadIdx.x;
    const unsigned long long r0 = reg_off[g];
    const uint32_t cap = (uint32_t)(reg_off[g + 1] - r0);
    uint32_t *pcnt = MODE == DEDUP_PARTS ? px.cnt + ((size_t)blockIdx.x << (DEDUP_MAX_PARTS_LOG2 + 2)) + part * 4u : nullptr;
    uint32_t n;
    unsigned long long rng_s0 = 0;
    uint32_t rng_id_base = 0, rng_id_bits = 0;  // the item's ids lie in [base, base + 2^bits)
    if (MODE == DEDUP_RANGES) {
        __shared__ uint32_t s_span[4];
        const uint32_t staged = cursor[g];
        if (staged > cap) return;  // (big_rng_hist_kernel has reported the overflow)
        const unsigned long long lo = (unsigned long long)blockIdx.x * RNG_T;
        if (tid == 0) {  // the bins whose first key lies in [lo, hi) (big_rng_binscan_kernel has looked them up)
            const uint32_t b0 = px.item_bin[blockIdx.x], b1 = px.item_bin[blockIdx.x + 1];
            s_span[0] = px.bin_start[b0];
            s_span[1] = px.bin_start[b1];  // (bin_start[rng_bins] = all keys)
            s_span[2] = b0;
            s_span[3] = b1 > b0 ? b1 - b0 : 1u;
        }
        __syncthreads();
        rng_s0 = s_span[0];
        {
            uint32_t lgb = 0;
            while ((1u << lgb) < px.rng_bins) lgb++;
            const uint32_t bin_shift = px.id_bits - lgb;  // (big_rng_hist_kernel's shift)
            uint32_t wb = 0;
            while ((1u << wb) < s_span[3]) wb++;
            rng_id_base = s_span[2] << bin_shift;
            rng_id_bits = wb + bin_shift;
        }
        n = lo < staged ? s_span[1] - s_span[0] : 0u;
        if (n > px.part_cap) {  // the keys do not spread (or one id is there tens of thousands of times): the caller repeats with the global sort
            if (tid == 0) atomicOr(&st->ranges_skew, 1u);
            n = 0;
        }
        if (n == 0) {
            if (tid == 0) { px.item_cnt[blockIdx.x] = 0; px.item_s0[blockIdx.x] = 0; }
            return;
        }
        const K *src_g = reinterpret_cast<const K *>(px.parted) + rng_s0;
        for (uint32_t i = tid; i < n; i += DEDUP_THREADS) a[i] = src_g[i];
        __syncthreads();
    } else if (MODE == DEDUP_PARTS) {
        __shared__ uint32_t s_np;
        const uint32_t lane = lane_id();
        if (tid == 0) s_np = 0;
        __syncthreads();
        const uint32_t staged = cursor[g];
        if (staged > cap) {  // the exact stage wanted to stage more than the region holds: the call is repeated larger
            if (tid == 0) {
                if (part == 0) {
                    atomicOr(&st->region_overflow, 1u);
                    unsigned long long need = ((unsigned long long)staged * 256ull + cap - 1) / (cap ? cap : 1);
                    atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
                }
                pcnt[0] = pcnt[1] = pcnt[2] = 0;
                pcnt[3] = 1;
            }
            return;
        }
        const uint32_t shift = px.id_bits - lg_parts, last_part = (1u << lg_parts) - 1u;
        // eight tuples per thread and round, their reads in flight together (one per round: 88 dependent round trips for the
        // 45 000 tuples of a 3 Gb record, 143 us for the launch)
        for (uint32_t i0 = 0; i0 < staged; i0 += DEDUP_THREADS * 8) {
            K kv[8];
            bool mine[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t i = i0 + (uint32_t)j * DEDUP_THREADS + tid;
                kv[j] = i < staged ? regions[r0 + i] : KeyOps<K>::pad();
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t i = i0 + (uint32_t)j * DEDUP_THREADS + tid;
                const uint32_t pp = KeyOps<K>::id(kv[j]) >> shift;
                mine[j] = i < staged && (pp < last_part ? pp : last_part) == part;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint64_t bal = __ballot(mine[j]);
                if (bal) {  // one LDS atomic per wave reserves room for its tuples
                    uint32_t at = 0;
                    if (lane == 0) at = atomicAdd(&s_np, (uint32_t)__builtin_popcountll(bal));
                    at = __builtin_amdgcn_readfirstlane(at) + rank_in(bal);
                    if (mine[j] && at < px.part_cap) a[at] = kv[j];
                }
            }
        }
        __syncthreads();
        n = s_np;
        if (n > px.part_cap) {  // this id range holds more than the LDS sort takes: larger regions mean more, narrower parts
            if (tid == 0) {
                atomicOr(&st->region_overflow, 1u);
                const unsigned long long need = ((unsigned long long)n * 256ull + px.part_cap - 1) / px.part_cap;
                atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
                pcnt[0] = pcnt[1] = pcnt[2] = 0;
                pcnt[3] = 1;
            }
            return;
        }
    } else if (FUSED) {
        __shared__ uint32_t s_n, s_pref[DEDUP_THREADS];
        __shared__ unsigned long long s_first[DEDUP_THREADS];
        const uint32_t lane = lane_id();
        if (blockIdx.x == 0 && fx.n_slices) scan_totals(fx.cand_count, fx.n_slices, fx.cand_cap, st, s_first);  // (s_first: not in use yet)
        if (tid == 0) s_n = 0;
        const unsigned long long cb = fx.chunk_off[g], ce = fx.chunk_off[g + 1];
        const long long glo = (long long)(cb * KSSD_CHUNK), ghi = (long long)(ce * KSSD_CHUNK);
        // the blocks of the scan that overlap the genome's chunks (an empty genome: none)
        const unsigned long long w0 = ce > cb ? cb / SCAN_BLOCK : 1, w1 = ce > cb ? (ce - 1) / SCAN_BLOCK : 0;
        __syncthreads();
        for (unsigned long long wbase = w0; wbase <= w1; wbase += DEDUP_THREADS) {
            // candidates of up to 512 blocks, flattened: prefix of their counts in LDS, then FUSE_PER per thread and round
            const unsigned long long w = wbase + tid;
            uint32_t cw = 0;
            unsigned long long first = 0;
            if (w <= w1) {
                const unsigned long long info = fx.blk_info[w];
                cw = scan_blk_count(info);
                first = scan_blk_first(info);
            }
            uint32_t total;
            const uint32_t before = block_excl_scan(cw, wsum, total);
            s_pref[tid] = before;
            s_first[tid] = first;
            // which block a flattened record index belongs to: a table of owners in the LDS room the sort uses later (2 bytes per
            // record; the nine-step search of the prefix array per record otherwise: 36 dependent LDS reads per thread and round)
            uint16_t *owner = reinterpret_cast<uint16_t *>(a + fx.lds_keys);
            const bool owned = total <= fx.bsort_keys * (uint32_t)(sizeof(K) / 2);
            if (owned)
                for (uint32_t j = 0; j < cw; j++) owner[before + j] = (uint16_t)tid;
            __syncthreads();
            KSSD_DEV_MARK_ONCE(dev_tA);
            for (uint32_t f0 = 0; f0 < total; f0 += DEDUP_THREADS * FUSE_PER) {
                bool ok[FUSE_PER];
                ulonglong2 cd[FUSE_PER];
                uint32_t dim[FUSE_PER];
                uint64_t u[FUSE_PER];
#pragma unroll
                for (int j = 0; j < FUSE_PER; j++) {
                    const uint32_t f = f0 + (uint32_t)j * DEDUP_THREADS + tid;
                    ok[j] = f < total;
                    cd[j] = make_ulonglong2(0ull, 0ull);
                    if (ok[j]) {
                        uint32_t lo = 0;
                        if (owned) {
                            lo = owner[f];
                        } else {
                            uint32_t hi = DEDUP_THREADS - 1;  // last block s with s_pref[s] <= f
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi + 1) >> 1;
                                if (s_pref[mid] <= f) lo = mid;
                                else hi = mid - 1;
                            }
                        }
                        cd[j] = fx.cand[s_first[lo] + (f - s_pref[lo])];
                    }
                }
#pragma unroll
                for (int j = 0; j < FUSE_PER; j++) {
                    const long long sp = (long long)cd[j].x, b0 = sp - P.out;
                    ok[j] = ok[j] && sp >= glo && sp < ghi && b0 >= glo && b0 + P.nb <= ghi;  // this genome's, and inside it
                    u[j] = 0;
                    dim[j] = 0;
                    if (ok[j]) {
                        const unsigned long long b0c = (unsigned long long)b0;
                        const bool known = (cd[j].y >> 63) != 0;
                        if (fx.carry) {
                            kssd_s2_canon(P, kssd_carry_fwd(P, cd[j].y & 0xFFFFFFFFFFull), u[j], dim[j]);
                            if (!known) {
                                const uint32_t *mp = fx.mask + (b0c >> 5);
                                const uint64_t m64 = (uint64_t)mp[0] | ((uint64_t)mp[1] << 32), need = (1ull << P.nb) - 1ull;
                                ok[j] = ((m64 >> (b0c & 31ull)) & need) == need;
                            }
                        } else {
                            const uint32_t *pp = fx.packed + (b0c >> 4), *mp = fx.mask + (b0c >> 5);
                            uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
                            if (!known) { m0 = mp[0]; m1 = mp[1]; }
                            ok[j] = kssd_s2_decode(P, pp[0], pp[1], pp[2], m0, m1, (uint32_t)b0c, u[j], dim[j]);
                        }
                    }
                }
                const KssdGBucket *GB = reinterpret_cast<const KssdGBucket *>(fx.G);
                KssdGBucket gb[FUSE_PER];
#pragma unroll
                for (int j = 0; j < FUSE_PER; j++) gb[j] = GB[kssd_g_slot(dim[j], P.g_mul[0], P.g_log2)];
#pragma unroll
                for (int j = 0; j < FUSE_PER; j++) {
                    uint32_t rank = 0;
                    int hit = kssd_g_match(gb[j], dim[j], rank);
                    if (hit == 2) hit = kssd_g_match(GB[kssd_g_slot(dim[j], P.g_mul[1], P.g_log2)], dim[j], rank) == 1 ? 1 : 0;
                    bool mine;
                    const uint32_t dr = kssd_s2_tuple(P, u[j], rank, mine);
                    ok[j] = ok[j] && hit == 1 && mine;
                    const uint64_t bal = __ballot(ok[j]);
                    if (bal) {  // one LDS atomic per wave reserves room for its survivors
                        uint32_t at = 0;
                        if (lane == 0) at = atomicAdd(&s_n, (uint32_t)__builtin_popcountll(bal));
                        at = __builtin_amdgcn_readfirstlane(at) + rank_in(bal);
                        if (ok[j] && at < fx.lds_keys) {
                            const uint32_t gpos = (uint32_t)((long long)cd[j].x - glo);
                            a[at] = fx.by_pos ? KeyOps<K>::make(gpos, dr) : KeyOps<K>::make(dr, gpos);
                        }
                    }
                }
                KSSD_DEV_MARK_ONCE(dev_tB);
            }
            __syncthreads();  // s_pref is rewritten by the next round of slices
        }
        __syncthreads();
        n = s_n;
    } else {
        n = cursor[g];
    }
    KSSD_DEV_MARK(dev_t1);
    const K *src = MODE != DEDUP_STAGED ? a : regions + r0;
    K *outp = regions + r0;  // where the kept keys go
    if (MODE == DEDUP_PARTS) outp = reinterpret_cast<K *>(px.out) + (((size_t)blockIdx.x << DEDUP_MAX_PARTS_LOG2) + part) * px.part_cap;
    if (MODE == DEDUP_RANGES) outp = reinterpret_cast<K *>(const_cast<void *>(px.parted)) + rng_s0;
    if (MODE != DEDUP_PARTS && MODE != DEDUP_RANGES && (n > cap || n > fx.lds_keys)) {
        if (tid == 0) {
            atomicOr(&st->region_overflow, 1u);
            // what the regions must grow by: the key array is 5/8 of the largest region (finish_sketch), and it is the array
            // that this genome may have outgrown
            const unsigned long long room = ((unsigned long long)cap * 5ull) / 8ull;
            unsigned long long need = ((unsigned long long)n * 256ull + room - 1) / (room ? room : 1);
            atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
            kept[g] = 0;
        }
        return;
    }
    if (MODE != DEDUP_RANGES && (flags & SKETCH_TRACK_FILL) && tid == 0)  // only while the regions are oversized after an overflow (see sketch_status)
        atomicMax(&st->max_need_q8, (uint32_t)(((unsigned long long)n * 256ull + cap - 1) / (cap ? cap : 1)));
    uint32_t np = 1;
    while (np < n) np <<= 1;
    if (tid == 0) { s_distinct = 0; s_zero_occ = 0; }
    // dynamic LDS: a[lds_keys] | b[bsort_keys] | counters[bsort_keys]  (lds_keys: the launch's array -- part_cap in PARTS mode)
    const K *sorted = a;
    bool bsorted = false;
    if (fx.bsort_keys && n > 64 && np <= fx.bsort_keys && !(MODE == DEDUP_RANGES && fx.by_pos)) {
        K *b = a + (MODE == DEDUP_PARTS || MODE == DEDUP_RANGES ? px.part_cap : fx.lds_keys);
        uint32_t *bcnt = reinterpret_cast<uint32_t *>(b + fx.bsort_keys);
        if (MODE == DEDUP_RANGES) bsorted = lds_bucket_sort<K>(src, b, bcnt, n, np, rng_id_base, rng_id_bits, 0xFFFFFFFFu, tid, wsum);
        else bsorted = lds_bucket_sort<K>(src, b, bcnt, n, np, 0u, fx.id_bits, DEDUP_BSORT_BUCKET, tid, wsum);
        if (bsorted) sorted = b;
    }
    if (bsorted) {
    } else if (np == 2 * DEDUP_THREADS) sort_in_registers<K, 2>(a, src, n, tid);
    else if (np == 4 * DEDUP_THREADS) sort_in_registers<K, 4>(a, src, n, tid);
    else if (np == 8 * DEDUP_THREADS) sort_in_registers<K, 8>(a, src, n, tid);
    else {
    for (uint32_t i = tid; i < np; i += DEDUP_THREADS)
        if (MODE == DEDUP_STAGED || i >= n) a[i] = i < n ? src[i] : KeyOps<K>::pad();  // (fused, parts: the keys are in place already)
    __syncthreads();
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (np >> 1); t += DEDUP_THREADS) {
                // t-th compare-exchange pair of this pass
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t p = i | j;
                const K x = a[i], y = a[p];
                const bool asc = (i & k) == 0;
                if ((x > y) == asc) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
    }
    KSSD_DEV_MARK(dev_t2);
    // runs of equal tuples; only the first n entries are real (in first-position mode the first entry of a run is
    // the tuple's first occurrence)
    uint32_t out_base = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += DEDUP_THREADS) {
        const uint32_t i = i0 + tid;
        bool keep = false, counted = false;
        K kv = 0;
        if (i < n) {
            kv = sorted[i];
            const uint32_t v = KeyOps<K>::id(kv);
            const bool start = (i == 0) || (KeyOps<K>::id(sorted[i - 1]) != v);
            if (start) {
                uint32_t lo = i + 1, hi = n;  // first index > i with a different tuple
                if (lo < n && KeyOps<K>::id(sorted[lo]) != v) hi = lo;  // the common case: a run of one
                while (lo < hi) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (KeyOps<K>::id(sorted[mid]) == v) lo = mid + 1;
                    else hi = mid;
                }
                const uint32_t len = lo - i;
                if (flags & KSSD_SKETCH_COUNTS) kv = KeyOps<K>::make(v, len < 65535u ? len : 65535u);  // OCCRC_MAX, global_basic.h
                keep = len >= min_occ;
                if ((flags & KSSD_SKETCH_UNIQ) && len > 1) keep = false;
                if (v == 0 && !(flags & KSSD_SKETCH_KEEP_ZERO)) {
                    keep = false;  // fasta2co leaves the slot empty (iseq2comem.c:258-261) ...
                    atomicAdd(&s_zero_occ, len);  // ... but counts every occurrence against the limit
                } else {
                    counted = true;
                }
            }
        }
        {   // distinct tuples: one LDS atomic per wave, not one per tuple (they would queue up on one address)
            const uint64_t cb = __ballot(counted);
            if (cb && lane_id() == 0) atomicAdd(&s_distinct, (uint32_t)__builtin_popcountll(cb));
        }
        uint32_t tot;
        const uint32_t pos = block_excl_scan(keep ? 1u : 0u, wsum, tot);
        if (keep) outp[out_base + pos] = kv;  // out_base+pos <= i: never overtakes unread input (input is in LDS)
        out_base += tot;
    }
    __syncthreads();
    KSSD_DEV_DO(if ((FUSED || MODE == DEDUP_RANGES) && fx.dev_times && tid == 0 && blockIdx.x < 65536) {
        fx.dev_times[blockIdx.x * 4] = dev_t0;
        fx.dev_times[blockIdx.x * 4 + 1] = fx.dev_split ? dev_tA : dev_t1;
        fx.dev_times[blockIdx.x * 4 + 2] = fx.dev_split ? dev_tB : dev_t2;
        fx.dev_times[blockIdx.x * 4 + 3] = fx.dev_split ? dev_t1 : __builtin_amdgcn_s_memrealtime();
    })
    if (tid == 0) {
        if (MODE == DEDUP_RANGES) {  // the genome's totals and its capacity rule: big_scan_kernel
            px.item_cnt[blockIdx.x] = out_base;
            px.item_s0[blockIdx.x] = (uint32_t)rng_s0;
            if (s_distinct) atomicAdd(&px.acc[0], s_distinct);
            if (s_zero_occ) atomicAdd(&px.acc[1], s_zero_occ);
        } else if (MODE == DEDUP_PARTS) {  // the genome's totals (and its capacity rule) are the finish kernel's
            pcnt[0] = out_base;
            pcnt[1] = s_distinct;
            pcnt[2] = s_zero_occ;
            pcnt[3] = 0;
        } else {
            kept[g] = out_base;
            if (!(flags & KSSD_SKETCH_NO_CAPACITY) && s_distinct + s_zero_occ > P.hashlimit) {
                // keycount > hashlimit (iseq2comem.c:261-263)
                atomicMax(&st->capacity_genome_p1, 0xFFFFFFFFu - g);  // keeps the smallest g
            }
        }
    }
}

// the parts of a medium genome back into its staging region, one behind the other (ascending: the parts are id ranges), and
// the genome's totals: kept ids, the capacity rule over all its distinct ids (iseq2comem.c:261-263)
template <typename K>
__global__ __launch_bounds__(256) void dedup_parts_finish_kernel(KssdParams P, const unsigned long long *__restrict__ reg_off, K *__restrict__ regions,
                                                                 uint32_t *__restrict__ kept, uint32_t flags, SketchStatus *st, PartArgs px)
{
    __shared__ uint32_t s_pre[(1 << DEDUP_MAX_PARTS_LOG2) + 1];
    __shared__ uint32_t s_bad;
    const uint2 md = px.list[blockIdx.x];
    const uint32_t g = md.x, n_parts = 1u << md.y;
    const uint32_t *cnt = px.cnt + ((size_t)blockIdx.x << (DEDUP_MAX_PARTS_LOG2 + 2));
    if (threadIdx.x == 0) {
        uint32_t run = 0, distinct = 0, zero_occ = 0, bad = 0;
        for (uint32_t p = 0; p < n_parts; p++) {
            s_pre[p] = run;
            run += cnt[p * 4];
            distinct += cnt[p * 4 + 1];
            zero_occ += cnt[p * 4 + 2];
            bad |= cnt[p * 4 + 3];
        }
        s_pre[n_parts] = run;
        s_bad = bad;
        if (blockIdx.y == 0) {
            kept[g] = bad ? 0u : run;
            if (!bad && !(flags & KSSD_SKETCH_NO_CAPACITY) && (unsigned long long)distinct + zero_occ > P.hashlimit)
                atomicMax(&st->capacity_genome_p1, 0xFFFFFFFFu - g);  // keeps the smallest g
        }
    }
    __syncthreads();
    if (s_bad) return;
    const unsigned long long r0 = reg_off[g];
    const K *out = reinterpret_cast<const K *>(px.out) + ((size_t)blockIdx.x << DEDUP_MAX_PARTS_LOG2) * px.part_cap;
    const uint32_t p = blockIdx.y;  // this workgroup's part (every workgroup of the genome computes the same prefix, the first one publishes the totals)
    if (p >= n_parts) return;
    const uint32_t b = s_pre[p], m = s_pre[p + 1] - b;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) regions[r0 + b + i] = out[(size_t)p * px.part_cap + i];
}

// ---------------------------------------------------------------------------------------------------
// kernel 2b: dedup of a genome whose staged tuples do not fit LDS (> DEDUP_MAX_N: > ~60 Mb at L3K10, FASTQ
// runs, chromosomes).  The staging region is padded with 0xFFFFFFFF, sorted by rocPRIM's device radix sort
// (a plain library sort: not worth a hand-written kernel for a handful of genomes per batch), then the same
// keep rules as sketch_dedup_kernel are applied by a count pass, a one-block scan and a write pass.
// ---------------------------------------------------------------------------------------------------
#define BIG_THREADS DEDUP_THREADS  // block_excl_scan is sized for it
#define BIG_TILE 2048  // sorted entries per workgroup

template <typename K>
__global__ void big_pad_kernel(K *__restrict__ region, unsigned long long cap, const uint32_t *__restrict__ cursor_g,
                               uint32_t *__restrict__ kept_g, uint32_t *__restrict__ acc /*[2]: distinct, zero occurrences*/,
                               SketchStatus *st)
{
    const unsigned long long n = *cursor_g;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc[0] = acc[1] = 0;
        if (n > cap) {
            atomicOr(&st->region_overflow, 1u);
            unsigned long long need = (n * 256ull + cap - 1) / (cap ? cap : 1);
            atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
            *kept_g = 0;
        } else {
            atomicMax(&st->max_need_q8, (uint32_t)((n * 256ull + cap - 1) / (cap ? cap : 1)));
        }
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (unsigned long long)gridDim.x * blockDim.x)
        if (i >= n) region[i] = KeyOps<K>::pad();
}

// WRITE = false: per-tile count of kept ids (+ the genome's distinct / zero-occurrence totals);
// WRITE = true: compaction into `out` at the tile's scanned offset
template <typename K, bool WRITE>
__global__ __launch_bounds__(BIG_THREADS) void big_runs_kernel(const K *__restrict__ a /*sorted*/, unsigned long long cap,
                                                                const uint32_t *__restrict__ cursor_g, uint32_t flags,
                                                                uint32_t min_occ, uint32_t *__restrict__ tile_cnt,
                                                                uint32_t *__restrict__ acc, K *__restrict__ out)
{
    __shared__ uint32_t wsum[DEDUP_THREADS / 64 + 1];
    const unsigned long long n = *cursor_g;
    if (n > cap) return;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * BIG_TILE;
    if (t0 >= n) { if (!WRITE && threadIdx.x == 0) tile_cnt[blockIdx.x] = 0; return; }
    uint32_t out_base = WRITE ? tile_cnt[blockIdx.x] : 0, distinct = 0, zero_occ = 0;
    for (uint32_t i0 = 0; i0 < BIG_TILE; i0 += BIG_THREADS) {
        const unsigned long long i = t0 + i0 + threadIdx.x;
        bool keep = false;
        K kv = 0;
        uint32_t v = 0;
        if (i < n) {
            kv = a[i];
            v = KeyOps<K>::id(kv);
            if (i == 0 || KeyOps<K>::id(a[i - 1]) != v) {
                // first index > i with a different tuple: gallop (runs are short -- one entry in a genome, a dozen in a read
                // set -- and a bisection of the whole array is 22 dependent reads for every one of them), then bisect the last stride
                unsigned long long lo = i + 1, d = 1;
                while (lo + d - 1 < n && KeyOps<K>::id(a[lo + d - 1]) == v) { lo += d; d <<= 1; }
                unsigned long long hi = lo + d - 1 < n ? lo + d - 1 : n;
                while (lo < hi) {
                    const unsigned long long mid = (lo + hi) >> 1;
                    if (KeyOps<K>::id(a[mid]) == v) lo = mid + 1;
                    else hi = mid;
                }
                const unsigned long long len = lo - i;
                if (flags & KSSD_SKETCH_COUNTS) kv = KeyOps<K>::make(v, len < 65535ull ? (uint32_t)len : 65535u);
                keep = len >= min_occ;
                if ((flags & KSSD_SKETCH_UNIQ) && len > 1) keep = false;
                if (v == 0 && !(flags & KSSD_SKETCH_KEEP_ZERO)) {
                    keep = false;  // fasta2co leaves the slot empty (iseq2comem.c:258-261) ...
                    zero_occ += (uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len);  // ... but counts every occurrence
                } else {
                    distinct++;
                }
            }
        }
        uint32_t tot;
        const uint32_t pos = block_excl_scan(keep ? 1u : 0u, wsum, tot);
        if (WRITE && keep) out[out_base + pos] = kv;
        out_base += tot;
    }
    if (!WRITE) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = out_base;
        uint32_t d_tot, z_tot;  // one atomic per workgroup, not one per thread
        block_excl_scan(distinct, wsum, d_tot);
        block_excl_scan(zero_occ, wsum, z_tot);
        if (threadIdx.x == 0) {
            if (d_tot) atomicAdd(&acc[0], d_tot);
            if (z_tot) atomicAdd(&acc[1], z_tot);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// kernel 2c: a large genome sorted in LDS after all (RANGES) -- the way read sets, chromosomes and --byread files take.
// The staged keys are partitioned by ranges of their leading field (id: its top bits are the canonical k-mer's outermost
// bases; --byread: the position), RNG_BIN_MEAN keys per bin on average:
//   big_rng_hist     every workgroup counts its share of the region per bin in LDS and writes its row of counts
//   big_rng_colscan  per bin: exclusive prefix over the workgroups' rows (where a workgroup's keys of the bin go), bin totals
//   big_rng_binscan  one workgroup: bin starts
//   big_rng_scatter  the same shares again: keys to their places (LDS cursors, no global atomic anywhere)
//   sketch_dedup_kernel<K, RANGES>  item k = the whole bins that begin in [k RNG_T, (k+1) RNG_T): LDS sort, keep rules, in place
//   big_scan_kernel  the items' kept counts -> offsets, the genome's total and capacity rule
//   big_rng_copy     the items' kept keys, one behind the other (the bins are ordered ranges: ascending), into the region
// Seven short launches (~50 us at 3.7 M staged keys) where the device radix sort of the whole region took four passes over
// all keys (0.2 ms).  Keys that do not spread -- ONE id tens of thousands of times (an amplicon, a spike-in), crafted ids --
// overflow an item's LDS array: the status says so (ranges_skew) and the repeated call sorts in global memory as before.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rng_bin(uint32_t lead, uint32_t shift, uint32_t last)
{
    const uint32_t b = lead >> shift;
    return b < last ? b : last;
}
__device__ __forceinline__ void rng_share(unsigned long long n, unsigned long long &lo, unsigned long long &hi)
{
    const unsigned long long per = (((n + gridDim.x - 1) / gridDim.x) + 1023ull) & ~1023ull;
    lo = (unsigned long long)blockIdx.x * per;
    hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}

template <typename K>
__global__ __launch_bounds__(1024) void big_rng_hist_kernel(const K *__restrict__ region, unsigned long long cap, const uint32_t *__restrict__ cursor_g,
                                                            uint32_t *__restrict__ kept_g, uint32_t *__restrict__ acc, SketchStatus *st,
                                                            uint32_t nb, uint32_t shift, uint32_t *__restrict__ wg_hist /*[gridDim.x][nb]*/)
{
    extern __shared__ uint32_t rng_h[];
    const unsigned long long n = *cursor_g;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc[0] = acc[1] = 0;
        const unsigned long long need = (n * 256ull + cap - 1) / (cap ? cap : 1);
        atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
        if (n > cap) {
            atomicOr(&st->region_overflow, 1u);
            *kept_g = 0;
        }
    }
    if (n > cap) return;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) rng_h[b] = 0;
    __syncthreads();
    unsigned long long lo, hi;
    rng_share(n, lo, hi);
    for (unsigned long long i0 = lo + threadIdx.x; i0 < hi; i0 += 4ull * blockDim.x) {
        K kv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) kv[j] = i0 + (unsigned long long)j * blockDim.x < hi ? region[i0 + (unsigned long long)j * blockDim.x] : (K)0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (i0 + (unsigned long long)j * blockDim.x < hi) atomicAdd(&rng_h[rng_bin(KeyOps<K>::id(kv[j]), shift, nb - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) wg_hist[(size_t)blockIdx.x * nb + b] = rng_h[b];
}

__global__ __launch_bounds__(256) void big_rng_colscan_kernel(unsigned long long cap, const uint32_t *__restrict__ cursor_g, uint32_t nb, uint32_t rows,
                                                               uint32_t *__restrict__ wg_hist, uint32_t *__restrict__ bin_tot)
{
    if (*cursor_g > cap) return;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t run = 0;
    for (uint32_t w0 = 0; w0 < rows; w0 += 8) {  // eight rows' counts in flight
        uint32_t t[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) t[j] = w0 + j < rows ? wg_hist[(size_t)(w0 + j) * nb + b] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (w0 + j < rows) wg_hist[(size_t)(w0 + j) * nb + b] = run;
            run += t[j];
        }
    }
    bin_tot[b] = run;
}

__global__ __launch_bounds__(DEDUP_THREADS) void big_rng_binscan_kernel(unsigned long long cap, const uint32_t *__restrict__ cursor_g, uint32_t nb,
                                                                         const uint32_t *__restrict__ bin_tot, uint32_t *__restrict__ bin_start /*nb+1*/,
                                                                         uint32_t *__restrict__ item_bin /*n_items+1*/, uint32_t n_items)
{
    extern __shared__ uint32_t rng_h[];  // nb + 1 bin starts
    __shared__ uint32_t wsum[DEDUP_THREADS / 64 + 1];
    if (*cursor_g > cap) return;
    for (uint32_t b = threadIdx.x; b < nb; b += DEDUP_THREADS) rng_h[b] = bin_tot[b];  // coalesced in, scanned in LDS, coalesced out
    __syncthreads();
    // tiles of DEDUP_THREADS consecutive bins, one per thread (a thread that walks its own run of the array meets every other
    // lane of its wave in one LDS bank)
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < nb; t0 += DEDUP_THREADS) {
        const uint32_t b = t0 + threadIdx.x;
        const uint32_t v = b < nb ? rng_h[b] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, wsum, total);
        if (b < nb) rng_h[b] = carry + ex;
        carry += total;
    }
    const uint32_t total = carry;
    if (threadIdx.x == 0) rng_h[nb] = total;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= nb; b += DEDUP_THREADS) bin_start[b] = rng_h[b];
    // item k = the whole bins that begin in [k RNG_T, (k+1) RNG_T): item_bin[k] = the first bin whose start is >= k RNG_T
    // (bin b is that bin for every k with start[b-1] < k RNG_T <= start[b]); beyond the last bin: nb
    for (uint32_t b = threadIdx.x; b <= nb; b += DEDUP_THREADS) {
        const uint32_t s1 = rng_h[b];
        uint32_t k0 = b ? rng_h[b - 1] / RNG_T + 1u : 0u;  // first k with k RNG_T > start[b-1]
        for (uint32_t k = k0; k <= n_items && (unsigned long long)k * RNG_T <= s1; k++) item_bin[k] = b;
    }
    for (uint32_t k = total / RNG_T + 1u + threadIdx.x; k <= n_items; k += DEDUP_THREADS) item_bin[k] = nb;
}

template <typename K>
__global__ __launch_bounds__(1024) void big_rng_scatter_kernel(const K *__restrict__ region, unsigned long long cap, const uint32_t *__restrict__ cursor_g,
                                                               uint32_t nb, uint32_t shift, const uint32_t *__restrict__ wg_hist,
                                                               const uint32_t *__restrict__ bin_start, K *__restrict__ parted)
{
    extern __shared__ uint32_t rng_h[];
    const unsigned long long n = *cursor_g;
    if (n > cap) return;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) rng_h[b] = bin_start[b] + wg_hist[(size_t)blockIdx.x * nb + b];
    __syncthreads();
    unsigned long long lo, hi;
    rng_share(n, lo, hi);
    for (unsigned long long i0 = lo + threadIdx.x; i0 < hi; i0 += 4ull * blockDim.x) {
        K kv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) kv[j] = i0 + (unsigned long long)j * blockDim.x < hi ? region[i0 + (unsigned long long)j * blockDim.x] : (K)0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (i0 + (unsigned long long)j * blockDim.x < hi) parted[atomicAdd(&rng_h[rng_bin(KeyOps<K>::id(kv[j]), shift, nb - 1u)], 1u)] = kv[j];
    }
}

template <typename K>
__global__ __launch_bounds__(256) void big_rng_copy_kernel(unsigned long long cap, const uint32_t *__restrict__ cursor_g, const K *__restrict__ parted,
                                                            const uint32_t *__restrict__ item_off /*scanned*/, const uint32_t *__restrict__ item_s0,
                                                            uint32_t n_items, const uint32_t *__restrict__ kept_g, K *__restrict__ region)
{
    if (*cursor_g > cap) return;
    const uint32_t k = blockIdx.x;
    const uint32_t o0 = item_off[k], o1 = k + 1 < n_items ? item_off[k + 1] : *kept_g;
    const K *src = parted + item_s0[k];
    for (uint32_t i = threadIdx.x; i < o1 - o0; i += blockDim.x) region[o0 + i] = src[i];
}

// one workgroup: exclusive scan of the tile counts, kept[g], the capacity rule (iseq2comem.c:261-263)
__global__ __launch_bounds__(1024) void big_scan_kernel(uint32_t *__restrict__ tile_cnt, uint32_t n_tiles, const uint32_t *__restrict__ acc,
                                                         uint32_t hashlimit, uint32_t flags, uint32_t g, unsigned long long cap,
                                                         const uint32_t *__restrict__ cursor_g, uint32_t *__restrict__ kept_g,
                                                         SketchStatus *st)
{
    __shared__ uint32_t part[1024];
    if (*cursor_g > cap) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_tiles + 1023) / 1024;
    const uint32_t b = tid * per, e = (b + per < n_tiles) ? b + per : n_tiles;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += tile_cnt[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; i++) { const uint32_t t = part[i]; part[i] = run; run += t; }
        *kept_g = run;
        if (!(flags & KSSD_SKETCH_NO_CAPACITY) && (unsigned long long)acc[0] + acc[1] > hashlimit)
            atomicMax(&st->capacity_genome_p1, 0xFFFFFFFFu - g);  // keeps the smallest g
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (uint32_t i = b; i < e; i++) { const uint32_t t = tile_cnt[i]; tile_cnt[i] = run; run += t; }
}

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
template <int SUBK, int ABL = 0>
static int launch_scan(kssd_gpu_ctx *c, const ScanArgs &a, int grid, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    hipExtLaunchKernelGGL((sketch_scan_kernel<SUBK, ABL>), dim3(grid), dim3(SCAN_THREADS), 0, s, ev_start, ev_stop, 0, a);
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

// per-genome dedup (LDS sort or, for large genomes, rocPRIM sort + run kernels), CSR offsets, gather; K = key type
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
        fx.cand = reinterpret_cast<const ulonglong2 *>(c->d_cand);
        fx.blk_info = c->d_blk_info;
        fx.chunk_off = (const unsigned long long *)c->d_chunk_off;
        fx.packed = pl.d_packed;
        fx.mask = pl.d_mask;
        fx.G = c->d_G;
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
    } else {
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
        if ((rc = ensure(&c->d_big_alt, &c->cap_big_alt, (size_t)max_big * kw + n_tiles_max + 8 + rng_words)) != KSSD_OK) return rc;
        if (!ranges) {
            size_t tmp_bytes = 0;
            HIPCK(rocprim::radix_sort_keys(nullptr, tmp_bytes, (K *)nullptr, (K *)nullptr, (size_t)max_big, 0u, (unsigned)(8 * sizeof(K)), s));
            if (tmp_bytes > c->cap_big_tmp) {
                if (c->d_big_tmp) hipFree(c->d_big_tmp);
                c->d_big_tmp = nullptr;
                c->cap_big_tmp = 0;
                if (hipMalloc(&c->d_big_tmp, tmp_bytes) != hipSuccess) return KSSD_ERR_NOMEM;
                c->cap_big_tmp = tmp_bytes;
            }
        }
        for (uint32_t g : c->h_big) {
            const uint64_t r0 = c->h_reg_off[g], cap = c->h_reg_off[g + 1] - r0;
            K *region = regions + r0, *sorted = reinterpret_cast<K *>(c->d_big_alt);
            uint32_t *tile_cnt = c->d_big_alt + (size_t)max_big * kw, *accum = tile_cnt + n_tiles_max;
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
            size_t tb = c->cap_big_tmp;
            HIPCK(rocprim::radix_sort_keys(c->d_big_tmp, tb, region, sorted, (size_t)cap, 0u, (unsigned)(8 * sizeof(K)), s));
            hipLaunchKernelGGL((big_runs_kernel<K, false>), dim3(n_tiles), dim3(BIG_THREADS), 0, s, (const K *)sorted,
                               (unsigned long long)cap, cur, flags, min_occ, tile_cnt, accum, (K *)nullptr);
            hipLaunchKernelGGL(big_scan_kernel, dim3(1), dim3(1024), 0, s, tile_cnt, n_tiles, (const uint32_t *)accum, c->P.hashlimit,
                               flags, g, (unsigned long long)cap, cur, c->d_kept + g, c->d_status);
            hipLaunchKernelGGL((big_runs_kernel<K, true>), dim3(n_tiles), dim3(BIG_THREADS), 0, s, (const K *)sorted,
                               (unsigned long long)cap, cur, flags, min_occ, tile_cnt, accum, region);
        }
    }
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
extern "C" int kssd_gpu_sketch_plan(kssd_gpu_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask,
                                    const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                    uint64_t *d_out_off, uint32_t *d_out_ids, uint64_t out_cap)
{
    if (!c) return KSSD_ERR_PARAM;
    c->plan.valid = false;
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
        if (cap > pos) cap = pos;  // a genome cannot emit more tuples than it has positions
        if (cap > big_min) {
            // a genome expected to fit the LDS sort keeps that path -- unless an earlier attempt has shown that this
            // batch emits far more than the sampling rate predicts (low-complexity sequence): then the factor decides
            if (c->region_factor <= 2.0 && (uint64_t)((double)pos * rate * 1.25) + 64 <= big_min) cap = big_min;
            else if (!(flags & KSSD_SKETCH_BY_POS) && cap <= ((uint64_t)big_min << DEDUP_MAX_PARTS_LOG2)) {
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
        if (cap <= big_min && cap > max_cap) max_cap = cap;
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
        hipLaunchKernelGGL(chunk_gid_by_genome_kernel, dim3(pl.n_genomes > 64 ? pl.n_genomes : 64, (unsigned)(parts ? parts : 1)), dim3(256), 0, s,
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
    a.cand = reinterpret_cast<ulonglong2 *>(c->d_cand); a.cand_cap = pl.cand_cap; a.cand_count = c->d_cand_count;
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
    hipEvent_t e0 = c->ev_a[0][evi], e1 = c->ev_b[0][evi];
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
    c->ev_n[0]++;
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
    x.chunk_off = (const unsigned long long *)c->d_chunk_off; x.G = c->d_G;
    x.cand = reinterpret_cast<const ulonglong2 *>(c->d_cand); x.cand_cap = pl.cand_cap; x.cand_count = c->d_cand_count;
    x.carry = kssd_carry_ok(c->P) ? 1u : 0u;
    x.n_slices = pl.n_slices;
    x.reg_off = (const unsigned long long *)c->d_reg_off; x.cursor = c->d_cursor; x.regions = c->d_regions;
    x.by_pos = (pl.flags & KSSD_SKETCH_BY_POS) ? 1u : 0u;
#ifdef KSSD_DEV
    x.dev_no_atomic = getenv("KSSD_DEV_EXACT_NO_ATOMIC") ? 1u : 0u;
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
    if (st.cand_overflow) {
        c->cand_floor = (uint64_t)st.cand_need + st.cand_need / 4 + 64;
        return KSSD_ERR_OVERFLOW;
    }
    if (st.ranges_skew && !st.region_overflow) {
        c->ranges_off = true;
        c->ranges_off_calls = 4;  // the repeated call and the further passes callers make over the same batch (occurrences, abundances)
        return KSSD_ERR_OVERFLOW;
    }
    if (st.region_overflow) {
        double need = (double)st.max_need_q8 / 256.0;  // emitted / capacity of the worst genome
        c->region_factor *= (need > 1.0 ? need : 1.0) * 1.25;
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
    if ((rc = ensure(&c->d_b_off, &c->cap_b_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if (out_pos && !(flags & (KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS))) flags |= KSSD_SKETCH_FIRST_POS;
    if (!out_pos) flags &= ~(KSSD_SKETCH_FIRST_POS | KSSD_SKETCH_COUNTS | KSSD_SKETCH_BY_POS);
    const double rate = (double)c->P.dim_end / (double)(1ull << (4 * c->P.subk));
    uint64_t out_cap = (uint64_t)((double)n_chunks * KSSD_CHUNK * rate * 1.5) + 1024;
    uint64_t total = 0;
    for (int attempt = 0; attempt < 12; attempt++) {
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
    return sketch_resident_impl(c, chunk_off, n_genomes, flags, min_occ, out_off, out_ids, out_pos, bad_genome);
}

// page-locked host memory for batches that travel by DMA (hipHostMalloc); NULL when it cannot be had
extern "C" void *kssd_gpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
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

extern "C" int kssd_gpu_kernel_time(kssd_gpu_ctx *c, int which, int reset, float *avg_ms, uint32_t *launches)
{
    if (!c || which < 0 || which > 1) return KSSD_ERR_PARAM;
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
    rc = tokenise_device_impl(c, c->d_text, text_off, text_len, n_files, c->d_in_packed, c->d_in_mask, chunk_off.data(), s, fq);
    if (rc != KSSD_OK) return rc;
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
