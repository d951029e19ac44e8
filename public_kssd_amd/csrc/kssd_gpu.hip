// kssd_gpu.hip -- gfx950 (MI355X / CDNA4) kernels and the C ABI of libkssd_gpu.so.
//
// Hot path of kssd: genome sketching (reference iseq2comem.c:188-356) and the
// shared-k-mer intersection + distances (co2mco.c:25-77, command_dist.c:763-790,1251-1266).
// Integer / indexing work: HBM streaming, LDS tables, wavefront ballot/scan.  No MFMA on purpose.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC kssd_gpu.hip -o libkssd_gpu.so
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/kssd_gpu.h"
#include "kssd_core.h"

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
    default: return "unknown kssd_gpu error";
    }
}

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
#define SCAN_THREADS 1024
#define SCAN_WAVES (SCAN_THREADS / 64)
#define QCAP 256          // per-wave candidate queue (u32: chunk offset << 12 | position), lives across chunks
#define EBUF 64           // per-wave emission buffer (u32 reduced tuples)
#define DEDUP_THREADS 256
#define DEDUP_MAX_N 32768 // ids one workgroup can sort in LDS (128 KiB)
#define EV_RING 128
#define SCAN_LDS_BYTES (KSSD_T1_BYTES + SCAN_WAVES * QCAP * 4 + SCAN_WAVES * EBUF * 4)

struct SketchStatus {
    unsigned long long total_ids;
    unsigned int region_overflow;  // some genome emitted more than its staging region holds
    unsigned int out_overflow;     // d_out_ids too small
    unsigned int max_need_q8;      // max over genomes of emitted / capacity, in 1/256 units
    unsigned int capacity_genome_p1;  // 1 + first genome over the reference's hash limit (0 = none)
};

struct kssd_gpu_ctx {
    int device;
    int cu_count;
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
    uint32_t *d_regions;
    size_t cap_regions;
    SketchStatus *d_status;
    double region_factor;
    uint32_t last_n_genomes;
    int last_launch_rc;
    std::vector<uint64_t> h_reg_off;
    // inverted index (dist)
    uint32_t n_ref;
    uint64_t n_ref_ids;
    uint32_t *d_ref_sz;     // n_ref sketch sizes
    uint32_t *d_hkeys;      // 4 arrays of 2^h_log2 u32: key | count | start | fill cursor
    uint32_t h_log2;
    size_t cap_hash;
    uint32_t *d_post;       // postings (genome indices) | entry->slot scratch | scan partials
    size_t cap_pairs;
    size_t cap_ref;
    // timing: ring of HIP event pairs around the dominant kernel of each path (0 = sketch scan, 1 = dist rows)
    hipEvent_t ev_a[2][EV_RING], ev_b[2][EV_RING];
    unsigned ev_n[2];
};

static int ctx_upload_tables(kssd_gpu_ctx *c, const std::vector<uint32_t> &accepted)
{
    const KssdParams &P = c->P;
    std::vector<uint8_t> T1;
    std::vector<KssdG> G;
    kssd_build_tables(P, accepted, T1, G);
    const size_t gn = G.size();
    HIPCK(hipMalloc(&c->d_T1, KSSD_T1_BYTES));
    HIPCK(hipMalloc(&c->d_G, gn * sizeof(KssdG)));
    HIPCK(hipMemcpy(c->d_T1, T1.data(), KSSD_T1_BYTES, hipMemcpyHostToDevice));
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
                    c->d_regions, c->d_status, c->d_ref_sz, c->d_hkeys, c->d_post};
    for (void *p : ptrs)
        if (p) hipFree(p);
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

// ---------------------------------------------------------------------------------------------------
// kernel 0: chunk -> genome map
// ---------------------------------------------------------------------------------------------------
__global__ void chunk_gid_kernel(const uint64_t *__restrict__ chunk_off, uint32_t n_genomes, uint64_t n_chunks,
                                 uint32_t *__restrict__ chunk_gid)
{
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t lo = 0, hi = n_genomes;  // last g with chunk_off[g] <= c
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid;
        else hi = mid;
    }
    chunk_gid[c] = lo;
}

// ---------------------------------------------------------------------------------------------------
// kernel 1: the scan.  One wave per 4096-position chunk iteration, one lane per 64 positions.
//   HBM -> registers: 16 B of packed bases + 8 B of mask per lane, coalesced, next chunk prefetched
//   stage 1: 33 LDS nibble reads per lane decide 64 positions (quad-core table, 128 KiB of LDS)
//   ballot compaction of the ~0.3 % candidate positions into a per-wave LDS queue that lives ACROSS
//   chunks: stage 2 only runs on full rounds of 64 candidates (it costs the same for 1 lane as for 64)
//   stage 2: exact evaluation, survivors staged in LDS and appended to the genome's region
// ---------------------------------------------------------------------------------------------------
struct ScanArgs {
    const uint32_t *packed;
    const uint32_t *mask;
    const uint32_t *chunk_gid;
    const unsigned long long *chunk_off;  // per genome, in chunks
    unsigned long long n_chunks;
    const uint8_t *T1;
    const KssdG *G;
    const unsigned long long *reg_off;
    uint32_t *cursor;
    uint32_t *regions;
};

__device__ __forceinline__ void flush_emissions(const ScanArgs &a, uint32_t gid, uint32_t n, const uint32_t *ebuf,
                                                uint32_t lane)
{
    // one returning atomic per flush, not per k-mer: a single cursor word saturates at ~90 atomics/us
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.cursor[gid], n);
    base = __builtin_amdgcn_readfirstlane(base);
    const unsigned long long r0 = a.reg_off[gid];
    const unsigned long long cap = a.reg_off[gid + 1] - r0;
    if (lane < n && (unsigned long long)base + lane < cap) a.regions[r0 + base + lane] = ebuf[lane];
}

// per-wave state of the scan that the helpers below share
struct WaveState {
    uint32_t *queue;   // LDS, QCAP entries: (chunk - c0) << 12 | position in chunk
    uint32_t *ebuf;    // LDS, EBUF reduced tuples waiting for the next region append
    uint32_t qn;       // queued candidates (wave-uniform)
    uint32_t ecount;   // staged emissions (wave-uniform)
    uint32_t gid;      // genome all queued candidates and staged emissions belong to
    long long glo, ghi;  // positions of that genome: [glo, ghi)
    unsigned long long c0;
};

// stage 2 for one round of up to 64 queued candidates (entries [first, first+n) of the queue)
__device__ __forceinline__ void stage2_round(const KssdParams &P, const ScanArgs &a, WaveState &w, uint32_t first, uint32_t n,
                                             uint32_t lane)
{
    bool ok = false;
    uint32_t dr = 0;
    if (lane < n) {
        const uint32_t e = w.queue[first + lane];
        const long long s = (long long)((w.c0 + (e >> 12)) * KSSD_CHUNK) + (long long)(e & 4095u);
        ok = kssd_stage2(P, s, w.glo, w.ghi, a.packed, a.mask, a.G, dr);
    }
    const uint64_t bal = __ballot(ok);
    const uint32_t m = __builtin_popcountll(bal);
    if (m) {
        if (w.ecount + m > EBUF) {
            wave_lds_sync();
            flush_emissions(a, w.gid, w.ecount, w.ebuf, lane);
            wave_lds_sync();
            w.ecount = 0;
        }
        if (ok) w.ebuf[w.ecount + rank_in(bal)] = dr;
        w.ecount += m;
    }
}

// run stage 2 on full rounds of 64 (all = false) or on everything that is queued (all = true)
__device__ __forceinline__ void drain_queue(const KssdParams &P, const ScanArgs &a, WaveState &w, bool all, uint32_t lane)
{
    wave_lds_sync();
    while (w.qn >= 64) {
        w.qn -= 64;
        stage2_round(P, a, w, w.qn, 64, lane);
    }
    if (all && w.qn) {
        stage2_round(P, a, w, 0, w.qn, lane);
        w.qn = 0;
    }
    wave_lds_sync();
}

// ABL != 0: development-only ablations for profiling (1 = loads only, 2 = + stage 1, 3 = + queue push);
// never used by the product path
template <int SUBK, int ABL = 0>
__global__ __launch_bounds__(SCAN_THREADS) void sketch_scan_kernel(KssdParams P, ScanArgs a)
{
    uint32_t abl_acc = 0;
    // static LDS: the table sits at LDS address 0, so its byte reads need no base add
    __shared__ __attribute__((aligned(16))) unsigned char smem[SCAN_LDS_BYTES];
    uint8_t *T1 = smem;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = lane_id();
    WaveState w;
    w.queue = reinterpret_cast<uint32_t *>(smem + KSSD_T1_BYTES) + wave * QCAP;
    w.ebuf = reinterpret_cast<uint32_t *>(smem + KSSD_T1_BYTES + SCAN_WAVES * QCAP * 4) + wave * EBUF;
    w.qn = 0;
    w.ecount = 0;

    for (uint32_t i = threadIdx.x * 16; i < KSSD_T1_BYTES; i += SCAN_THREADS * 16)
        *reinterpret_cast<uint4 *>(T1 + i) = *reinterpret_cast<const uint4 *>(a.T1 + i);
    __syncthreads();

    // static partition: every wave of the grid owns one contiguous run of chunks, so a wave stays inside
    // one genome for long stretches and concurrent waves append to different genomes
    const unsigned long long total_waves = (unsigned long long)gridDim.x * SCAN_WAVES;
    const unsigned long long wid = (unsigned long long)blockIdx.x * SCAN_WAVES + wave;
    const unsigned long long per = (a.n_chunks + total_waves - 1) / total_waves;
    unsigned long long c0 = wid * per, c1 = c0 + per;
    if (c1 > a.n_chunks) c1 = a.n_chunks;
    if (c0 >= c1) return;
    w.c0 = c0;
    w.gid = 0xFFFFFFFFu;
    w.glo = w.ghi = 0;

    uint32_t W[5], M[2];
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(a.packed + c0 * 256 + lane * 4);
        W[0] = v.x; W[1] = v.y; W[2] = v.z; W[3] = v.w;
        W[4] = a.packed[c0 * 256 + lane * 4 + 4];
        const uint2 m = *reinterpret_cast<const uint2 *>(a.mask + c0 * 128 + lane * 2);
        M[0] = m.x; M[1] = m.y;
    }
    uint32_t gid_next = a.chunk_gid[c0];
    for (unsigned long long c = c0; c < c1; ++c) {
        // prefetch the next chunk while this one is processed
        uint32_t Wn[5] = {0, 0, 0, 0, 0}, Mn[2] = {0, 0};
        const uint32_t gid = gid_next;
        if (c + 1 < c1) {
            const uint4 v = *reinterpret_cast<const uint4 *>(a.packed + (c + 1) * 256 + lane * 4);
            Wn[0] = v.x; Wn[1] = v.y; Wn[2] = v.z; Wn[3] = v.w;
            Wn[4] = a.packed[(c + 1) * 256 + lane * 4 + 4];
            const uint2 m = *reinterpret_cast<const uint2 *>(a.mask + (c + 1) * 128 + lane * 2);
            Mn[0] = m.x; Mn[1] = m.y;
            gid_next = a.chunk_gid[c + 1];
        }
        if (ABL == 1) {
            abl_acc ^= W[0] ^ W[1] ^ W[2] ^ W[3] ^ W[4] ^ M[0] ^ M[1] ^ gid;
        } else {
            if (gid != w.gid) {
                // genome boundary: everything queued or staged belongs to the previous genome
                if (w.qn) drain_queue(P, a, w, true, lane);
                if (w.ecount) { wave_lds_sync(); flush_emissions(a, w.gid, w.ecount, w.ebuf, lane); wave_lds_sync(); }
                w.ecount = 0;
                w.gid = gid;
                w.glo = (long long)(a.chunk_off[gid] * KSSD_CHUNK);
                w.ghi = (long long)(a.chunk_off[gid + 1] * KSSD_CHUNK);
            }
            uint32_t cl, ch;
            kssd_stage1<SUBK>(W, T1, cl, ch);
            cl &= M[0];  // the window start itself must be a base: kills padding / N stretches early
            ch &= M[1];
            if (ABL == 2) {
                abl_acc ^= cl ^ ch;
            } else {
                // ballot compaction: every pass moves one candidate of every lane that still has one
                const uint32_t ebase = ((uint32_t)(c - c0) << 12) | (lane << 6);
                for (;;) {
                    const bool has = (cl | ch) != 0;
                    const uint64_t bal = __ballot(has);
                    if (!bal) break;
                    if (w.qn + 64 > QCAP) drain_queue(P, a, w, false, lane);
                    if (has) {
                        uint32_t b;
                        if (cl) { b = __builtin_ctz(cl); cl &= cl - 1; }
                        else { b = 32 + __builtin_ctz(ch); ch &= ch - 1; }
                        w.queue[w.qn + rank_in(bal)] = ebase | b;
                    }
                    w.qn += __builtin_popcountll(bal);
                }
                if (ABL == 3) {
                    if (w.qn >= 64) { wave_lds_sync(); abl_acc ^= w.queue[lane]; w.qn = 0; wave_lds_sync(); }
                } else if (w.qn >= 64) {
                    drain_queue(P, a, w, false, lane);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) W[i] = Wn[i];
        M[0] = Mn[0];
        M[1] = Mn[1];
    }
    if (ABL == 0 || ABL == 3) {
        if (w.qn && ABL == 0) drain_queue(P, a, w, true, lane);
        if (w.ecount) { wave_lds_sync(); flush_emissions(a, w.gid, w.ecount, w.ebuf, lane); }
    }
    if (ABL != 0 && abl_acc == 0x9e3779b9u) a.regions[0] = abl_acc;  // keeps the ablated work alive
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

__global__ __launch_bounds__(DEDUP_THREADS) void sketch_dedup_kernel(KssdParams P, const unsigned long long *__restrict__ reg_off,
                                                                      const uint32_t *__restrict__ cursor,
                                                                      uint32_t *__restrict__ regions, uint32_t *__restrict__ kept,
                                                                      uint32_t flags, uint32_t min_occ, SketchStatus *st)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *a = reinterpret_cast<uint32_t *>(smem);
    __shared__ uint32_t wsum[DEDUP_THREADS / 64 + 1];
    __shared__ uint32_t s_distinct, s_zero_occ;
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const unsigned long long r0 = reg_off[g];
    const uint32_t cap = (uint32_t)(reg_off[g + 1] - r0);
    uint32_t n = cursor[g];
    if (n > cap) {
        if (tid == 0) {
            atomicOr(&st->region_overflow, 1u);
            unsigned long long need = ((unsigned long long)n * 256ull + cap - 1) / (cap ? cap : 1);
            atomicMax(&st->max_need_q8, (uint32_t)(need > 0xFFFFFFFFull ? 0xFFFFFFFFull : need));
            kept[g] = 0;
        }
        return;
    }
    uint32_t np = 1;
    while (np < n) np <<= 1;
    for (uint32_t i = tid; i < np; i += DEDUP_THREADS) a[i] = i < n ? regions[r0 + i] : 0xFFFFFFFFu;
    if (tid == 0) { s_distinct = 0; s_zero_occ = 0; }
    __syncthreads();
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (np >> 1); t += DEDUP_THREADS) {
                // t-th compare-exchange pair of this pass
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t p = i | j;
                const uint32_t x = a[i], y = a[p];
                const bool asc = (i & k) == 0;
                if ((x > y) == asc) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
    // runs of equal values; only the first n entries are real
    uint32_t out_base = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += DEDUP_THREADS) {
        const uint32_t i = i0 + tid;
        bool keep = false;
        uint32_t v = 0;
        if (i < n) {
            v = a[i];
            const bool start = (i == 0) || (a[i - 1] != v);
            if (start) {
                uint32_t lo = i + 1, hi = n;  // first index > i with a different value
                while (lo < hi) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (a[mid] == v) lo = mid + 1;
                    else hi = mid;
                }
                const uint32_t len = lo - i;
                keep = len >= min_occ;
                if ((flags & KSSD_SKETCH_UNIQ) && len > 1) keep = false;
                if (v == 0 && !(flags & KSSD_SKETCH_KEEP_ZERO)) {
                    keep = false;  // fasta2co leaves the slot empty (iseq2comem.c:258-261) ...
                    atomicAdd(&s_zero_occ, len);  // ... but counts every occurrence against the limit
                } else {
                    atomicAdd(&s_distinct, 1u);
                }
            }
        }
        uint32_t tot;
        const uint32_t pos = block_excl_scan(keep ? 1u : 0u, wsum, tot);
        if (keep) regions[r0 + out_base + pos] = v;  // out_base+pos <= i: never overtakes unread input (input is in LDS)
        out_base += tot;
    }
    __syncthreads();
    if (tid == 0) {
        kept[g] = out_base;
        if (!(flags & KSSD_SKETCH_NO_CAPACITY) && s_distinct + s_zero_occ > P.hashlimit) {
            // keycount > hashlimit (iseq2comem.c:261-263)
            atomicMax(&st->capacity_genome_p1, 0xFFFFFFFFu - g);  // keeps the smallest g
        }
    }
}

// kernel 3: exclusive scan of the kept counts -> CSR offsets (single workgroup, n_genomes is small)
__global__ __launch_bounds__(1024) void sketch_offsets_kernel(const uint32_t *__restrict__ kept, uint32_t n,
                                                               unsigned long long *__restrict__ out_off,
                                                               unsigned long long out_cap, SketchStatus *st)
{
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = tid * per, e = (b + per < n) ? b + per : n;
    unsigned long long s = 0;
    for (uint32_t i = b; i < e; i++) s += kept[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { unsigned long long t = part[i]; part[i] = run; run += t; }
        out_off[n] = run;
        st->total_ids = run;
        if (run > out_cap) st->out_overflow = 1;
    }
    __syncthreads();
    unsigned long long run = part[tid];
    for (uint32_t i = b; i < e; i++) { out_off[i] = run; run += kept[i]; }
}

// kernel 4: gather every genome's kept ids into the dense CSR
__global__ __launch_bounds__(256) void sketch_gather_kernel(const unsigned long long *__restrict__ reg_off,
                                                             const uint32_t *__restrict__ regions,
                                                             const uint32_t *__restrict__ kept,
                                                             const unsigned long long *__restrict__ out_off,
                                                             uint32_t *__restrict__ out_ids, const SketchStatus *st)
{
    if (st->out_overflow) return;
    const uint32_t g = blockIdx.x;
    const uint32_t n = kept[g];
    const unsigned long long r0 = reg_off[g], o0 = out_off[g];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out_ids[o0 + i] = regions[r0 + i];
}

// ---------------------------------------------------------------------------------------------------
// sketch entry points
// ---------------------------------------------------------------------------------------------------
template <int SUBK, int ABL = 0>
static int launch_scan(kssd_gpu_ctx *c, const ScanArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((sketch_scan_kernel<SUBK, ABL>), dim3(grid), dim3(SCAN_THREADS), 0, s, c->P, a);
    HIPCK(hipGetLastError());
    return KSSD_OK;
}

extern "C" int kssd_gpu_sketch_device(kssd_gpu_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask,
                                      const uint64_t *h_chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                      uint64_t *d_out_off, uint32_t *d_out_ids, uint64_t out_cap, void *stream)
{
    if (!c || c->dist_only || !h_chunk_off || !d_out_off || (!d_out_ids && out_cap)) return KSSD_ERR_PARAM;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipSetDevice(c->device));
    c->last_launch_rc = KSSD_OK;
    c->last_n_genomes = n_genomes;
    if (min_occ < 1) min_occ = 1;
    const uint64_t n_chunks = h_chunk_off[n_genomes];
    HIPCK(hipMemsetAsync(c->d_status, 0, sizeof(SketchStatus), s));
    if (n_genomes == 0) {
        HIPCK(hipMemsetAsync(d_out_off, 0, sizeof(uint64_t), s));
        return KSSD_OK;
    }
    // staging regions: expected emissions = positions * dim_end / 16^subk, times a safety factor
    const double rate = (double)c->P.dim_end / (double)(1ull << (4 * c->P.subk));
    c->h_reg_off.resize((size_t)n_genomes + 1);
    uint64_t acc = 0, max_cap = 0;
    for (uint32_t g = 0; g < n_genomes; g++) {
        if (h_chunk_off[g + 1] < h_chunk_off[g]) return KSSD_ERR_PARAM;
        const uint64_t pos = (h_chunk_off[g + 1] - h_chunk_off[g]) * KSSD_CHUNK;
        uint64_t cap = (uint64_t)((double)pos * rate * c->region_factor) + 256;
        if (cap > pos) cap = pos;  // a genome cannot emit more tuples than it has positions
        if (cap > DEDUP_MAX_N) {
            if ((uint64_t)((double)pos * rate * 1.25) + 64 > DEDUP_MAX_N) {
                c->last_launch_rc = KSSD_ERR_UNSUPPORTED;  // genome too large for the LDS dedup (next: global sort path)
                return KSSD_ERR_UNSUPPORTED;
            }
            cap = DEDUP_MAX_N;
        }
        c->h_reg_off[g] = acc;
        acc += cap;
        if (cap > max_cap) max_cap = cap;
    }
    c->h_reg_off[n_genomes] = acc;
    int rc;
    if ((rc = ensure(&c->d_chunk_gid, &c->cap_chunks, (size_t)n_chunks + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_chunk_off, &c->cap_chunk_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_reg_off, &c->cap_reg_off, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_cursor, &c->cap_cursor, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_kept, &c->cap_kept, (size_t)n_genomes + 1)) != KSSD_OK) return rc;
    if ((rc = ensure(&c->d_regions, &c->cap_regions, (size_t)acc + 1)) != KSSD_OK) return rc;

    HIPCK(hipMemcpyAsync(c->d_chunk_off, h_chunk_off, ((size_t)n_genomes + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(c->d_reg_off, c->h_reg_off.data(), ((size_t)n_genomes + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCK(hipMemsetAsync(c->d_cursor, 0, (size_t)n_genomes * 4, s));
    if (n_chunks) {
        hipLaunchKernelGGL(chunk_gid_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, s,
                           (const uint64_t *)c->d_chunk_off, n_genomes, n_chunks, c->d_chunk_gid);
        ScanArgs a;
        a.packed = d_packed; a.mask = d_mask; a.chunk_gid = c->d_chunk_gid; a.n_chunks = n_chunks;
        a.chunk_off = (const unsigned long long *)c->d_chunk_off;
        a.T1 = c->d_T1; a.G = c->d_G; a.reg_off = (const unsigned long long *)c->d_reg_off;
        a.cursor = c->d_cursor; a.regions = c->d_regions;
        uint64_t want = (n_chunks + SCAN_WAVES - 1) / SCAN_WAVES;
        int grid = (int)(want < (uint64_t)c->cu_count ? want : (uint64_t)c->cu_count);
        const unsigned evi = c->ev_n[0] % EV_RING;
        HIPCK(hipEventRecord(c->ev_a[0][evi], s));
        switch (c->P.subk) {
        case 2: rc = launch_scan<2>(c, a, grid, s); break;
        case 3: rc = launch_scan<3>(c, a, grid, s); break;
        case 4: rc = launch_scan<4>(c, a, grid, s); break;
        case 5: rc = launch_scan<5>(c, a, grid, s); break;
        case 6: {
            static const int abl = getenv("KSSD_DEV_ABLATE") ? atoi(getenv("KSSD_DEV_ABLATE")) : 0;  // profiling only
            if (abl == 1) rc = launch_scan<6, 1>(c, a, grid, s);
            else if (abl == 2) rc = launch_scan<6, 2>(c, a, grid, s);
            else if (abl == 3) rc = launch_scan<6, 3>(c, a, grid, s);
            else rc = launch_scan<6>(c, a, grid, s);
            break;
        }
        case 7: rc = launch_scan<7>(c, a, grid, s); break;
        default: rc = KSSD_ERR_UNSUPPORTED;
        }
        if (rc != KSSD_OK) return rc;
        HIPCK(hipEventRecord(c->ev_b[0][evi], s));
        c->ev_n[0]++;
    }
    uint32_t np = 1;
    while (np < max_cap) np <<= 1;
    const size_t dlds = (size_t)np * 4;
    HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_dedup_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dlds < 65536 ? 65536 : dlds)));
    hipLaunchKernelGGL(sketch_dedup_kernel, dim3(n_genomes), dim3(DEDUP_THREADS), dlds, s, c->P,
                       (const unsigned long long *)c->d_reg_off, (const uint32_t *)c->d_cursor, c->d_regions, c->d_kept,
                       flags, min_occ, c->d_status);
    hipLaunchKernelGGL(sketch_offsets_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t *)c->d_kept, n_genomes,
                       (unsigned long long *)d_out_off, (unsigned long long)out_cap, c->d_status);
    hipLaunchKernelGGL(sketch_gather_kernel, dim3(n_genomes), dim3(256), 0, s, (const unsigned long long *)c->d_reg_off,
                       (const uint32_t *)c->d_regions, (const uint32_t *)c->d_kept, (const unsigned long long *)d_out_off,
                       d_out_ids, (const SketchStatus *)c->d_status);
    HIPCK(hipGetLastError());
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
    if (st.region_overflow) {
        double need = (double)st.max_need_q8 / 256.0;  // emitted / capacity of the worst genome
        c->region_factor *= (need > 1.0 ? need : 1.0) * 1.25;
        return KSSD_ERR_OVERFLOW;
    }
    if (st.out_overflow) return KSSD_ERR_OVERFLOW;
    if (st.capacity_genome_p1) {
        if (bad_genome) *bad_genome = (int64_t)(0xFFFFFFFFu - st.capacity_genome_p1);
        return KSSD_ERR_CAPACITY;
    }
    return KSSD_OK;
}

extern "C" int kssd_gpu_sketch_batch(kssd_gpu_ctx *c, const uint32_t *packed, const uint32_t *mask,
                                     const uint64_t *chunk_off, uint32_t n_genomes, uint32_t flags, uint32_t min_occ,
                                     uint64_t **out_off, uint32_t **out_ids, int64_t *bad_genome)
{
    if (!c || !chunk_off || !out_off || !out_ids) return KSSD_ERR_PARAM;
    HIPCK(hipSetDevice(c->device));
    *out_off = nullptr;
    *out_ids = nullptr;
    const uint64_t n_chunks = chunk_off[n_genomes];
    const size_t pw = (size_t)n_chunks * KSSD_CHUNK_WORDS + KSSD_PACK_SLACK_WORDS;
    const size_t mw = (size_t)n_chunks * KSSD_CHUNK_MASKW + KSSD_PACK_SLACK_WORDS;
    uint32_t *d_p = nullptr, *d_m = nullptr, *d_ids = nullptr;
    uint64_t *d_off = nullptr;
    int rc = KSSD_OK;
    auto cleanup = [&]() {
        if (d_p) hipFree(d_p);
        if (d_m) hipFree(d_m);
        if (d_ids) hipFree(d_ids);
        if (d_off) hipFree(d_off);
    };
#define BCK(x) do { if ((x) != hipSuccess) { rc = hip_fail(hipGetLastError(), #x, __LINE__); cleanup(); return rc; } } while (0)
    BCK(hipMalloc(&d_p, pw * 4));
    BCK(hipMalloc(&d_m, mw * 4));
    BCK(hipMalloc(&d_off, ((size_t)n_genomes + 1) * 8));
    BCK(hipMemset(d_p, 0, pw * 4));
    BCK(hipMemset(d_m, 0, mw * 4));
    if (n_chunks) {
        BCK(hipMemcpy(d_p, packed, (size_t)n_chunks * KSSD_CHUNK_WORDS * 4, hipMemcpyHostToDevice));
        BCK(hipMemcpy(d_m, mask, (size_t)n_chunks * KSSD_CHUNK_MASKW * 4, hipMemcpyHostToDevice));
    }
    const double rate = (double)c->P.dim_end / (double)(1ull << (4 * c->P.subk));
    uint64_t out_cap = (uint64_t)((double)n_chunks * KSSD_CHUNK * rate * 1.5) + 1024;
    uint64_t total = 0;
    for (int attempt = 0; attempt < 12; attempt++) {
        if (d_ids) { hipFree(d_ids); d_ids = nullptr; }
        BCK(hipMalloc(&d_ids, (size_t)out_cap * 4));
        rc = kssd_gpu_sketch_device(c, d_p, d_m, chunk_off, n_genomes, flags, min_occ, d_off, d_ids, out_cap, nullptr);
        if (rc != KSSD_OK) break;
        rc = kssd_gpu_sketch_status(c, &total, bad_genome, nullptr);
        if (rc != KSSD_ERR_OVERFLOW) break;
        if (total > out_cap) out_cap = total + 1024;
    }
    if (rc == KSSD_OK) {
        uint64_t *h_off = (uint64_t *)malloc(((size_t)n_genomes + 1) * 8);
        uint32_t *h_ids = (uint32_t *)malloc((size_t)(total ? total : 1) * 4);
        if (!h_off || !h_ids) { free(h_off); free(h_ids); cleanup(); return KSSD_ERR_NOMEM; }
        BCK(hipMemcpy(h_off, d_off, ((size_t)n_genomes + 1) * 8, hipMemcpyDeviceToHost));
        if (total) BCK(hipMemcpy(h_ids, d_ids, (size_t)total * 4, hipMemcpyDeviceToHost));
        *out_off = h_off;
        *out_ids = h_ids;
    }
    cleanup();
    return rc;
#undef BCK
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
// distance path (index build + row kernel) lives in kssd_dist.inc
// ---------------------------------------------------------------------------------------------------
#include "kssd_dist.inc"
