// kssd_core.h -- arithmetic shared by the gfx950 kernels and the host-side table builders.
//
// Everything here is a restatement of what the reference computes per k-mer
// (iseq2comem.c:54-77 constants, :245-253 canonical strand / .shuf filter / reduced tuple), arranged
// for a two-stage GPU scan:
//   stage 1  a superset filter evaluated for 4 window positions per LDS lookup ("quad core" table)
//   stage 2  the exact evaluation of the few surviving candidates
// Functions marked KSSD_HD compile for host and device so that tests can drive the same bit
// manipulation on the CPU (tests/emu) -- the product only ever runs them on the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KSSD_HD __host__ __device__ __forceinline__
#else
#define KSSD_HD static inline
#endif

#define KSSD_CHUNK 4096      // positions per chunk (one wave-iteration: 64 lanes x 64 positions)
#define KSSD_T1_BITS 18      // log2(entries) of the quad-core nibble table
#define KSSD_T1_BYTES (1u << (KSSD_T1_BITS - 1))
#define KSSD_MIN_DIM_SMP 4096  // MIN_SUBCTX_DIM_SMP_SZ, command_shuffle.h:29
#define KSSD_COMPONENT_SZ 7    // reference Makefile:4
#define KSSD_CTX_SPC_USE_L 8   // global_basic.h:45-47
#define KSSD_EMPTY_KEY 0xFFFFFFFFu

struct KssdParams {
    int32_t k, subk, drlevel;
    int32_t out;       // k - subk: bases on each side of the sub-context      iseq2comem.c:59
    int32_t nb;        // 2k bases per k-mer ("TL")                             iseq2comem.c:68
    int32_t comp_bits; // iseq2comem.c:527
    uint32_t comp_num; // iseq2comem.c:63-64
    uint32_t dim_end;  // accepted permutation ranks are [0, dim_end)          iseq2comem.c:74-76
    uint32_t hashsize, hashlimit;  // command_dist.c:217-236, iseq2comem.c:61
    uint32_t g_log2;   // exact table G has 2^g_log2 slots
    uint64_t dim_mask; // 4*subk ones
};

// hash-table sizes of the reference, only used for the capacity rule (global_basic.c:74-81)
static const uint32_t kssd_primes[25] = {
    251u, 509u, 1021u, 2039u, 4093u, 8191u, 16381u, 32749u, 65521u, 131071u, 262139u, 524287u,
    1048573u, 2097143u, 4194301u, 8388593u, 16777213u, 33554393u, 67108859u, 134217689u,
    268435399u, 536870909u, 1073741789u, 2147483647u, 4294967291u};

// returns 0 on success, -1 if the reference would reject the parameters,
// -2 if the reference accepts them but the device path does not (yet)
static inline int kssd_params_init(KssdParams *p, int k, int subk, int drlevel)
{
    if (k < subk || subk >= 8 || subk < 1 || drlevel < 0 || k > 15) return -1;  // command_shuffle.c:163-168
    int pidx = 4 * (k - drlevel) - KSSD_CTX_SPC_USE_L - 7;                        // command_dist.c:220
    if (pidx < 0 || pidx > 24) return -1;
    if (subk < 2) return -2;                       // quad-core filter needs >= 4 bases of sub-context
    if (4 * (k - drlevel) > 32) return -2;         // reduced tuple must fit the u32 the formats store
    if (subk < drlevel) return -1;
    p->k = k; p->subk = subk; p->drlevel = drlevel;
    p->out = k - subk;
    p->nb = 2 * k;
    int extra = k - drlevel - KSSD_COMPONENT_SZ;
    p->comp_bits = extra > 0 ? 4 * extra : 0;
    p->comp_num = extra > 0 ? (1u << p->comp_bits) : 1u;
    uint64_t sub = 1ull << (4 * (subk - drlevel));
    p->dim_end = (uint32_t)(sub > KSSD_MIN_DIM_SMP ? sub : KSSD_MIN_DIM_SMP);
    if ((uint64_t)p->dim_end > (1ull << (4 * subk))) return -1;  // more ranks than sub-contexts
    if (p->dim_end > (1u << 16)) return -2;        // device tables sized for <= 65536 accepted ranks
    p->hashsize = kssd_primes[pidx];
    p->hashlimit = (uint32_t)(p->hashsize * 0.6);  // LD_FCTR, global_basic.h:49
    uint32_t lg = 0;
    while ((1u << lg) < 4u * p->dim_end) lg++;
    p->g_log2 = lg;
    p->dim_mask = (1ull << (4 * subk)) - 1;
    return 0;
}

// reverse complement of the low `nbases` bases of x (2 bits per base, A=0 C=1 G=2 T=3)
KSSD_HD uint64_t kssd_revcomp(uint64_t x, int nbases)
{
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * nbases);
}

// index of a quad core in the nibble table: identity while it fits, multiplicative hash beyond
template <int CORE_BITS>
KSSD_HD uint32_t kssd_t1_index(uint32_t core)
{
    if (CORE_BITS <= KSSD_T1_BITS) return core;
    return (core * 0x9E3779B1u) >> (32 - KSSD_T1_BITS);
}

KSSD_HD uint32_t kssd_t1_index_rt(uint32_t core, int core_bits)
{
    if (core_bits <= KSSD_T1_BITS) return core;
    return (core * 0x9E3779B1u) >> (32 - KSSD_T1_BITS);
}

KSSD_HD uint32_t kssd_g_slot(uint32_t dim, uint32_t g_log2) { return (dim * 0x9E3779B1u) >> (32 - g_log2); }

// Stage 1 for one lane: W[0..4] = the 5 packed words covering the lane's 64 window-start positions
// plus the 2*subk-1 bases after them.  Bit i of the result <=> the sub-context that starts at the
// lane's position i MAY be one of the accepted sub-contexts or the reverse complement of one.
// One table read answers 4 consecutive positions: they share the 2*subk-3 bases in the middle ("quad
// core").  Every position is tested under two quad alignments (quads starting at 4q and at 4q+2): the
// two cores together pin 2*subk-1 of its 2*subk bases, which cuts the candidates ~10x (3.1 % -> 0.3 %
// at subk = 6) for the price of a second LDS read per 4 positions.
// One quad lookup.  The core is pulled out of the 64-bit window with 2 spare bits below it, so that
//   byte address = field >> 3, nibble select = field & 4   (identity index, CB <= 18)
// costs one shift/alignbit, one bfe, one and, one bfe after the LDS read.
template <int CB, typename T1PTR>
KSSD_HD uint32_t kssd_quad_nibble(uint32_t whi, uint32_t wlo, int bo, T1PTR T1)
{
    const uint64_t win = ((uint64_t)whi << 32) | wlo;
    if (CB <= KSSD_T1_BITS) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int sh = 62 - bo - CB;  // compile-time after unrolling
        const uint32_t field = sh >= 32 ? (whi >> (sh - 32)) : __builtin_amdgcn_alignbit(whi, wlo, sh);
        const uint32_t byte = T1[__builtin_amdgcn_ubfe(field, 3, CB - 1)];
        return __builtin_amdgcn_ubfe(byte, field & 4u, 4);
#else
        const uint32_t field = (uint32_t)(win >> (62 - bo - CB));  // core in bits [2, 2+CB)
        const uint32_t byte = T1[(field >> 3) & ((1u << (CB - 1)) - 1u)];
        return (byte >> (field & 4u)) & 0xFu;
#endif
    } else {
        const uint32_t core = (uint32_t)(win >> (64 - bo - CB)) & ((1u << CB) - 1u);
        const uint32_t idx = kssd_t1_index<CB>(core);
        return ((uint32_t)T1[idx >> 1] >> ((idx & 1u) * 4u)) & 0xFu;
    }
}

template <int SUBK, typename T1PTR>
KSSD_HD void kssd_stage1(const uint32_t (&W)[5], T1PTR T1, uint32_t &cand_lo, uint32_t &cand_hi)
{
    constexpr int CB = 2 * (2 * SUBK - 3);  // bits of a quad core
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int o = 8 * q + 6;  // bit offset of the core of quad q, counted from the top of W[0]
        const uint32_t nib = kssd_quad_nibble<CB>(W[o >> 5], W[(o >> 5) + 1], o & 31, T1);
        if (q < 8) lo |= nib << (4 * q);
        else hi |= nib << (4 * (q - 8));
    }
    uint32_t lo2 = 0, hi2 = 0;
#pragma unroll
    for (int q = -1; q < 16; q++) {
        const int o = 8 * q + 10;  // quads shifted by two positions: windows 4q+2 .. 4q+5
        const int wi = o >> 5;
        const uint32_t nib = kssd_quad_nibble<CB>(W[wi], wi < 4 ? W[wi < 4 ? wi + 1 : 4] : 0u, o & 31, T1);
        if (q < 0) lo2 |= nib >> 2;  // windows -2,-1 belong to the previous lane
        else if (q < 7) lo2 |= nib << (4 * q + 2);
        else if (q == 7) { lo2 |= nib << 30; hi2 |= nib >> 2; }  // windows 30..33 straddle the two words
        else hi2 |= nib << (4 * q - 30);  // q = 15: windows 64,65 fall off the top
    }
    cand_lo = lo & lo2;
    cand_hi = hi & hi2;
}

struct KssdG {  // one slot of the exact table: accepted sub-context -> permutation rank
    uint32_t key;  // sub-context, KSSD_EMPTY_KEY when free
    uint32_t rank;
};

// Stage 2: exact evaluation of the k-mer whose sub-context starts at global position s.
// [lo_ok, hi_ok) = positions the k-mer may touch (same genome, inside the batch).
// Returns true and the reduced tuple when the reference would insert it (iseq2comem.c:243-253).
KSSD_HD bool kssd_stage2(const KssdParams &P, int64_t s, int64_t lo_ok, int64_t hi_ok,
                         const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask,
                         const KssdG *__restrict__ G, uint32_t &dr_out)
{
    const int64_t b0 = s - P.out;  // first base of the k-mer
    if (b0 < lo_ok || b0 + P.nb > hi_ok) return false;
    // validity: all nb mask bits set  (run counter "base > TL", iseq2comem.c:243)
    const uint64_t mw = (uint64_t)b0 >> 5;
    const int mo = (int)(b0 & 31);
    uint64_t m64 = (uint64_t)mask[mw] | ((uint64_t)mask[mw + 1] << 32);
    const uint64_t need = (1ull << P.nb) - 1ull;
    if (((m64 >> mo) & need) != need) return false;
    // the 2*nb bits of the forward k-mer out of three packed words
    const uint64_t pw = (uint64_t)b0 >> 4;
    const int sh = (int)(b0 & 15) * 2;
    uint64_t hi64 = ((uint64_t)packed[pw] << 32) | packed[pw + 1];
    uint64_t w2 = packed[pw + 2];
    uint64_t top = sh ? ((hi64 << sh) | (w2 >> (32 - sh))) : hi64;
    uint64_t fwd = top >> (64 - 2 * P.nb);
    uint64_t rev = kssd_revcomp(fwd, P.nb);
    uint64_t u = fwd < rev ? fwd : rev;                       // iseq2comem.c:245
    uint32_t dim = (uint32_t)((u >> (2 * P.out)) & P.dim_mask);  // :246
    // exact membership + rank (the reference reads the 16^subk-entry permutation here, :247-249)
    uint32_t slot = kssd_g_slot(dim, P.g_log2);
    const uint32_t gmask = (1u << P.g_log2) - 1u;
    uint32_t rank;
    for (;;) {
        KssdG e = G[slot];
        if (e.key == dim) { rank = e.rank; break; }
        if (e.key == KSSD_EMPTY_KEY) return false;
        slot = (slot + 1) & gmask;
    }
    // reduced tuple (:250-253): outer bases packed above the rank, literally as the reference adds them
    uint64_t upper = u & (((1ull << (2 * P.out)) - 1ull) << (2 * (P.k + P.subk)));
    uint64_t lower = u & ((1ull << (2 * P.out)) - 1ull);
    uint64_t dr = ((upper + (lower << (2 * P.nb - 4 * P.out))) >> (4 * P.drlevel)) + rank;
    dr_out = (uint32_t)dr;
    return true;
}

#include <vector>
// Host-side construction of the two device tables from the accepted sub-contexts
// (accepted[r] = the sub-context whose permutation rank is r, r < dim_end).
//   T1  stage-1 pattern set: accepted sub-contexts and their reverse complements, every one entered
//       under the 4 alignments a quad can see it in; nibble bit j <=> window = quad position j
//   G   stage-2 exact map sub-context -> rank (open addressing, linear probing)
static inline void kssd_build_tables(const KssdParams &P, const std::vector<uint32_t> &accepted,
                                     std::vector<uint8_t> &T1, std::vector<KssdG> &G)
{
    const int Lp = 2 * P.subk, Lc = Lp - 3, CB = 2 * Lc;
    T1.assign(KSSD_T1_BYTES, 0);
    auto add = [&](uint64_t x) {
        for (int j = 0; j < 4; j++) {
            uint32_t core = (uint32_t)((x >> (2 * j)) & ((1ull << CB) - 1ull));  // bases [3-j, 3-j+Lc) of x
            uint32_t idx = kssd_t1_index_rt(core, CB);
            T1[idx >> 1] |= (uint8_t)((1u << j) << ((idx & 1u) * 4u));
        }
    };
    for (size_t r = 0; r < accepted.size(); r++) {
        add(accepted[r]);
        add(kssd_revcomp(accepted[r], Lp));
    }
    const size_t gn = (size_t)1 << P.g_log2;
    G.assign(gn, KssdG{KSSD_EMPTY_KEY, 0});
    for (size_t r = 0; r < accepted.size(); r++) {
        uint32_t slot = kssd_g_slot(accepted[r], P.g_log2);
        while (G[slot].key != KSSD_EMPTY_KEY) slot = (slot + 1) & (uint32_t)(gn - 1);
        G[slot] = KssdG{accepted[r], (uint32_t)r};
    }
}

// accepted[] from a full .shuf permutation (iseq2comem.c:247-249); false if it is not a permutation
static inline bool kssd_accepted_from_table(const KssdParams &P, const int32_t *table, std::vector<uint32_t> &accepted)
{
    accepted.assign(P.dim_end, KSSD_EMPTY_KEY);
    const uint64_t n = 1ull << (4 * P.subk);
    uint32_t found = 0;
    for (uint64_t x = 0; x < n; x++) {
        const int32_t r = table[x];
        if (r >= 0 && (uint32_t)r < P.dim_end) {
            if (accepted[r] != KSSD_EMPTY_KEY) return false;
            accepted[r] = (uint32_t)x;
            found++;
        }
    }
    return found == P.dim_end;
}
