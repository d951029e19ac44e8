// kssd_core.h -- arithmetic shared by the gfx950 kernels and the host-side table builders.
//
// Everything here is a restatement of what the reference computes per k-mer
// (iseq2comem.c:54-77 constants, :245-253 canonical strand / .shuf filter / reduced tuple), arranged
// for a two-stage GPU scan:
//   stage 1  a superset filter evaluated for KSSD_GW window positions per LDS lookup (group-filter table)
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
#define KSSD_T1_BYTES (1u << 17)  // stage-1 group-filter table: one byte per 17-bit index
#ifndef KSSD_GW
#define KSSD_GW 5                // windows per table read (KssdGrp)
#endif
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
    uint32_t g_log2;   // exact table G: 2^g_log2 buckets of two slots
    uint32_t g_mul[2]; // their multiplicative hashes (chosen by kssd_build_tables)
    uint64_t dim_mask; // 4*subk ones
    // k - drlevel = 9: the reduced tuple has 36 bits (the reference splits it over 256 component files of 28-bit ids,
    // iseq2comem.c:63-64,527,542-543).  The device works in 2^pass_bits = 16 passes over the candidates of ONE scan: pass s
    // keeps the tuples whose low four bits are s and calls tuple >> 4 their id -- 32 bits, like the sixteen components of
    // k - drlevel = 8.  Component file of an id of pass s: ((id & 15) << 4) | s, stored id: id >> 4.
    uint32_t pass_bits, pass;
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
    if (subk < 2) return -2;                       // group filter needs >= 4 bases of sub-context
    if (4 * (k - drlevel) > 36) return -2;         // (never: the table sizes above end at k - drlevel = 9)
    p->pass_bits = 4 * (k - drlevel) > 32 ? (uint32_t)(4 * (k - drlevel) - 32) : 0u;  // ids are 32 bits: the rest names the pass
    p->pass = 0;
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
    while ((1u << lg) < 2u * p->dim_end) lg++;
    p->g_log2 = lg;
    p->g_mul[0] = 0x9E3779B1u;
    p->g_mul[1] = 0x85EBCA6Bu;
    p->dim_mask = (1ull << (4 * subk)) - 1;
    return 0;
}

// reverse complement of the low `nbases` bases of x (2 bits per base, A=0 C=1 G=2 T=3)
KSSD_HD uint64_t kssd_revcomp(uint64_t x, int nbases)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_bfrev_b32 x2 reverses the base order and swaps the two bits of every base; swap those back
    uint64_t r = __builtin_bitreverse64(x);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return (~r) >> (64 - 2 * nbases);
#endif
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * nbases);
}

// ---------------------------------------------------------------------------------------------------
// Stage 1, group filter: a superset test evaluated for GW window positions per LDS read.
// The canonical k-mer's sub-context is either the forward middle 2*subk bases m or their reverse complement,
// so a position can only be sampled if m is in S = A u rc(A) (A = the dim_end accepted sub-contexts).
// GW consecutive windows share the C = 2*subk + 1 - GW bases in the middle (the "group core").  The table is
// indexed by the first IDXB stream bits from the core on; entry bit j <=> "some pattern of S, placed at window
// j of the group, agrees with these bits" (index bits beyond a window's end are wildcards for that window).
// Byte entries: no sub-byte select, answers merge with one shift-or.  Every position is tested under two
// alignments (A: groups at GW*q, B: groups at GW*q + SB - GW); 0.8 % of the positions survive at L3K10
// (true rate 0.049 %).  The two alignments are separate issue / merge halves so that the kernel can keep one
// batch of LDS reads in flight while it works on the other (s_waitcnt lgkmcnt can only count to 15).
// ---------------------------------------------------------------------------------------------------
template <int SUBK, int GW>
struct KssdGrp {
    static constexpr int LP = 2 * SUBK;                       // bases of a sub-context
    static constexpr int W = (LP >= GW + 1) ? GW : (LP - 1);  // windows per group (>= 1 shared base)
    static constexpr int C = LP + 1 - W;                      // bases shared by the windows of a group
    static constexpr int IDXB = (2 * LP < 17) ? 2 * LP : 17;  // index bits (128 KiB of byte entries at most)
    static constexpr int SB = W / 2;                          // offset of the second alignment
    static constexpr int NA = (63 / W) + 1;                   // groups of alignment A: Q = W*q
    static constexpr int QB0 = SB - W;                        // first group of alignment B
    static constexpr int NB = (63 - QB0) / W + 1;             // groups of alignment B: Q = QB0 + W*q
    static constexpr int NMAX = NA > NB ? NA : NB;
    static constexpr int count(int aln) { return aln ? NB : NA; }
    static constexpr int first(int aln) { return aln ? QB0 : 0; }
};

// index field of the group that starts at lane position Q (compile-time): IDXB stream bits from the core on
template <int IDXB>
KSSD_HD uint32_t kssd_grp_field(const uint32_t (&Wd)[5], const uint32_t (&V)[4], int start)
{
    const int i = start >> 5, off = start & 31;
#if defined(__HIP_DEVICE_COMPILE__)
    if (off + IDXB <= 32) return __builtin_amdgcn_ubfe(Wd[i], 32 - off - IDXB, IDXB);
    return __builtin_amdgcn_ubfe(V[i < 4 ? i : 3], 48 - off - IDXB, IDXB);  // V[i] = stream bits [32i+16, 32i+48)
#else
    (void)V;
    const uint64_t win = ((uint64_t)Wd[i] << 32) | (i < 4 ? Wd[i + 1] : 0u);
    return (uint32_t)(win >> (64 - off - IDXB)) & ((1u << IDXB) - 1u);
#endif
}

// issue the table reads of one alignment (ALN 0 = A, 1 = B); raw[q] = answer bits of group q
template <int SUBK, int GW, int ALN, typename T1PTR>
KSSD_HD void kssd_grp_issue(const uint32_t (&Wd)[5], T1PTR T1, uint32_t (&raw)[KssdGrp<SUBK, GW>::NMAX])
{
    typedef KssdGrp<SUBK, GW> Gp;
    uint32_t V[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
        V[i] = __builtin_amdgcn_alignbit(Wd[i], Wd[i + 1], 16);
#else
        V[i] = (Wd[i] << 16) | (Wd[i + 1] >> 16);
#endif
    }
#pragma unroll
    for (int q = 0; q < Gp::NMAX; q++) {
        if (q < Gp::count(ALN)) {
            const int Q = Gp::first(ALN) + Gp::W * q;
            raw[q] = T1[kssd_grp_field<Gp::IDXB>(Wd, V, 2 * (Q + Gp::W - 1))];
        } else {
            raw[q] = 0;
        }
    }
}

// merge the answers of one alignment into the 64-position masks (bit p <=> lane position p may be sampled)
#if defined(__HIP_DEVICE_COMPILE__)
// One v_alignbit_b32 per answer and mask word: the groups of a word are taken in ascending order and each one is pushed in
// at the TOP -- acc = ({n, acc} >> k)[31:0] puts the low k bits of n into bits 32-k .. 31 and moves what is there down.
// The k of a word's groups add up to 32, so the first one ends at bit 0.  A group that starts below the word (alignment B's
// first group starts at position -3, alignment A's group 6 straddles positions 30 .. 34) is pushed in whole: its bits
// that belong below the word fall off the bottom on the way.  A group that ends above the word gives its low bits only.
// Why not (n << Q) | acc: a funnel shift reads only the low k <= 5 bits of n, and the compiler knows it -- the table byte
// needs no zero extension.  With v_lshl_or_b32 it put a v_and_b32 0xff in front of every answer of alignment A (the
// extension had been moved away from the ds_read_u8 that does it for free), 13 of ~200 instructions per 64 positions.
template <int W, int Q, int W0>
__device__ __forceinline__ void kssd_grp_push(uint32_t n, uint32_t &acc)
{
    constexpr int p0 = Q > W0 ? Q : W0, p1 = (Q + W < W0 + 32) ? Q + W : W0 + 32, c = p1 - p0;
    if (c > 0) {
        constexpr int k = (Q + W > W0 + 32) ? c : W;
        acc = __builtin_amdgcn_alignbit(n, acc, k > 0 ? k : 1);
    }
}
template <int SUBK, int GW, int ALN, int q>
struct KssdMergeLoop {
    static __device__ __forceinline__ void run(const uint32_t (&raw)[KssdGrp<SUBK, GW>::NMAX], uint32_t &lo, uint32_t &hi)
    {
        typedef KssdGrp<SUBK, GW> Gp;
        if (q < Gp::count(ALN)) {
            constexpr int Q = Gp::first(ALN) + Gp::W * q;
            kssd_grp_push<Gp::W, Q, 0>(raw[q < Gp::NMAX ? q : 0], lo);
            kssd_grp_push<Gp::W, Q, 32>(raw[q < Gp::NMAX ? q : 0], hi);
            KssdMergeLoop<SUBK, GW, ALN, (q + 1 < Gp::NMAX ? q + 1 : -1)>::run(raw, lo, hi);
        }
    }
};
template <int SUBK, int GW, int ALN>
struct KssdMergeLoop<SUBK, GW, ALN, -1> {
    static __device__ __forceinline__ void run(const uint32_t (&)[KssdGrp<SUBK, GW>::NMAX], uint32_t &, uint32_t &) {}
};
#endif

template <int SUBK, int GW, int ALN>
KSSD_HD void kssd_grp_merge(const uint32_t (&raw)[KssdGrp<SUBK, GW>::NMAX], uint32_t &lo, uint32_t &hi)
{
    typedef KssdGrp<SUBK, GW> Gp;
    lo = 0;
    hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    KssdMergeLoop<SUBK, GW, ALN, 0>::run(raw, lo, hi);
#else
#pragma unroll
    for (int q = 0; q < Gp::count(ALN); q++) {
        int Q = Gp::first(ALN) + Gp::W * q;
        uint32_t n = raw[q];
        if (Q < 0) { n >>= -Q; Q = 0; }
        if (Q < 32) {
            lo |= n << Q;
            if (Q + Gp::W > 32) hi |= n >> (32 - Q);
        } else {
            hi |= n << (Q - 32);  // answers for positions >= 64 fall off the top
        }
    }
#endif
}

// both alignments at once (CPU emulation in tests/emu; the kernel interleaves the halves)
template <int SUBK, int GW, typename T1PTR>
KSSD_HD void kssd_stage1g(const uint32_t (&Wd)[5], T1PTR T1, uint32_t &cand_lo, uint32_t &cand_hi)
{
    uint32_t ra[KssdGrp<SUBK, GW>::NMAX], rb[KssdGrp<SUBK, GW>::NMAX];
    uint32_t alo, ahi, blo, bhi;
    kssd_grp_issue<SUBK, GW, 0>(Wd, T1, ra);
    kssd_grp_issue<SUBK, GW, 1>(Wd, T1, rb);
    kssd_grp_merge<SUBK, GW, 0>(ra, alo, ahi);
    kssd_grp_merge<SUBK, GW, 1>(rb, blo, bhi);
    cand_lo = alo & blo;
    cand_hi = ahi & bhi;
}

// ---------------------------------------------------------------------------------------------------
// Stage 1.5: exact-pattern Bloom test of the few candidates stage 1 lets through.  Stage 1 only proves that
// each alignment agrees with SOME pattern of S; 94 % of its candidates are not in S at all.  Candidates are buffered as
// positions; the lane that takes one into a dense round of 64 fetches the packed words around it (kssd_carry_from_words),
// and the 2*subk-base pattern m on top of them is looked up in a blocked Bloom filter of S in LDS: one 32-bit word per
// lookup, 3 bits per pattern, 16 bits of filter per pattern -> ~1.6 % false positives.  What survives (~0.06 % of the
// positions, half of them true samples) is all that ever reaches the exact stage 2 and its global-memory reads.
// ---------------------------------------------------------------------------------------------------
#define KSSD_BLOOM_WORDS 4096u
KSSD_HD uint32_t kssd_bloom_hash(uint32_t m) { return m * 0x9E3779B1u; }
KSSD_HD uint32_t kssd_bloom_word(uint32_t h) { return h >> 20; }
KSSD_HD uint32_t kssd_bloom_bits(uint32_t h)
{
    return (1u << ((h >> 15) & 31u)) | (1u << ((h >> 10) & 31u)) | (1u << ((h >> 5) & 31u));
}

// the 2*SUBK bases that start at lane position b (0..63), out of the lane's 5 packed words
template <int SUBK>
KSSD_HD uint32_t kssd_extract_m(const uint32_t (&W)[5], uint32_t b)
{
    // bit-select on all-ones / all-zeros masks (v_bfi_b32): written as a select or an indexed array the
    // compiler turns W into a scratch-memory array, and every lookup into a memory round trip
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t s1 = (uint32_t)__builtin_amdgcn_sbfe((int)b, 4, 1), s2 = (uint32_t)__builtin_amdgcn_sbfe((int)b, 5, 1);
#else
    const uint32_t s1 = 0u - ((b >> 4) & 1u), s2 = 0u - ((b >> 5) & 1u);
#endif
#define KSSD_BSEL(m, x, y) (((m) & (x)) | (~(m) & (y)))  /* m ? x : y, bitwise */
    const uint32_t hi = KSSD_BSEL(s2, KSSD_BSEL(s1, W[3], W[2]), KSSD_BSEL(s1, W[1], W[0]));
    const uint32_t lo = KSSD_BSEL(s2, KSSD_BSEL(s1, W[4], W[3]), KSSD_BSEL(s1, W[2], W[1]));
#undef KSSD_BSEL
    const uint32_t sh = (b & 15u) * 2u;
    const uint32_t top = (uint32_t)(((((uint64_t)hi << 32) | lo) << sh) >> 32);
    return top >> (32 - 4 * SUBK);
}

// What goes to the exact stage together with a candidate's position, so that stage 2 does not have to read the packed
// stream again (one 128-byte HBM line per candidate for 5 useful bytes): the 16 bases from the sub-context's first base on
// (top32: the word the Bloom pattern is the top of) and the 4 bases in front of it (front, 8 bits).  The lane that takes a
// buffered candidate into a Bloom round fetches three packed words around its position p -- wm = the word in front of the
// one that holds p (0 for the batch's first word), wa = that word, wb = the next -- and cuts both out with two 64-bit shifts
// (r = p & 15).  20 bases [p-4, p+16) hold the whole 2k-mer whenever out = k - subk <= 4 and 2 subk + out <= 16
// (kssd_carry_ok: L3K10, L2K8, K9 ...); other parameter sets ignore the payload and read the stream.
KSSD_HD void kssd_carry_from_words(uint32_t wm, uint32_t wa, uint32_t wb, uint32_t r, uint32_t &top32, uint32_t &front)
{
    const uint32_t sh = 2u * r;
    top32 = (uint32_t)(((((uint64_t)wa << 32) | wb) << sh) >> 32);
    front = (uint32_t)(((((uint64_t)wm << 32) | wa) << sh) >> 32) & 0xFFu;
}

// A bit per hash value of the accepted sub-contexts, in front of the exact table (per-genome kernel, FUSED): half of the scan's
// candidates are in S by their reverse complement only and fail the exact lookup of their canonical sub-context -- 65 536 bits
// (8 KiB of LDS the bucket sort's counters occupy later) answer "not accepted" for 94 % of those without the 16-byte read.
#define KSSD_GFILT_WORDS 2048u
KSSD_HD uint32_t kssd_gfilt_bit(uint32_t dim) { return (dim * 0x9E3779B1u) >> 16; }

struct KssdG {  // one slot of the exact table: accepted sub-context -> permutation rank
    uint32_t key;  // sub-context, KSSD_EMPTY_KEY when free
    uint32_t rank;
};

// The exact table: 2^g_log2 buckets of TWO slots (16 bytes, one read).  An accepted sub-context sits in the bucket g_mul[0]
// sends it to -- at a quarter load 98.6 % of the buckets hold all of theirs -- or, when that bucket was full, in the bucket
// g_mul[1] names; bit 31 of a bucket's first rank word says "some key of this bucket went elsewhere", so that a lookup reads
// the second bucket only then (1.4 % of the lookups).  (Round 2 until here: two-choice cuckoo with one-slot buckets,
// two 8-byte reads per lookup in two different lines -- the exact stage is bound by the number of random reads the L2
// takes per second, ~85 G/s: 8.6 M candidates of a read set took 210 us, the 2.9 M of the default batch 71.)
#define KSSD_G_MOVED 0x80000000u
KSSD_HD uint32_t kssd_g_slot(uint32_t dim, uint32_t mul, uint32_t g_log2) { return (dim * mul) >> (32 - g_log2); }
// (16-byte aligned: ONE 16-byte load per lookup -- without the attribute the compiler read a bucket as four separate words, and a CU
// pays for every vector-memory instruction of a wave whatever it fetches: 12 of the per-genome kernel's ~20 loads per candidate round)
struct __attribute__((aligned(16))) KssdGBucket {  // = KssdG[2]
    uint32_t key0, rank0, key1, rank1;
};
// the bucket's verdict: 1 = found (rank set), 0 = not in the table, 2 = look into the second bucket
// (written without branches: every word of the bucket is used on every path, so the bucket stays ONE 16-byte load -- with early
// returns the compiler fetched the words one by one, under the branches, as four vector-memory instructions)
KSSD_HD int kssd_g_match(const KssdGBucket &b, uint32_t dim, uint32_t &rank)
{
    const bool m0 = b.key0 == dim, m1 = b.key1 == dim;
    rank = m0 ? (b.rank0 & ~KSSD_G_MOVED) : b.rank1;  // (meaningful only when 1 is returned)
    return (m0 | m1) ? 1 : ((b.rank0 & KSSD_G_MOVED) ? 2 : 0);
}
KSSD_HD bool kssd_g_find(const KssdParams &P, const KssdG *__restrict__ G, uint32_t dim, uint32_t &rank)
{
    const KssdGBucket *B = reinterpret_cast<const KssdGBucket *>(G);
    int r = kssd_g_match(B[kssd_g_slot(dim, P.g_mul[0], P.g_log2)], dim, rank);
    if (r == 2) r = kssd_g_match(B[kssd_g_slot(dim, P.g_mul[1], P.g_log2)], dim, rank) == 1 ? 1 : 0;
    return r == 1;
}

// Stage 2, arithmetic only.  fwd = the 2k bases of the k-mer (first base in the highest of the 4k bits);
// u = canonical k-mer (iseq2comem.c:245), dim = its sub-context (:246).
KSSD_HD void kssd_s2_canon(const KssdParams &P, uint64_t fwd, uint64_t &u, uint32_t &dim)
{
    const uint64_t rev = kssd_revcomp(fwd, P.nb);
    u = fwd < rev ? fwd : rev;
    dim = (uint32_t)((u >> (2 * P.out)) & P.dim_mask);
}

// p0..p2 = packed words (b0>>4)+0..2, m0,m1 = mask words (b0>>5)+0..1, b0 = first base of the k-mer.
// Returns true if all 2k bases are valid (run counter "base > TL", iseq2comem.c:243).
KSSD_HD bool kssd_s2_decode(const KssdParams &P, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t m0, uint32_t m1,
                            uint32_t b0_low, uint64_t &u, uint32_t &dim)
{
    const int mo = (int)(b0_low & 31u);
    const uint64_t m64 = (uint64_t)m0 | ((uint64_t)m1 << 32);
    const uint64_t need = (1ull << P.nb) - 1ull;
    const bool valid = ((m64 >> mo) & need) == need;
    const int sh = (int)(b0_low & 15u) * 2;
    const uint64_t hi64 = ((uint64_t)p0 << 32) | p1;
    const uint64_t top = sh ? ((hi64 << sh) | ((uint64_t)p2 >> (32 - sh))) : hi64;
    kssd_s2_canon(P, top >> (64 - 2 * P.nb), u, dim);
    return valid;
}

// the carried payload (kssd_carry_from_words: 20 bases from 4 in front of the sub-context on) holds the whole k-mer
KSSD_HD bool kssd_carry_ok(const KssdParams &P) { return P.out <= 4 && 2 * P.subk + P.out <= 16; }
KSSD_HD uint64_t kssd_carry_payload(uint32_t top32, uint32_t front) { return ((uint64_t)(front & 0xFFu) << 32) | top32; }
// the 2k bases of the k-mer out of the payload: they start 4 - out bases into its 20
KSSD_HD uint64_t kssd_carry_fwd(const KssdParams &P, uint64_t payload40)
{
    return (payload40 >> (2 * (16 + P.out - P.nb))) & ((1ull << (2 * P.nb)) - 1ull);
}

// reduced tuple (iseq2comem.c:250-253): outer bases packed above the rank, literally as the reference adds them
KSSD_HD uint64_t kssd_s2_tuple64(const KssdParams &P, uint64_t u, uint32_t rank)
{
    const uint64_t upper = u & (((1ull << (2 * P.out)) - 1ull) << (2 * (P.k + P.subk)));
    const uint64_t lower = u & ((1ull << (2 * P.out)) - 1ull);
    return ((upper + (lower << (2 * P.nb - 4 * P.out))) >> (4 * P.drlevel)) + rank;
}
// the id the sketch stores (the whole tuple up to 32 bits; tuple >> pass_bits beyond), and whether the tuple belongs to
// the pass the parameters name
KSSD_HD uint32_t kssd_s2_tuple(const KssdParams &P, uint64_t u, uint32_t rank, bool &mine)
{
    const uint64_t dr = kssd_s2_tuple64(P, u, rank);
    mine = (dr & ((1ull << P.pass_bits) - 1ull)) == P.pass;
    return (uint32_t)(dr >> P.pass_bits);
}
KSSD_HD uint32_t kssd_s2_tuple(const KssdParams &P, uint64_t u, uint32_t rank)
{
    bool mine;
    return kssd_s2_tuple(P, u, rank, mine);
}

// Stage 2 in one piece: exact evaluation of the k-mer whose sub-context starts at global position s.
// [lo_ok, hi_ok) = positions the k-mer may touch (same genome, inside the batch).
// Returns true and the reduced tuple when the reference would insert it (iseq2comem.c:243-253).
KSSD_HD bool kssd_stage2(const KssdParams &P, int64_t s, int64_t lo_ok, int64_t hi_ok,
                         const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask,
                         const KssdG *__restrict__ G, uint32_t &dr_out)
{
    const int64_t b0 = s - P.out;  // first base of the k-mer
    if (b0 < lo_ok || b0 + P.nb > hi_ok) return false;
    const uint64_t mw = (uint64_t)b0 >> 5, pw = (uint64_t)b0 >> 4;
    uint64_t u;
    uint32_t dim;
    if (!kssd_s2_decode(P, packed[pw], packed[pw + 1], packed[pw + 2], mask[mw], mask[mw + 1], (uint32_t)b0, u, dim)) return false;
    // exact membership + rank (the reference reads the 16^subk-entry permutation here, :247-249)
    uint32_t rank;
    if (!kssd_g_find(P, G, dim, rank)) return false;
    dr_out = kssd_s2_tuple(P, u, rank);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// log(x) / k for the distance epilogue (MashD = log(1/(2J) + 0.5) / 2k, AafD = log(1/C) / 2k, command_dist.c:1251,1265).
// The host computes fl(fl(log x) / k) with glibc's log (error < 0.52 ulp).  A device log that is merely "within 1 ulp"
// differs from glibc's in ~1 % of the pairs, and the division by 2k = 20 turns one ulp of the logarithm into up to 1.6
// ulps of the quotient: measured on the bench matrix, 0.2 % of the distances were 2 ulps away from the host's.
// So log x is computed in double-double (error ~0.01 ulp), which shows which doubles glibc can have returned; the
// host's two roundings are repeated on them (details at the end of the function).  The distance is the host's bit for
// bit in ~98.5 % of the pairs and ONE ulp away in the rest -- the tolerance north_star states, by construction.
//   x = m * 2^e, m in [sqrt(1/2), sqrt(2)];  log x = e ln2 + 2 atanh(s),  s = (m - 1) / (m + 1)  (|s| <= 0.1716)
// IEEE arithmetic and fma only: the CPU tests run the same code against glibc and against 60-digit decimals.
// Arguments here are >= 1; anything that is not a positive finite normal number goes to the library function.
// ---------------------------------------------------------------------------------------------------
KSSD_HD double kssd_log_over_k(double x, double k)
{
    uint64_t ix;
    __builtin_memcpy(&ix, &x, 8);
    if (!(x >= 2.2250738585072014e-308) || (ix >> 52) >= 0x7FFull) {  // 0, < 0, subnormal, inf, nan: the library function
#if defined(__HIP_DEVICE_COMPILE__)
        return log(x) / k;
#else
        return __builtin_log(x) / k;
#endif
    }
    int e = (int)(ix >> 52) - 1023;
    uint64_t im = (ix & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    __builtin_memcpy(&m, &im, 8);                       // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }   // [sqrt(1/2), sqrt(2)], exact
    const double f = m - 1.0;                           // exact
    const double dh = 2.0 + f;                          // m + 1 = dh + dl exactly (|f| < 2)
    const double dl = (2.0 - dh) + f;
    const double sh = f / dh;
    const double r = __builtin_fma(-sh, dh, f) - sh * dl;  // f - sh (dh + dl)
    const double sl = r / dh;                           // s = sh + sl to ~2^-105
    const double z = sh * sh;
    // atanh(s) / s - 1 = z/3 + z^2/5 + ... ; z <= 0.0295: the term after z^12/25 is below 2^-64
    double q = 1.0 / 25.0;
    q = __builtin_fma(q, z, 1.0 / 23.0);
    q = __builtin_fma(q, z, 1.0 / 21.0);
    q = __builtin_fma(q, z, 1.0 / 19.0);
    q = __builtin_fma(q, z, 1.0 / 17.0);
    q = __builtin_fma(q, z, 1.0 / 15.0);
    q = __builtin_fma(q, z, 1.0 / 13.0);
    q = __builtin_fma(q, z, 1.0 / 11.0);
    q = __builtin_fma(q, z, 1.0 / 9.0);
    q = __builtin_fma(q, z, 1.0 / 7.0);
    q = __builtin_fma(q, z, 1.0 / 5.0);
    q = __builtin_fma(q, z, 1.0 / 3.0);
    const double w = z * q;                             // <= 0.0099: its rounding error is ~0.01 ulp of the logarithm
    const double hi1 = 2.0 * sh;                        // log m = hi1 + lo1
    const double lo1 = 2.0 * sl + hi1 * w;
    const double ln2_hi = 6.93147180369123816490e-01;   // 21 trailing zero bits: e * ln2_hi is exact
    const double ln2_lo = 1.90821492927058770002e-10;
    const double ed = (double)e;
    const double t = ed * ln2_hi;
    const double a = t + hi1;                           // TwoSum(t, hi1) = a + b
    const double bb = a - t;
    const double b = (t - (a - bb)) + (hi1 - bb);
    const double c = b + (lo1 + ed * ln2_lo);
    const double lh = a + c;                            // log x = lh + ll (Fast2Sum: |a| >= |c|)
    const double ll = c - (lh - a);
    const double qh = lh / k;                           // what the host computes if its log returned lh
    if (lh == 0.0) return qh;                           // x = 1
    // Which double does glibc return?  Its log is within 0.519 ulp of log x = lh + ll.  If ll is clearly less than half
    // an ulp of lh, the neighbours of lh are more than 0.519 ulp away from the truth and glibc's value MUST be lh: then
    // qh is the host's distance bit for bit (92 % of the arguments).
    uint64_t lb;
    __builtin_memcpy(&lb, &lh, 8);
    const uint64_t ub = (((lb >> 52) & 0x7FFull) - 52) << 52;  // lh >= 2^-53 here: exponent field > 52
    double u;
    __builtin_memcpy(&u, &ub, 8);                       // ulp(lh)
    if (__builtin_fabs(ll) < 0.46 * u) return qh;
    // Otherwise glibc may also have returned the neighbour on ll's side.  If both candidates give the same or adjacent
    // distances, qh is within one ulp of the host whichever it chose (and equal to it whenever glibc rounded
    // correctly); if the division pulled them two ulps apart, the correctly rounded quotient of the double-double
    // logarithm lies between them: within 0.53 ulp of log(x)/k, the host within 1.33 ulp of it -> at most ONE ulp apart.
    const double h2 = (ll > 0 ? lh + u : lh - u) / k;
    uint64_t q1, q2;
    __builtin_memcpy(&q1, &qh, 8);
    __builtin_memcpy(&q2, &h2, 8);
    if (q1 - q2 + 1ull <= 2ull) return qh;              // |q1 - q2| <= 1
    const double rem = __builtin_fma(-qh, k, lh) + ll;  // (lh + ll) / k, rounded once
    return qh + rem / k;
}

#include <vector>
// Host-side construction of the two device tables from the accepted sub-contexts
// (accepted[r] = the sub-context whose permutation rank is r, r < dim_end).
//   T1  stage-1 pattern set: accepted sub-contexts and their reverse complements, every one entered
//       at the GW window offsets a group can see it in; entry bit j <=> window j of the group
//   bloom  stage-1.5 blocked Bloom filter of the same pattern set
//   G   stage-2 exact map sub-context -> rank (two-choice cuckoo, see KssdG)
// returns false when no placement of the keys is found in 2^16 attempts (an attempt places 4 096 distinct keys greedily with
// probability ~1 / 400: hundreds of attempts are ordinary, microseconds each; more than four EQUAL keys can never be placed --
// callers check that the sub-contexts are distinct)
static inline bool kssd_build_tables(KssdParams &P, const std::vector<uint32_t> &accepted, int GW,
                                     std::vector<uint8_t> &T1, std::vector<uint32_t> &bloom, std::vector<KssdG> &G)
{
    // stage-1 group-filter table (see KssdGrp): W, IDXB as the kernel instantiation uses them
    const int LP = 2 * P.subk;
    const int W = (LP >= GW + 1) ? GW : (LP - 1);
    const int IDXB = (2 * LP < 17) ? 2 * LP : 17;
    T1.assign(KSSD_T1_BYTES, 0);
    auto add = [&](uint64_t x) {
        for (int j = 0; j < W; j++) {
            const int start = 2 * (W - 1 - j);                                // index field starts here, counted from the pattern's top bit
            const int avail = IDXB < 2 * LP - start ? IDXB : 2 * LP - start;  // field bits that lie inside window j
            const uint32_t known = (uint32_t)((x >> (2 * LP - start - avail)) & ((1ull << avail) - 1ull));
            const int wild = IDXB - avail;
            for (uint32_t fill = 0; fill < (1u << wild); fill++) T1[(known << wild) | fill] |= (uint8_t)(1u << j);
        }
    };
    bloom.assign(KSSD_BLOOM_WORDS, 0);
    auto add_bloom = [&](uint32_t x) {
        const uint32_t h = kssd_bloom_hash(x);
        bloom[kssd_bloom_word(h)] |= kssd_bloom_bits(h);
    };
    for (size_t r = 0; r < accepted.size(); r++) {
        const uint32_t rc = (uint32_t)kssd_revcomp(accepted[r], LP);
        add(accepted[r]);
        add(rc);
        add_bloom(accepted[r]);
        add_bloom(rc);
    }
    // every key into the bucket its first multiplier names, or -- that one full -- into its second bucket, the first one marked.
    // Both full (3 - 4 keys of 4 096 at a quarter load): one of the four keys that sit there moves to ITS other bucket if that one
    // has room (a key may sit in its first bucket, or in its second with the first one marked: moving home needs nothing, moving
    // out marks the bucket it leaves), and the key takes the freed slot.  Without that step an attempt succeeded once in a few
    // hundred and a context took 20 - 60 ms of multiplier lottery; with it the first attempt nearly always does.  Still nothing
    // free: new multipliers, start over.
    const size_t nb = (size_t)1 << P.g_log2;
    uint64_t seed = 0x243F6A8885A308D3ull;
    auto put = [&](size_t b, int k, uint32_t key, uint32_t rank) {
        G[2 * b + k].key = key;
        G[2 * b + k].rank = (G[2 * b + k].rank & KSSD_G_MOVED) | rank;  // (slot 0 carries its bucket's mark)
    };
    auto free_slot = [&](size_t b) -> int {
        for (int k = 0; k < 2; k++)
            if (G[2 * b + k].key == KSSD_EMPTY_KEY) return k;
        return -1;
    };
    for (int attempt = 0;; attempt++) {
        if (attempt == (1 << 16)) return false;
        G.assign(2 * nb, KssdG{KSSD_EMPTY_KEY, 0});
        bool ok = true;
        for (size_t r = 0; r < accepted.size() && ok; r++) {
            const uint32_t key = accepted[r];
            const size_t b0 = kssd_g_slot(key, P.g_mul[0], P.g_log2), b1 = kssd_g_slot(key, P.g_mul[1], P.g_log2);
            int k = free_slot(b0);
            if (k >= 0) { put(b0, k, key, (uint32_t)r); continue; }
            G[2 * b0].rank |= KSSD_G_MOVED;  // (the key will not sit in its first bucket -- unless a slot is freed there below: the mark is harmless then)
            k = free_slot(b1);
            if (k >= 0) { put(b1, k, key, (uint32_t)r); continue; }
            ok = false;
            for (int pass = 0; pass < 2 && !ok; pass++) {
                const size_t b = pass ? b1 : b0;
                for (int v = 0; v < 2 && !ok; v++) {
                    const uint32_t vkey = G[2 * b + v].key, vrank = G[2 * b + v].rank & ~KSSD_G_MOVED;
                    const size_t v0 = kssd_g_slot(vkey, P.g_mul[0], P.g_log2), v1 = kssd_g_slot(vkey, P.g_mul[1], P.g_log2);
                    const size_t alt = b == v0 ? v1 : v0;  // (v0 == v1: no other bucket)
                    if (alt == b) continue;
                    const int ka = free_slot(alt);
                    if (ka < 0) continue;
                    put(alt, ka, vkey, vrank);
                    if (alt == v1) G[2 * v0].rank |= KSSD_G_MOVED;  // the victim left its first bucket
                    put(b, v, key, (uint32_t)r);                    // (b = b1: the key's first bucket is marked above)
                    ok = true;
                }
            }
        }
        if (ok) break;
        for (int i = 0; i < 2; i++) {  // splitmix64 -> odd multipliers
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            P.g_mul[i] = (uint32_t)(z ^ (z >> 31)) | 1u;
        }
    }
    return true;
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
