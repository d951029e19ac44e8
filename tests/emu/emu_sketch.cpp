// emu_sketch.cpp -- TEST HELPER (not part of the product, never loaded by public_kssd_amd).
// Drives the host/device-shared bit manipulation of public_kssd_amd/csrc/kssd_core.h (stage 1 group
// filter, stage 1.5 Bloom test, stage 2 exact evaluation) lane by lane on the CPU, so that the packed layout, the table
// construction and every shift can be checked against the oracle without a GPU.  The wave-level
// machinery (queues, ballots, atomics, LDS) only exists in the HIP kernels and is covered by -m gpu tests.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../public_kssd_amd/csrc/kssd_core.h"

template <int SUBK, int GW>
static void run(const KssdParams &P, const uint32_t *packed, const uint32_t *mask, uint64_t n_chunks, const uint32_t *gid,
                const uint8_t *T1, const uint32_t *bloom, const KssdG *G, std::vector<uint64_t> &out, uint64_t *n_cand,
                std::vector<uint64_t> *where = nullptr)
{
    const int64_t total = (int64_t)n_chunks * KSSD_CHUNK;
    for (uint64_t c = 0; c < n_chunks; c++) {
        const int64_t cbeg = (int64_t)c * KSSD_CHUNK;
        int64_t lo = cbeg, hi = cbeg + KSSD_CHUNK;
        if (c > 0 && gid[c - 1] == gid[c]) lo -= KSSD_CHUNK;
        if (c + 1 < n_chunks && gid[c + 1] == gid[c]) hi += KSSD_CHUNK;
        if (hi > total) hi = total;
        for (int lane = 0; lane < 64; lane++) {
            uint32_t W[5];
            for (int i = 0; i < 5; i++) W[i] = packed[c * 256 + lane * 4 + i];
            uint32_t cl, ch;
            kssd_stage1g<SUBK, GW>(W, T1, cl, ch);
            cl &= mask[c * 128 + lane * 2];
            ch &= mask[c * 128 + lane * 2 + 1];
            for (int b = 0; b < 64; b++) {
                const uint32_t bit = b < 32 ? (cl >> b) & 1u : (ch >> (b - 32)) & 1u;
                if (!bit) continue;
                n_cand[0]++;
                const uint32_t h = kssd_bloom_hash(kssd_extract_m<SUBK>(W, (uint32_t)b));
                const uint32_t bits = kssd_bloom_bits(h);
                if ((bloom[kssd_bloom_word(h)] & bits) != bits) continue;
                n_cand[1]++;
                uint32_t dr;
                if (kssd_carry_ok(P)) {
                    // what the Bloom round's lane cuts out of the three packed words around the candidate's position must give
                    // the pattern the scanning lane saw and the k-mer the packed stream holds
                    const uint64_t p = (uint64_t)(cbeg + lane * 64 + b), wi = p >> 4;
                    const uint32_t *pp = packed + (wi ? wi - 1 : 0);
                    uint32_t top32, front;
                    kssd_carry_from_words(wi ? pp[0] : 0u, wi ? pp[1] : pp[0], wi ? pp[2] : pp[1], (uint32_t)p & 15u, top32, front);
                    if ((top32 >> (32 - 4 * SUBK)) != kssd_extract_m<SUBK>(W, (uint32_t)b)) { n_cand[0] = ~0ull; return; }
                    const int64_t b0 = cbeg + lane * 64 + b - P.out;
                    if (b0 >= 0) {
                        const uint64_t pw = (uint64_t)b0 >> 4;
                        uint64_t u1, u2;
                        uint32_t d1, d2;
                        kssd_s2_decode(P, packed[pw], packed[pw + 1], packed[pw + 2], ~0u, ~0u, (uint32_t)b0, u1, d1);
                        kssd_s2_canon(P, kssd_carry_fwd(P, kssd_carry_payload(top32, front)), u2, d2);
                        if (u1 != u2 || d1 != d2) { n_cand[0] = ~0ull; return; }
                    }
                }
                if (kssd_stage2(P, cbeg + lane * 64 + b, lo, hi, packed, mask, G, dr)) {
                    out.push_back(((uint64_t)gid[c] << 32) | dr);
                    if (where) where->push_back((uint64_t)(cbeg + lane * 64 + b));  // position of the sub-context start
                }
            }
        }
    }
}

// also the brute-force variant: stage 2 on every position, to prove stage 1 loses nothing
static void run_all(const KssdParams &P, const uint32_t *packed, const uint32_t *mask, uint64_t n_chunks, const uint32_t *gid,
                    const KssdG *G, std::vector<uint64_t> &out)
{
    const int64_t total = (int64_t)n_chunks * KSSD_CHUNK;
    for (uint64_t c = 0; c < n_chunks; c++) {
        const int64_t cbeg = (int64_t)c * KSSD_CHUNK;
        int64_t lo = cbeg, hi = cbeg + KSSD_CHUNK;
        if (c > 0 && gid[c - 1] == gid[c]) lo -= KSSD_CHUNK;
        if (c + 1 < n_chunks && gid[c + 1] == gid[c]) hi += KSSD_CHUNK;
        if (hi > total) hi = total;
        for (int p = 0; p < KSSD_CHUNK; p++) {
            uint32_t dr;
            if (kssd_stage2(P, cbeg + p, lo, hi, packed, mask, G, dr)) out.push_back(((uint64_t)gid[c] << 32) | dr);
        }
    }
}

static long emu_impl(int k, int subk, int drlevel, const int32_t *table, const uint32_t *packed, const uint32_t *mask,
                     uint64_t n_chunks, const uint32_t *chunk_gid, int brute, uint64_t *out, uint64_t cap,
                     uint64_t *n_cand, int gw, uint64_t *where_out);

extern "C" long emu_sketch(int k, int subk, int drlevel, const int32_t *table, const uint32_t *packed, const uint32_t *mask,
                           uint64_t n_chunks, const uint32_t *chunk_gid, int brute, uint64_t *out, uint64_t cap,
                           uint64_t *n_cand, int gw)
{
    return emu_impl(k, subk, drlevel, table, packed, mask, n_chunks, chunk_gid, brute, out, cap, n_cand, gw, nullptr);
}

// the same, and where[i] = batch position (of the sub-context start) entry i was sampled at; entries come in position order
extern "C" long emu_sketch_where(int k, int subk, int drlevel, const int32_t *table, const uint32_t *packed, const uint32_t *mask,
                                 uint64_t n_chunks, const uint32_t *chunk_gid, uint64_t *out, uint64_t *where, uint64_t cap)
{
    uint64_t n_cand[2];
    return emu_impl(k, subk, drlevel, table, packed, mask, n_chunks, chunk_gid, 0, out, cap, n_cand, 0, where);
}

static long emu_impl(int k, int subk, int drlevel, const int32_t *table, const uint32_t *packed, const uint32_t *mask,
                     uint64_t n_chunks, const uint32_t *chunk_gid, int brute, uint64_t *out, uint64_t cap,
                     uint64_t *n_cand, int gw, uint64_t *where_out)
{
    std::vector<uint64_t> where;
    std::vector<uint64_t> *wp = where_out ? &where : nullptr;
    KssdParams P;
    if (kssd_params_init(&P, k, subk, drlevel) != 0) return -1;
    std::vector<uint32_t> acc;
    if (!kssd_accepted_from_table(P, table, acc)) return -2;
    std::vector<uint8_t> T1;
    std::vector<uint32_t> bloom;
    std::vector<KssdG> G;
    if (gw == 0) gw = KSSD_GW;
    kssd_build_tables(P, acc, gw, T1, bloom, G);
    std::vector<uint64_t> res;
    n_cand[0] = n_cand[1] = 0;  // after stage 1, after stage 1.5
    if (brute) run_all(P, packed, mask, n_chunks, chunk_gid, G.data(), res);
    else {
#define RUN(S)                                                                                           \
    case S:                                                                                              \
        if (gw == 4) run<S, 4>(P, packed, mask, n_chunks, chunk_gid, T1.data(), bloom.data(), G.data(), res, n_cand, wp); \
        else if (gw == 5) run<S, 5>(P, packed, mask, n_chunks, chunk_gid, T1.data(), bloom.data(), G.data(), res, n_cand, wp); \
        else return -5;                                                                                  \
        break;
        switch (subk) {
            RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
            default: return -3;
        }
#undef RUN
    }
    if (res.size() > cap) return -4;
    memcpy(out, res.data(), res.size() * 8);
    if (where_out) memcpy(where_out, where.data(), where.size() * 8);
    return (long)res.size();
}

// the distance epilogue's log(x) / k (kssd_core.h:kssd_log_over_k), as the device evaluates it
extern "C" void emu_log_over_k(const double *x, double k, double *y, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) y[i] = kssd_log_over_k(x[i], k);
}

// the exact table (kssd_build_tables: every key into one of its two buckets, at most one key moved to make room) against its own
// lookup: every accepted sub-context found with its rank, sub-contexts that are not accepted not found; returns the number of
// wrong answers over `sets` random sets, -1 if a table could not be built
extern "C" long emu_check_exact_table(int k, int subk, int drlevel, uint32_t seed, int sets, int absent_per_set)
{
    KssdParams P0;
    if (kssd_params_init(&P0, k, subk, drlevel) != 0) return -2;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
    const uint32_t space = 1u << (4 * subk);
    long bad = 0;
    for (int t = 0; t < sets; t++) {
        KssdParams P = P0;
        std::vector<uint8_t> in(space, 0);
        std::vector<uint32_t> acc;
        while (acc.size() < P.dim_end) {
            const uint32_t v = next() % space;
            if (!in[v]) { in[v] = 1; acc.push_back(v); }
        }
        std::vector<uint8_t> T1;
        std::vector<uint32_t> bloom;
        std::vector<KssdG> G;
        if (!kssd_build_tables(P, acc, KSSD_GW, T1, bloom, G)) return -1;
        for (uint32_t r = 0; r < acc.size(); r++) {
            uint32_t rank = ~0u;
            if (!kssd_g_find(P, G.data(), acc[r], rank) || rank != r) bad++;
        }
        for (int i = 0; i < absent_per_set; i++) {
            const uint32_t v = next() % space;
            uint32_t rank;
            if (!in[v] && kssd_g_find(P, G.data(), v, rank)) bad++;
        }
    }
    return bad;
}
