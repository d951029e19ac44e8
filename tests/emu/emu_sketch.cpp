// emu_sketch.cpp -- TEST HELPER (not part of the product, never loaded by public_kssd_amd).
// Drives the host/device-shared bit manipulation of public_kssd_amd/csrc/kssd_core.h (stage 1 quad-core
// filter, stage 2 exact evaluation) lane by lane on the CPU, so that the packed layout, the table
// construction and every shift can be checked against the oracle without a GPU.  The wave-level
// machinery (queues, ballots, atomics, LDS) only exists in the HIP kernels and is covered by -m gpu tests.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../public_kssd_amd/csrc/kssd_core.h"

template <int SUBK>
static void run(const KssdParams &P, const uint32_t *packed, const uint32_t *mask, uint64_t n_chunks, const uint32_t *gid,
                const uint8_t *T1, const KssdG *G, std::vector<uint64_t> &out, uint64_t *n_cand)
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
            kssd_stage1<SUBK>(W, T1, cl, ch);
            cl &= mask[c * 128 + lane * 2];
            ch &= mask[c * 128 + lane * 2 + 1];
            for (int b = 0; b < 64; b++) {
                const uint32_t bit = b < 32 ? (cl >> b) & 1u : (ch >> (b - 32)) & 1u;
                if (!bit) continue;
                (*n_cand)++;
                uint32_t dr;
                if (kssd_stage2(P, cbeg + lane * 64 + b, lo, hi, packed, mask, G, dr)) out.push_back(((uint64_t)gid[c] << 32) | dr);
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

extern "C" long emu_sketch(int k, int subk, int drlevel, const int32_t *table, const uint32_t *packed, const uint32_t *mask,
                           uint64_t n_chunks, const uint32_t *chunk_gid, int brute, uint64_t *out, uint64_t cap,
                           uint64_t *n_cand)
{
    KssdParams P;
    if (kssd_params_init(&P, k, subk, drlevel) != 0) return -1;
    std::vector<uint32_t> acc;
    if (!kssd_accepted_from_table(P, table, acc)) return -2;
    std::vector<uint8_t> T1;
    std::vector<KssdG> G;
    kssd_build_tables(P, acc, T1, G);
    std::vector<uint64_t> res;
    *n_cand = 0;
    if (brute) run_all(P, packed, mask, n_chunks, chunk_gid, G.data(), res);
    else switch (subk) {
        case 2: run<2>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        case 3: run<3>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        case 4: run<4>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        case 5: run<5>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        case 6: run<6>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        case 7: run<7>(P, packed, mask, n_chunks, chunk_gid, T1.data(), G.data(), res, n_cand); break;
        default: return -3;
    }
    if (res.size() > cap) return -4;
    memcpy(out, res.data(), res.size() * 8);
    return (long)res.size();
}
