// scanbench -- development tool (GPU box): times the sketch scan kernel of libkssd_gpu.so on random packed
// genomes without Python, prints a checksum of the sketches so that kernel variants can be compared, and
// holds the FETCH_SIZE calibration kernels (known byte counts read with 16 / 8 / 4 B per lane).
//
//   profiles/scanbench [genomes=400] [length=5000000] [reps=10]      env KSSD_DEV_ABLATE=<1..3> (profiles/sb_modes.sh)
//   profiles/scanbench calib                                          (run under rocprofv3 --pmc FETCH_SIZE)
// build: make -C public_kssd_amd tools
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../include/kssd_gpu.h"
extern "C" int kssd_gpu_dev_wavetimes(unsigned long long *out, uint32_t n_waves);  // libkssd_gpu_dev.so only
extern "C" int kssd_gpu_dev_deduptimes(unsigned long long *out, uint32_t n_genomes);

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

static inline uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_random(uint32_t *p, size_t n, uint64_t seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (uint32_t)(z ^ (z >> 31));
    }
}

// ---- FETCH_SIZE calibration: each kernel reads exactly `bytes` once, coalesced, at one width per lane ----
__global__ void calib_read16(const uint4 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void calib_read8(const uint2 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void calib_read4(const uint32_t *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345u) *sink = acc;
}

static int calib()
{
    const size_t bytes = (size_t)1 << 30;  // 1 GiB per kernel, far beyond the 256 MiB Infinity Cache
    uint32_t *buf, *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    fill_random<<<4096, 256>>>(buf, bytes / 4, 1);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        calib_read16<<<2048, 256>>>((const uint4 *)buf, bytes / 16, sink);
        calib_read8<<<2048, 256>>>((const uint2 *)buf, bytes / 8, sink);
        calib_read4<<<2048, 256>>>(buf, bytes / 4, sink);
    }
    CK(hipDeviceSynchronize());
    printf("calib: every calib_read* launch read %zu bytes\n", bytes);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "calib")) return calib();
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 400;
    const uint64_t L = argc > 2 ? (uint64_t)atoll(argv[2]) : 5000000;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const int k = getenv("SB_K") ? atoi(getenv("SB_K")) : 10, subk = getenv("SB_SUBK") ? atoi(getenv("SB_SUBK")) : 6,
              drl = getenv("SB_DRL") ? atoi(getenv("SB_DRL")) : 3;

    // a random accepted set (what a .shuf would give): dim_end distinct sub-contexts
    uint64_t seed = 20260101;
    const uint64_t space = 1ull << (4 * subk);
    uint64_t sub = 1ull << (4 * (subk - drl));
    const uint32_t dim_end = (uint32_t)(sub > 4096 ? sub : 4096);
    std::vector<uint32_t> acc;
    {
        std::vector<uint32_t> all;
        while (all.size() < 2 * (size_t)dim_end) all.push_back((uint32_t)(splitmix(seed) % space));
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        // deterministic shuffle back to random order
        for (size_t i = all.size() - 1; i > 0; i--) std::swap(all[i], all[splitmix(seed) % (i + 1)]);
        acc.assign(all.begin(), all.begin() + dim_end);
    }
    kssd_shuf_hdr hdr = {1, k, subk, drl};
    kssd_gpu_ctx *ctx = nullptr;
    int rc = kssd_gpu_create_compact(&ctx, &hdr, acc.data(), dim_end, 0);
    if (rc) { fprintf(stderr, "create: %s\n", kssd_gpu_strerror(rc)); return 2; }

    const uint64_t chunks = (L + KSSD_CHUNK_BASES - 1) / KSSD_CHUNK_BASES;
    const uint64_t n_chunks = chunks * G;
    uint32_t *d_p, *d_m, *d_ids;
    uint64_t *d_off;
    CK(hipMalloc(&d_p, (n_chunks * KSSD_CHUNK_WORDS + 64) * 4));
    CK(hipMalloc(&d_m, (n_chunks * KSSD_CHUNK_MASKW + 64) * 4));
    fill_random<<<4096, 256>>>(d_p, n_chunks * KSSD_CHUNK_WORDS + 64, 42);
    CK(hipMemset(d_m, 0xFF, (n_chunks * KSSD_CHUNK_MASKW + 64) * 4));
    // tail of every genome beyond L and a sprinkle of invalid positions
    {
        std::vector<uint32_t> one(chunks * KSSD_CHUNK_MASKW, 0xFFFFFFFFu);
        for (uint64_t p = L; p < chunks * KSSD_CHUNK_BASES; p++) one[p >> 5] &= ~(1u << (p & 31));
        uint64_t s2 = 7;
        for (int i = 0; i < (int)(L / 10000); i++) { uint64_t p = splitmix(s2) % L; one[p >> 5] &= ~(1u << (p & 31)); }
        for (uint32_t g = 0; g < G; g++)
            CK(hipMemcpy(d_m + g * chunks * KSSD_CHUNK_MASKW, one.data(), one.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<uint64_t> chunk_off(G + 1);
    for (uint32_t g = 0; g <= G; g++) chunk_off[g] = g * chunks;
    const uint64_t cap = (uint64_t)((double)G * L * dim_end / (double)space * 1.3) + 4096;
    CK(hipMalloc(&d_ids, cap * 4));
    CK(hipMalloc(&d_off, (G + 1) * 8));
    CK(hipDeviceSynchronize());

    uint64_t total = 0;
    int64_t bad = -1;
    for (int i = 0; i < 6; i++) {
        rc = kssd_gpu_sketch_device(ctx, d_p, d_m, chunk_off.data(), G, 0, 1, d_off, d_ids, cap, nullptr);
        if (rc) { fprintf(stderr, "sketch: %s\n", kssd_gpu_strerror(rc)); return 2; }
        rc = kssd_gpu_sketch_status(ctx, &total, &bad, nullptr);
        if (rc != KSSD_ERR_OVERFLOW) break;
    }
    if (rc) { fprintf(stderr, "status: %s\n", kssd_gpu_strerror(rc)); return 2; }
    for (int i = 0; i < 2; i++) kssd_gpu_sketch_device(ctx, d_p, d_m, chunk_off.data(), G, 0, 1, d_off, d_ids, cap, nullptr);
    CK(hipDeviceSynchronize());
    float ms;
    uint32_t nl;
    kssd_gpu_kernel_time(ctx, 0, 1, &ms, &nl);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; i++) kssd_gpu_sketch_device(ctx, d_p, d_m, chunk_off.data(), G, 0, 1, d_off, d_ids, cap, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float whole_ms = 0;
    CK(hipEventElapsedTime(&whole_ms, e0, e1));
    whole_ms /= reps;
    kssd_gpu_kernel_time(ctx, 0, 1, &ms, &nl);
    rc = kssd_gpu_sketch_status(ctx, &total, &bad, nullptr);

    uint64_t s1 = 0, bl = 0;
    kssd_gpu_scan_stats(ctx, &s1, &bl, nullptr);
    printf("stats: stage1 %llu (%.4f %%) bloom %llu (%.4f %%)\n", (unsigned long long)s1,
           100.0 * s1 / ((double)G * chunks * KSSD_CHUNK_BASES), (unsigned long long)bl,
           100.0 * bl / ((double)G * chunks * KSSD_CHUNK_BASES));
    std::vector<uint32_t> ids(total);
    std::vector<uint64_t> off(G + 1);
    CK(hipMemcpy(ids.data(), d_ids, total * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(off.data(), d_off, (G + 1) * 8, hipMemcpyDeviceToHost));
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < total; i++) h = (h ^ ids[i]) * 1099511628211ull;
    for (uint32_t g = 0; g <= G; g++) h = (h ^ off[g]) * 1099511628211ull;
    const double bytes = 0.375 * (double)G * (double)(chunks * KSSD_CHUNK_BASES) + 4.0 * total;
    printf("whole sketch call (scan + exact + dedup + CSR): %.4f ms\n", whole_ms);
    printf("variant=%s genomes=%u len=%llu rc=%d ids=%llu checksum=%016llx scan_ms=%.4f (%u launches) algorithmic %.1f GB/s\n",
           getenv("KSSD_DEV_ABLATE") ? getenv("KSSD_DEV_ABLATE") : "product", G, (unsigned long long)L, rc,
           (unsigned long long)total, (unsigned long long)h, ms, nl, bytes / ms / 1e6);
    if (getenv("KSSD_DEV_WAVETIME")) {
        // where the launch's time goes per wave: start skew, table copy, the chunk loop, and how long the machine waits for its
        // slowest wave (static partition: every wave owns the same number of chunks)
        const uint32_t nw = 256 * 16;
        std::vector<unsigned long long> t(nw * 3);
        if (kssd_gpu_dev_wavetimes(t.data(), nw) == 0) {
            // (time stamps of different XCDs do not share a base: only differences inside a wave are compared)
            std::vector<double> tab(nw), loop(nw);
            for (uint32_t w = 0; w < nw; w++) {
                tab[w] = (double)(t[3 * w + 1] - t[3 * w]);
                loop[w] = (double)(t[3 * w + 2] - t[3 * w + 1]);
            }
            auto pr = [&](const char *nm, std::vector<double> v) {
                std::sort(v.begin(), v.end());
                double mean = 0;
                for (double x : v) mean += x;
                mean /= (double)v.size();
                printf("  %-34s min %9.0f  p10 %9.0f  median %9.0f  mean %9.0f  p90 %9.0f  max %9.0f\n", nm, v[0], v[v.size() / 10], v[v.size() / 2], mean,
                       v[v.size() * 9 / 10], v.back());
            };
            printf("per-wave ticks of the last scan launch (4096 waves):\n");
            pr("table copy + barrier", tab);
            pr("chunk loop", loop);
            std::vector<double> cumax(256), cumin(256), xcd(8, 0.0);
            for (uint32_t b = 0; b < 256; b++) {
                double mx = 0, mn = 1e30;
                for (uint32_t w = 0; w < 16; w++) { mx = std::max(mx, loop[b * 16 + w]); mn = std::min(mn, loop[b * 16 + w]); xcd[b % 8] += loop[b * 16 + w] / (32.0 * 16.0); }
                cumax[b] = mx;
                cumin[b] = mn;
            }
            pr("slowest wave of a workgroup", cumax);
            pr("fastest wave of a workgroup", cumin);
            printf("  mean chunk loop by XCD (workgroup %% 8):");
            for (int x = 0; x < 8; x++) printf(" %.0f", xcd[x]);
            double mean = 0, mx = 0;
            for (uint32_t w = 0; w < nw; w++) { mean += loop[w] / nw; mx = std::max(mx, loop[w]); }
            printf("\n  mean / max of the chunk loops: %.3f (what a launch that ended with its average wave would take)\n", mean / mx);
        }
    }
    if (getenv("KSSD_DEV_DEDUPTIME")) {
        std::vector<unsigned long long> t((size_t)G * 4);
        if (kssd_gpu_dev_deduptimes(t.data(), G) == 0) {
            // (s_memrealtime: 10 ns ticks on one base for the whole chip)
            double a = 0, b = 0, c = 0;
            unsigned long long first = ~0ull, last = 0;
            for (uint32_t g = 0; g < G; g++) {
                a += (double)(t[4 * g + 1] - t[4 * g]); b += (double)(t[4 * g + 2] - t[4 * g + 1]); c += (double)(t[4 * g + 3] - t[4 * g + 2]);
                first = std::min(first, t[4 * g]);
                last = std::max(last, t[4 * g + 3]);
            }
            if (getenv("KSSD_DEV_GATHERSPLIT"))
                printf("per-genome kernel, mean us per workgroup: block table + owners %.2f, first round of records %.2f, further rounds %.2f\n", a / G / 100, b / G / 100, c / G / 100);
            else
                printf("per-genome kernel, mean us per workgroup: candidates -> keys in LDS %.2f, sort %.2f, runs + keep rules + write %.2f\n", a / G / 100, b / G / 100, c / G / 100);
            std::vector<double> st(G), en(G);
            for (uint32_t g = 0; g < G; g++) { st[g] = (double)(t[4 * g] - first) / 100; en[g] = (double)(t[4 * g + 3] - first) / 100; }
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            printf("  workgroup starts after the first one (us): median %.2f  p90 %.2f  max %.2f;  ends: median %.2f  p90 %.2f  last %.2f\n", st[G / 2], st[G * 9 / 10],
                   st[G - 1], en[G / 2], en[G * 9 / 10], (double)(last - first) / 100);
        }
    }
    kssd_gpu_destroy(ctx);
    return 0;
}
