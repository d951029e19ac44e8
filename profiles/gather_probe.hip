// profiles/gather_probe.hip -- development tool (round 4): what does a divergent (gather) load cost on MI355X, by width?
// Every thread makes N dependent-free random reads of 4, 8 or 16 bytes from a table of 128 KiB (L2 resident: the exact table of
// the sketch path) or 32 MiB (the index of the distance path), 4 waves per SIMD resident; also 16-byte reads whose addresses agree
// inside groups of 4 lanes (one 64-byte line per group: the posting walk).  Prints lane-loads per second and bytes per second.
// build: hipcc --offload-arch=gfx950 -O3 profiles/gather_probe.hip -o profiles/gather_probe      run (GPU box): profiles/gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int WIDTH, int GROUP>
__global__ __launch_bounds__(512) void gather(const uint32_t *__restrict__ tab, uint32_t mask_words, int n, uint32_t *out)
{
    uint32_t x = (blockIdx.x * 512u + threadIdx.x) / GROUP * 2654435761u + 12345u, acc = 0;
    const uint32_t sub = (threadIdx.x % GROUP) * (WIDTH / 4);
#pragma unroll 8
    for (int i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t w = ((x >> 4) & mask_words & ~(uint32_t)(GROUP * WIDTH / 4 - 1)) + sub;  // aligned to the group's span
        if (WIDTH == 4) acc ^= tab[w];
        if (WIDTH == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(tab + w); acc ^= v.x ^ v.y; }
        if (WIDTH == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(tab + w); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int WIDTH, int GROUP>
static void run(const char *name, const uint32_t *tab, size_t bytes, uint32_t *out)
{
    const int n = 256, blocks = 256 * 4 * 4;  // 16 workgroups of 8 waves per CU: four rounds of the resident set
    const uint32_t mask = (uint32_t)(bytes / 4 - 1);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<WIDTH, GROUP>), dim3(blocks), dim3(512), 0, 0, tab, mask, n, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((gather<WIDTH, GROUP>), dim3(blocks), dim3(512), 0, 0, tab, mask, n, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double loads = 5.0 * blocks * 512.0 * n, s = ms * 1e-3;
    printf("%-44s table %6zu KiB: %7.1f G lane-loads/s  %7.2f TB/s  (%.2f cycles per lane-load and CU at 2.4 GHz)\n", name, bytes >> 10, loads / s / 1e9,
           loads * WIDTH / s / 1e12, 2.4e9 * 256 * s / loads);
}
int main()
{
    uint32_t *tab, *out;
    const size_t big = 32u << 20;
    hipMalloc(&tab, big);
    hipMalloc(&out, 64);
    hipMemset(tab, 1, big);
    for (size_t bytes : {(size_t)128 << 10, big}) {
        run<4, 1>("random 4-byte loads", tab, bytes, out);
        run<8, 1>("random 8-byte loads", tab, bytes, out);
        run<16, 1>("random 16-byte loads", tab, bytes, out);
        run<16, 4>("16-byte loads, 4 lanes per 64-byte line", tab, bytes, out);
        run<4, 16>("4-byte loads, 16 lanes per 64-byte line", tab, bytes, out);
        run<4, 64>("4-byte loads, a wave per 256-byte run", tab, bytes, out);
    }
    return 0;
}
