// init_probe: where the fixed cost of a command goes before its first useful kernel ("runtime start").
// Times, in one fresh process: hipInit, the first call that creates the device context, the first allocation, a kernel out
// of this program's own (tiny) code object, then the product library: a context for the search, a two-genome search (the
// first kernel out of libkssd_gpu.so's code object: the object is loaded here), the same search again.
//   hipcc --offload-arch=gfx950 -O2 profiles/init_probe.hip -o profiles/init_probe -Iinclude -Lpublic_kssd_amd -lkssd_gpu
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "kssd_gpu.h"

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
__global__ void tiny(uint32_t *p) { p[threadIdx.x] = threadIdx.x; }

int main()
{
    double t[16];
    int n = 0;
    t[n++] = now();
    if (hipInit(0) != hipSuccess) return 1;
    t[n++] = now();  // 1: hipInit
    if (hipSetDevice(0) != hipSuccess || hipFree(0) != hipSuccess) return 1;
    t[n++] = now();  // 2: device context
    uint32_t *d = nullptr;
    if (hipMalloc(&d, 1 << 20) != hipSuccess) return 1;
    t[n++] = now();  // 3: first allocation
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    t[n++] = now();  // 4: first kernel (own code object)
    void *pin = nullptr;
    if (hipHostMalloc(&pin, 64 << 20, 0) != hipSuccess) return 1;
    t[n++] = now();  // 5: 64 MiB page-locked
    kssd_gpu_ctx *c = nullptr;
    if (kssd_gpu_create_for_dist(&c, 10, 0) != 0) return 2;
    t[n++] = now();  // 6: search context
    const uint64_t off[3] = {0, 3, 5};
    const uint32_t ids[5] = {5, 9, 77, 9, 1000};
    uint32_t shared[4];
    double j[4], m[4], cc[4], a[4];
    if (kssd_gpu_dist(c, off, ids, 2, off, ids, 2, shared, j, m, cc, a) != 0) return 3;
    t[n++] = now();  // 7: first search (library code object loaded)
    if (kssd_gpu_dist(c, off, ids, 2, off, ids, 2, shared, j, m, cc, a) != 0) return 3;
    t[n++] = now();  // 8: second search
    // a sketch context (tables of 4 096 accepted sub-contexts built and uploaded), then a second one
    {
        kssd_shuf_hdr hdr = {1, 10, 6, 3};
        static uint32_t acc[4096];
        uint32_t x = 12345u;
        for (int i = 0; i < 4096; i++) { x = x * 1664525u + 1013904223u; acc[i] = ((x >> 8) & 0xFFF000u) | (uint32_t)i; }  // distinct, below 16^6
        kssd_gpu_ctx *s1 = nullptr, *s2 = nullptr;
        if (kssd_gpu_create_compact(&s1, &hdr, acc, 4096, 0) != 0) return 4;
        t[n++] = now();  // 9
        if (kssd_gpu_create_compact(&s2, &hdr, acc, 4096, 0) != 0) return 4;
        t[n++] = now();  // 10
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 5;
        t[n++] = now();  // 11
        hipStream_t st;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 5;
        t[n++] = now();  // 12
    }
    const char *what[] = {"hipInit", "hipSetDevice + hipFree(0)", "hipMalloc 1 MiB", "first kernel, own code object", "hipHostMalloc 64 MiB",
                          "kssd_gpu_create_for_dist", "first kssd_gpu_dist (2 x 2)", "second kssd_gpu_dist",
                          "kssd_gpu_create_compact (first)", "kssd_gpu_create_compact (second)", "hipGetDeviceProperties", "hipStreamCreateWithFlags"};
    for (int i = 1; i < n; i++) printf("%-36s %8.2f ms\n", what[i - 1], (t[i] - t[i - 1]) * 1e3);
    printf("%-36s %8.2f ms   shared %u %u %u %u\n", "total", (t[n - 1] - t[0]) * 1e3, shared[0], shared[1], shared[2], shared[3]);
    return 0;
}
