// profiles/pinned_probe.hip -- development tool: how fast can a host thread WRITE the kinds of host memory a packed batch
// may live in, and how fast does each kind travel to the device?  (hipHostMalloc default / non-coherent / plain malloc)
// build: make -C public_kssd_amd tools     run (GPU box): profiles/pinned_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main()
{
    const size_t N = 64u << 20;
    void *dev = nullptr;
    hipMalloc(&dev, N);
    struct { const char *name; unsigned flags; int kind; } cases[] = {
        {"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, 0},
        {"hipHostMalloc coherent", hipHostMallocCoherent, 0}, {"malloc (pageable)", 0, 1}, {"malloc + hipHostRegister", 0, 2}};
    for (auto &c : cases) {
        void *p = nullptr;
        if (c.kind == 0) { if (hipHostMalloc(&p, N, c.flags) != hipSuccess) { printf("%-28s alloc failed\n", c.name); continue; } }
        else { p = malloc(N); memset(p, 1, N); if (c.kind == 2 && hipHostRegister(p, N, hipHostRegisterDefault) != hipSuccess) { printf("%-28s register failed\n", c.name); continue; } }
        volatile unsigned *w = (volatile unsigned *)p;
        double t0 = now();
        for (size_t i = 0; i < N / 4; i++) w[i] = (unsigned)i * 2654435761u;  // whole words, ascending: what the tokeniser does
        double t1 = now();
        unsigned acc = 0;
        for (size_t i = 0; i < N / 4; i += 16) acc += w[i];
        double t2 = now();
        hipMemcpy(dev, p, N, hipMemcpyHostToDevice);
        double t3 = now();
        hipMemcpy(dev, p, N, hipMemcpyHostToDevice);
        double t4 = now();
        printf("%-28s write %.2f GB/s   strided read %.3f s (%u)   H2D %.1f GB/s (second pass %.1f)\n", c.name, N / (t1 - t0) / 1e9, t2 - t1, acc,
               N / (t3 - t2) / 1e9, N / (t4 - t3) / 1e9);
        if (c.kind == 0) hipHostFree(p); else { if (c.kind == 2) hipHostUnregister(p); free(p); }
    }
    return 0;
}
