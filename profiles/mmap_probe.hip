// profiles/mmap_probe.hip -- development tool (round 4): can the raw text of the input files travel to the device straight from
// the page cache?  For a file of N bytes in tmpfs it times, each on its own:
//   hipInit + context                    (what a process pays before anything else)
//   hipHostMalloc(N)                     page-locked memory the classical way
//   pread(file -> that memory)           the CPU copy the command line's readers make today
//   H2D from it
//   mmap(file, MAP_PRIVATE / MAP_SHARED | MAP_POPULATE) + hipHostRegister of the mapping   (no CPU copy) and H2D from it
//   H2D straight from the unregistered mapping (pageable path of the runtime)
// build: hipcc --offload-arch=gfx950 -O2 profiles/mmap_probe.hip -o profiles/mmap_probe     run (GPU box): profiles/mmap_probe [MiB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t N = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    const char *path = "/dev/shm/kssd_mmap_probe.bin";
    {
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
        char *blk = (char *)malloc(1 << 20);
        for (size_t i = 0; i < (1u << 20); i++) blk[i] = "ACGT\n"[i % 5];
        for (size_t at = 0; at < N; at += 1 << 20) if (write(fd, blk, 1 << 20) != (1 << 20)) { perror("write"); return 1; }
        close(fd);
        free(blk);
    }
    double t0 = now();
    if (hipInit(0) != hipSuccess) { printf("hipInit failed\n"); return 1; }
    hipSetDevice(0);
    hipFree(nullptr);
    double t1 = now();
    printf("hipInit + context                          %.3f s\n", t1 - t0);
    void *dev = nullptr;
    hipMalloc(&dev, N);
    hipStream_t s;
    hipStreamCreate(&s);
    void *pin = nullptr;
    t0 = now();
    if (hipHostMalloc(&pin, N, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    t1 = now();
    printf("hipHostMalloc %zu MiB                      %.3f s (%.2f s/GB)\n", N >> 20, t1 - t0, (t1 - t0) / (N / 1e9));
    int fd = open(path, O_RDONLY);
    t0 = now();
    for (size_t at = 0; at < N;) { ssize_t r = pread(fd, (char *)pin + at, N - at, (off_t)at); if (r <= 0) break; at += (size_t)r; }
    t1 = now();
    printf("pread (1 thread) into page-locked memory   %.3f s (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e9);
    t0 = now();
    hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    t1 = now();
    printf("H2D from page-locked memory                %.3f s (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e9);
    hipHostFree(pin);
    struct { const char *name; int flags; unsigned reg; } cases[] = {
        {"MAP_PRIVATE|POPULATE, register default", MAP_PRIVATE | MAP_POPULATE, hipHostRegisterDefault},
        {"MAP_SHARED|POPULATE, register default", MAP_SHARED | MAP_POPULATE, hipHostRegisterDefault},
        {"MAP_SHARED|POPULATE, register read-only", MAP_SHARED | MAP_POPULATE, hipHostRegisterReadOnly},
        {"MAP_SHARED (no populate), register default", MAP_SHARED, hipHostRegisterDefault}};
    for (auto &c : cases) {
        t0 = now();
        void *m = mmap(nullptr, N, PROT_READ | ((c.flags & MAP_PRIVATE) ? PROT_WRITE : 0), c.flags, fd, 0);
        t1 = now();
        if (m == MAP_FAILED) { printf("%-44s mmap failed\n", c.name); continue; }
        hipError_t e = hipHostRegister(m, N, c.reg);
        double t2 = now();
        if (e != hipSuccess) { printf("%-44s mmap %.3f s, hipHostRegister FAILED: %s\n", c.name, t1 - t0, hipGetErrorString(e)); (void)hipGetLastError(); munmap(m, N); continue; }
        e = hipMemcpyAsync(dev, m, N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t3 = now();
        hipHostUnregister(m);
        double t4 = now();
        munmap(m, N);
        printf("%-44s mmap %.3f s, register %.3f s (%.2f s/GB), H2D %.3f s (%.1f GB/s)%s, unregister %.3f s\n", c.name, t1 - t0, t2 - t1,
               (t2 - t1) / (N / 1e9), t3 - t2, N / (t3 - t2) / 1e9, e == hipSuccess ? "" : " [copy failed]", t4 - t3);
    }
    {
        void *m = mmap(nullptr, N, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        t0 = now();
        hipMemcpyAsync(dev, m, N, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        t1 = now();
        printf("H2D straight from the mapping (pageable)     %.3f s (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e9);
        munmap(m, N);
    }
    // many small registrations: 5 MB files, one registration each (what 1 024 genome files would need)
    {
        const size_t F = 5u << 20, nf = N / F;
        void *m = mmap(nullptr, N, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        t0 = now();
        size_t ok = 0;
        for (size_t i = 0; i < nf; i++) ok += hipHostRegister((char *)m + i * F, F, hipHostRegisterDefault) == hipSuccess;
        t1 = now();
        for (size_t i = 0; i < nf; i++) hipMemcpyAsync((char *)dev + i * F, (char *)m + i * F, F, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t2 = now();
        for (size_t i = 0; i < nf; i++) hipHostUnregister((char *)m + i * F);
        double t3 = now();
        printf("%zu registrations of 5 MiB: %zu ok, register %.3f s (%.2f ms each), H2D %.3f s (%.1f GB/s), unregister %.3f s\n", nf, ok, t1 - t0,
               (t1 - t0) / nf * 1e3, t2 - t1, N / (t2 - t1) / 1e9, t3 - t2);
        munmap(m, N);
    }
    close(fd);
    unlink(path);
    return 0;
}
