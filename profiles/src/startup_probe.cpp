// Where a process's first 0.2 s on the device go: the library's load, the runtime's start, the device's context, the code object, the
// first page-locked and device allocations.  Built and run on the GPU box by profiles/startup_probe.sh.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    double t0 = now(), t;
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    t = now(); printf("dlopen(libkssd_gpu.so)      %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    int n = 0;
    hipGetDeviceCount(&n);
    t = now(); printf("hipGetDeviceCount -> %d      %7.1f ms\n", n, 1e3 * (t - t0)); t0 = t;
    hipSetDevice(0); hipFree(nullptr);
    t = now(); printf("hipSetDevice + hipFree(0)   %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    int (*warm)(int) = (int (*)(int))dlsym(lib, "kssd_gpu_warm_up");
    int rc = warm(0);
    t = now(); printf("kssd_gpu_warm_up rc %d       %7.1f ms (code object + first launch)\n", rc, 1e3 * (t - t0)); t0 = t;
    void *h[8];
    for (int i = 0; i < 4; i++) {
        hipHostMalloc(&h[i], 96u << 20, hipHostMallocDefault);
        t = now(); printf("hipHostMalloc 96 MiB #%d     %7.1f ms\n", i, 1e3 * (t - t0)); t0 = t;
    }
    void *d;
    hipMalloc(&d, 1u << 30);
    t = now(); printf("hipMalloc 1 GiB             %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    hipMalloc(&d, 256u << 20);
    t = now(); printf("hipMalloc 256 MiB           %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    t = now(); printf("hipStreamCreate             %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    hipMemcpyAsync(d, h[0], 96u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    t = now(); printf("first H2D 96 MiB            %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    hipMemcpyAsync(d, h[1], 96u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    t = now(); printf("second H2D 96 MiB           %7.1f ms\n", 1e3 * (t - t0)); t0 = t;
    return 0;
}
