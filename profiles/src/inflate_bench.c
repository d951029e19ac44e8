/* kssd_gunzip_mem on one .gz file, N times: MB/s of text out (best and median).  cc -O2 inflate_bench.c -L.. -lkssd_host */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "../../public_kssd_amd/host/kssd_host.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *z = malloc(n); if (fread(z, 1, n, f) != (size_t)n) return 1;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    unsigned char *out = NULL; size_t cap = 0, len = 0;
    double t[256];
    for (int r = 0; r < reps; r++) {
        double t0 = now();
        int rc = kssd_gunzip_mem(z, n, &out, &cap, &len);
        t[r] = now() - t0;
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    qsort(t, reps, sizeof *t, cmp);
    printf("%s: %ld -> %zu bytes, best %.0f MB/s, median %.0f MB/s\n", argv[1], n, len, len / t[0] / 1e6, len / t[reps / 2] / 1e6);
    return 0;
}
