/* kssd_gunzip_mem on one .gz file, N times, and kssd_gunzip_mem2 on the file twice over (two buffers): MB/s of text out, best and
 * median.  cc -O2 inflate_bench.c -L../../public_kssd_amd -lkssd_host */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../public_kssd_amd/host/kssd_host.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *z = malloc(n); if (fread(z, 1, n, f) != (size_t)n) return 1;
    unsigned char *z2 = malloc(n); memcpy(z2, z, n);
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    unsigned char *out = NULL, *out2 = NULL; size_t cap = 0, len = 0, cap2 = 0, len2 = 0;
    double t[256];
    for (int r = 0; r < reps; r++) {
        double t0 = now();
        int rc = kssd_gunzip_mem(z, n, &out, &cap, &len);
        t[r] = now() - t0;
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    qsort(t, reps, sizeof *t, cmp);
    printf("%s: %ld -> %zu bytes, one at a time: best %.0f MB/s, median %.0f MB/s", argv[1], n, len, len / t[0] / 1e6, len / t[reps / 2] / 1e6);
    for (int r = 0; r < reps; r++) {
        const unsigned char *in[2] = {z, z2}; size_t il[2] = {n, n};
        unsigned char **o[2] = {&out, &out2}; size_t *c[2] = {&cap, &cap2}, *l[2] = {&len, &len2}; int rc[2];
        double t0 = now();
        kssd_gunzip_mem2(in, il, o, c, l, rc);
        t[r] = now() - t0;
        if (rc[0] || rc[1] || len != len2 || memcmp(out, out2, len)) { printf("pair rc %d %d\n", rc[0], rc[1]); return 1; }
    }
    qsort(t, reps, sizeof *t, cmp);
    printf("; two in step: best %.0f MB/s, median %.0f MB/s\n", 2 * len / t[0] / 1e6, 2 * len / t[reps / 2] / 1e6);
    return 0;
}
