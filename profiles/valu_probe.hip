// valu_probe -- development tool (GPU box): issue cost of the integer VALU and LDS-crossbar instructions the scan kernel
// is made of, in shader cycles per wave-instruction and SIMD, at 1 and 4 waves per SIMD (s_memtime around an unrolled,
// dependency-free instruction stream).  Answers "is an integer VALU op 2 or 4 cycles per wave64 on gfx950?" -- the
// number the scan kernel's VALU ceiling is computed from.
//   hipcc --offload-arch=gfx950 -O3 profiles/valu_probe.hip -o profiles/valu_probe && profiles/valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITERS 512

template <int OP>
__global__ __launch_bounds__(1024) void probe(uint32_t *out, unsigned long long *ticks, uint32_t seed)
{
    __shared__ uint32_t lds[32768];  // 128 KiB
    for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t a[8], b = threadIdx.x * 2654435761u + seed, c = seed ^ 0x5bd1e995u;
    unsigned long long d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = (unsigned long long)b * (i + 5) + seed;
    asm volatile("s_mov_b64 s[10:11], %0" : : "s"((unsigned long long)seed * 0x9E3779B97F4A7C15ull) : "s10", "s11");
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = b * (i + 3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#define ONE(i)                                                                                                         \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                           \
    if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                           \
    if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));                                    \
    if (OP == 3) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[i]));                                                 \
    if (OP == 4) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));                                \
    if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(a[i]) : "v"(b));                                     \
    if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                         \
    if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                                   \
    if (OP == 8) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                                      \
    if (OP == 9) asm volatile("ds_permute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                                       \
    if (OP == 10) asm volatile("ds_read_u8 %0, %1" : "=v"(a[i]) : "v"((b + i * 7919u) & 0x1FFFFu));                      \
    if (OP == 11) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"((b + i * 7919u) & 0x1FFFCu));                     \
    if (OP == 12) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                                      \
    if (OP == 13) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));                                                       \
    if (OP == 14) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));     \
    if (OP == 15) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                                   \
    if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");           \
    if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, 0, 1, s[10:11]" : "=v"(a[i]) : : "s10", "s11");                     \
    if (OP == 18) asm volatile("v_cmp_gt_u32_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(b) : "s10", "s11");                  \
    if (OP == 19) asm volatile("v_cmp_gt_u32_e64 s[10:11], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11"); \
    if (OP == 20) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d[i]) : "v"(b & 31));                                    \
    if (OP == 21) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                          \
    if (OP == 22) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                                  \
    if (OP == 23) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                              \
    if (OP == 24) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));                                                    \
    if (OP == 25) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));                                                  \
    if (OP == 26) asm volatile("v_bfe_i32 %0, %0, 4, 1" : "+v"(a[i]));                                                     \
    if (OP == 27) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                              \
    if (OP == 28) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(a[i]) : "v"(b), "v"(c));                    \
    if (OP == 29) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                                 \
    if (OP == 30) asm volatile("v_readlane_b32 s10, %0, 63" : : "v"(a[i]) : "s10");                                        \
    if (OP == 31) asm volatile("s_nop 0");
        REP8(ONE)
        REP8(ONE)
        if (OP >= 8 && OP <= 11) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); b += a[0] & 0xFF; }
    }
    if (OP >= 8 && OP <= 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= a[i] ^ (uint32_t)d[i] ^ (uint32_t)(d[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static int run(const char *name, int threads)
{
    const int blocks = 256;
    uint32_t *out;
    unsigned long long *ticks;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CK(hipMalloc(&ticks, (size_t)blocks * threads / 64 * 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 12345u + w);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 999u);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * threads / 64);
    CK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto t : h) avg += (double)t;
    avg /= (double)h.size();
    const double n_inst = (double)ITERS * 16;
    const int waves_per_simd = threads / 256;
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9; wall time x an assumed 2.4 GHz is printed beside it
    printf("%-22s waves/SIMD %d  kernel %.3f ms  ticks/wave %.0f  -> %.2f cycles per wave-instruction and SIMD @2.4 GHz (wall), per CU %.2f\n", name,
           waves_per_simd, ms, avg, ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd), ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd * 4));
    hipFree(out);
    hipFree(ticks);
    return 0;
}

int main()
{
    for (int threads : {256, 1024}) {
        run<0>("v_add_u32", threads);
        run<1>("v_and_b32", threads);
        run<2>("v_alignbit_b32", threads);
        run<3>("v_bfe_u32", threads);
        run<4>("v_bfi_b32", threads);
        run<5>("v_lshl_or_b32", threads);
        run<6>("v_mul_lo_u32", threads);
        run<7>("v_cndmask_b32", threads);
        run<12>("v_bcnt_u32_b32", threads);
        run<13>("v_ffbl_b32", threads);
        run<14>("v_mov_b32 dpp wave_shr", threads);
        run<15>("v_mbcnt_lo", threads);
        run<16>("v_cndmask_e64 sgpr", threads);
        run<17>("v_cndmask_e64 0,1,sgpr", threads);
        run<18>("v_cmp_gt_u32_e64 sgpr", threads);
        run<19>("v_cmp+s_nop+v_cndmask", threads);
        run<20>("v_lshlrev_b64", threads);
        run<21>("v_lshl_add_u64", threads);
        run<22>("v_or3_b32", threads);
        run<23>("v_xor_b32", threads);
        run<24>("v_lshrrev_b32", threads);
        run<25>("v_mov_b32", threads);
        run<26>("v_bfe_i32", threads);
        run<27>("v_min_u32", threads);
        run<28>("v_bitop3_b32", threads);
        run<29>("v_perm_b32", threads);
        run<30>("v_readlane_b32", threads);
        run<31>("s_nop 0", threads);
        run<8>("ds_bpermute_b32", threads);
        run<9>("ds_permute_b32", threads);
        run<10>("ds_read_u8 random", threads);
        run<11>("ds_read_b32 random", threads);
    }
    return 0;
}
