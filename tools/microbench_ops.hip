// microbench_ops.hip -- issue cost of the instructions the NTT butterfly block uses (gfx950), independent streams.
//   hipcc --offload-arch=gfx950 -O3 -o microbench_ops tools/microbench_ops.hip && ./microbench_ops [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 2048

#define KERNEL(NAME, BODY)                                                                      \
    __global__ void __launch_bounds__(256) NAME(u64 *out)                                       \
    {                                                                                           \
        u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = 5, a5 = 6, a6 = 7, a7 = 9; \
        unsigned b = blockIdx.x + 3;                                                            \
        for (int it = 0; it < ITER; it++) {                                                     \
            asm volatile(REP16(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(b) : "vcc"); \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }
#define KERNEL32(NAME, BODY)                                                                    \
    __global__ void __launch_bounds__(256) NAME(u64 *out)                                       \
    {                                                                                           \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = 5, a5 = 6, a6 = 7, a7 = 9; \
        unsigned b = blockIdx.x + 3;                                                            \
        for (int it = 0; it < ITER; it++) {                                                     \
            asm volatile(REP16(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(b) : "vcc"); \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }
// 8 instructions per BODY, each on its own accumulator
KERNEL(k_mad, "v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n v_mad_u64_u32 %2, vcc, %8, %8, %2\n v_mad_u64_u32 %3, vcc, %8, %8, %3\n"
              "v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7\n")
KERNEL(k_mad_s, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n"
                       "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n")
KERNEL(k_lshr_b64, "v_lshrrev_b64 %0, 31, %0\n v_lshrrev_b64 %1, 31, %1\n v_lshrrev_b64 %2, 31, %2\n v_lshrrev_b64 %3, 31, %3\n"
                   "v_lshrrev_b64 %4, 31, %4\n v_lshrrev_b64 %5, 31, %5\n v_lshrrev_b64 %6, 31, %6\n v_lshrrev_b64 %7, 31, %7\n")
// 32-bit ops act on the low halves (sub-register of a pair cannot be named: use the pair name = its low register)
KERNEL32(k_bfi, "v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
              "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
                  "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n")
KERNEL32(k_not, "v_not_b32 %0, %1\n v_not_b32 %1, %2\n v_not_b32 %2, %3\n v_not_b32 %3, %4\n v_not_b32 %4, %5\n v_not_b32 %5, %6\n v_not_b32 %6, %7\n v_not_b32 %7, %0\n")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 31, %1\n v_ashrrev_i32 %1, 31, %2\n v_ashrrev_i32 %2, 31, %3\n v_ashrrev_i32 %3, 31, %4\n"
               "v_ashrrev_i32 %4, 31, %5\n v_ashrrev_i32 %5, 31, %6\n v_ashrrev_i32 %6, 31, %7\n v_ashrrev_i32 %7, 31, %0\n")
KERNEL32(k_addc, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
               "v_add_co_u32 %4, vcc, %4, %5\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_add_co_u32 %6, vcc, %6, %7\n v_addc_co_u32 %7, vcc, %7, %0, vcc\n")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                 "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                 "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n"
                      "v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n")
KERNEL32(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                     "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n")
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %3, %3, %4, %5\n"
                  "v_fma_f64 %4, %4, %5, %6\n v_fma_f64 %5, %5, %6, %7\n v_fma_f64 %6, %6, %7, %0\n v_fma_f64 %7, %7, %0, %1\n")
KERNEL(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %8, %8, %0\n v_mad_i64_i32 %1, vcc, %8, %8, %1\n v_mad_i64_i32 %2, vcc, %8, %8, %2\n v_mad_i64_i32 %3, vcc, %8, %8, %3\n"
                      "v_mad_i64_i32 %4, vcc, %8, %8, %4\n v_mad_i64_i32 %5, vcc, %8, %8, %5\n v_mad_i64_i32 %6, vcc, %8, %8, %6\n v_mad_i64_i32 %7, vcc, %8, %8, %7\n")

template <class K>
static void run(const char *name, K kern, u64 *d, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks_per_cu waves per SIMD (256 threads = 4 waves = 1 per SIMD per block)
    const double winstr = 5.0 * blocks_per_cu * (double)ITER * 16 * 8;
    const double cyc = ms * 1e-3 * 2.4e9 / winstr;
    printf("%-18s %8.3f ms   %5.2f cycles per wave64 instruction per SIMD (at 2.4 GHz nominal)\n", name, ms / 5, cyc);
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    printf("%d waves per SIMD\n", wps);
    u64 *d;
    hipMalloc((void **)&d, 256 * 8 * 256 * 8);
    run("v_mad_u64_u32", k_mad, d, wps);
    run("v_mad_u64_u32 sgpr", k_mad_s, d, wps);
    run("v_mad_i64_i32", k_mad_i64_i32, d, wps);
    run("v_lshl_add_u64", k_lshl_add_u64, d, wps);
    run("v_lshrrev_b64", k_lshr_b64, d, wps);
    run("v_bfi_b32", k_bfi, d, wps);
    run("v_add_u32", k_add_u32, d, wps);
    run("v_not_b32", k_not, d, wps);
    run("v_ashrrev_i32", k_ashr, d, wps);
    run("add_co+addc pair/2", k_addc, d, wps);
    run("v_mul_lo_u32", k_mul_lo, d, wps);
    run("v_mul_hi_u32", k_mul_hi, d, wps);
    run("v_mad_u32_u24", k_mad_u32_u24, d, wps);
    run("v_mul_hi_u32_u24", k_mul_hi_u24, d, wps);
    run("v_fma_f64", k_fma_f64, d, wps);
    return 0;
}
