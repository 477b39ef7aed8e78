// microbench_fp64.hip -- what would a ~50-bit butterfly on the FP64 pipe cost against the 60-bit integer one?
// Probes (cycles per wave64 butterfly per SIMD, at 2 and 8 waves per SIMD):
//   0: integer Harvey butterfly with the 10-mad Shoup product (the production one, kernels_ntt_fast.hip)
//   1: FP64 butterfly, p < 2^50: T = b w mod p with (h, l) = two-product(b, w), q = rint(b w'), r = fma(-q, p, h) + l;
//      outputs kept in (-p, p) with one conditional correction each
// Stand-alone: hipcc --offload-arch=gfx950 -O3 -o exp/microbench_fp64 tools/microbench_fp64.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define ITER 2048
#define CH 8

__device__ __forceinline__ u64 mad_u(u32 a, u32 b, u64 c)
{
    u64 d, sc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sc) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u64 mul_u(u32 a, u32 b)
{
    u64 d, sc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(sc) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u64 shoup4(u64 b, u64 w, u64 ws, u64 nq)
{
    const u32 bl = (u32)b, bh = (u32)(b >> 32), wl = (u32)w, wh = (u32)(w >> 32), sl = (u32)ws, sh = (u32)(ws >> 32);
    const u32 nql = (u32)nq, nqh = (u32)(nq >> 32);
    const u64 m1 = mul_u(bl, sh);
    const u64 cr = mad_u(bh, sl, m1);
    const u64 top = mul_u(bh, sh);
    const u64 qe = (top << 1) + (cr >> 31);
    u64 acc = mul_u((u32)qe, nql);
    acc = mad_u(bl, wl, acc);
    u64 c = mul_u((u32)qe, nqh);
    c = mad_u((u32)(qe >> 32), nql, c);
    c = mad_u(bl, wh, c);
    c = mad_u(bh, wl, c);
    return acc + ((u64)(u32)c << 32);
}

template <int OP>
__global__ void __launch_bounds__(512) probe(u64 *out, u64 seed, u64 q, u64 w, u64 ws63, double p, double wd, double wdp)
{
    if (OP == 0) {
        u64 x[CH];
        for (int c = 0; c < CH; c++) x[c] = (seed * (threadIdx.x + 1 + c * 977) + blockIdx.x) % q;
        const u64 nq = 0 - q, q4 = 4 * q;
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int c = 0; c < CH; c += 2) {
                u64 a = x[c], b = x[c + 1];
                const u64 u = a >= q4 ? a - q4 : a;
                const u64 v = shoup4(b, w, ws63, nq);
                x[c] = u + v;
                x[c + 1] = u - v + q4;
            }
        }
        u64 s = 0;
        for (int c = 0; c < CH; c++) s += x[c];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        double x[CH];
        for (int c = 0; c < CH; c++) x[c] = (double)((seed * (threadIdx.x + 1 + c * 977) + blockIdx.x) % (u64)p);
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int c = 0; c < CH; c += 2) {
                const double a = x[c], b = x[c + 1];
                const double h = b * wd;
                const double l = fma(b, wd, -h);
                const double qf = rint(b * wdp);
                double t = fma(-qf, p, h) + l;            // in (-p, p) (+ rounding slack)
                double s = a + t, d = a - t;              // in (-2p, 2p)
                s = s >= p ? s - p : (s <= -p ? s + p : s);
                d = d >= p ? d - p : (d <= -p ? d + p : d);
                x[c] = s;
                x[c + 1] = d;
            }
        }
        double s = 0;
        for (int c = 0; c < CH; c++) s += x[c];
        out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(long long)s;
    }
}

template <int OP>
static void run(const char *name, int waves_per_simd)
{
    const int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves per block, one per SIMD -> blocks per CU = waves per SIMD
    u64 *out;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    const u64 q = 1152921504606584833ULL, w = 123456789012345ULL;
    const u64 ws63 = (u64)(((unsigned __int128)w << 63) / q);
    const double p = 1125899906826241.0;  // a 50-bit prime = 1 mod 2^15 (value only matters as a magnitude here)
    const double wd = 123456789012.0, wdp = wd / p;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u, q, w, ws63, p, wd, wdp);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u, q, w, ws63, p, wd, wdp);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // butterflies per SIMD: waves_per_simd waves x ITER x CH/2; cycles at the nominal 2.4 GHz
    const double bf = (double)waves_per_simd * ITER * (CH / 2);
    printf("%-28s %d waves/SIMD: %7.1f us, %6.1f cycles per wave64 butterfly (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e3, ms * 1e-3 * 2.4e9 / bf);
    hipFree(out);
}
int main()
{
    for (int wps : {2, 4, 8}) {
        run<0>("int 60-bit (10-mad Shoup)", wps);
        run<1>("fp64 50-bit (two-product)", wps);
    }
    return 0;
}
