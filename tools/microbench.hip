// microbench.hip -- integer ALU throughput probes for gfx950 (which 64-bit modmul formulation is
// cheapest?).  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define ITER 4096
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
// Shoup multiply, quotient estimate from 3 partial products: result in [0, 4q)
__device__ __forceinline__ u64 shoup4(u64 b, u64 w, u64 ws, u64 nq)
{
    const u32 bl = (u32)b, bh = (u32)(b >> 32), wl = (u32)w, wh = (u32)(w >> 32), sl = (u32)ws, sh = (u32)(ws >> 32);
    const u32 nql = (u32)nq, nqh = (u32)(nq >> 32);
    const u64 m1 = mul_u(bl, sh);          // < 2^63  (sh < 2^31: ws is the 63-bit Shoup constant)
    const u64 cr = mad_u(bh, sl, m1);      // both cross terms, < 2^64 for b < 2^63
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
__global__ void __launch_bounds__(512) probe(u64 *out, u64 seed, u64 q, u64 w, u64 wsh)
{
    u64 x[CH];
    for (int c = 0; c < CH; c++) x[c] = seed * (threadIdx.x + 1 + c * 977) + blockIdx.x;
    u32 lo32 = (u32)seed | 1, hi32 = (u32)(seed >> 13) | 1;
    double dq = (double)(q >> 12), dinv = 1.0 / dq;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) {  // 32x32->64 mad (v_mad_u64_u32)
                x[c] = (u64)(u32)x[c] * lo32 + x[c];
            } else if (OP == 1) {  // v_mul_lo_u32
                x[c] = (u32)((u32)x[c] * lo32 + hi32);
            } else if (OP == 2) {  // v_mul_hi_u32
                x[c] = __umulhi((u32)x[c], lo32) + hi32;
            } else if (OP == 3) {  // 64-bit mulhi
                x[c] = __umul64hi(x[c], wsh) + w;
            } else if (OP == 4) {  // 64-bit mullo
                x[c] = x[c] * w + wsh;
            } else if (OP == 5) {  // Shoup lazy modmul
                x[c] = x[c] * w - __umul64hi(x[c], wsh) * q;
            } else if (OP == 6) {  // Harvey butterfly (pairs of chains)
                if (c & 1) {
                    u64 u = x[c - 1], v = x[c];
                    u = u >= 2 * q ? u - 2 * q : u;
                    u64 t = v * w - __umul64hi(v, wsh) * q;
                    x[c - 1] = u + t;
                    x[c] = u - t + 2 * q;
                }
            } else if (OP == 12) {  // asm Shoup [0,4q)
                x[c] = shoup4(x[c] >> 1, w, wsh >> 1, 0 - q);
            } else if (OP == 13) {  // asm butterfly, [0,8q) invariant
                if (c & 1) {
                    u64 u = x[c - 1], v = x[c];
                    const u64 q4 = 4 * q;
                    u = u >= q4 ? u - q4 : u;
                    u64 t = shoup4(v >> 1, w, wsh >> 1, 0 - q);
                    x[c - 1] = u + t;
                    x[c] = u - t + q4;
                }
            } else if (OP == 14) {  // asm butterfly, sign-select conditional subtract
                if (c & 1) {
                    u64 u = x[c - 1], v = x[c];
                    const u64 q4 = 4 * q;
                    const u64 tt = u - q4;
                    u = (long long)tt < 0 ? u : tt;
                    u64 t = shoup4(v >> 1, w, wsh >> 1, 0 - q);
                    x[c - 1] = u + t;
                    x[c] = u - t + q4;
                }
            } else if (OP == 7) {  // 64-bit add
                x[c] = x[c] + w + (x[c] >> 7);
            } else if (OP == 8) {  // f64 fma
                double d = __longlong_as_double(x[c]);
                d = fma(d, dinv, dq);
                x[c] = __double_as_longlong(d);
            } else if (OP == 9) {  // full 128-bit product accumulate (mac128)
                u64 lo = x[c] * w, hi = __umul64hi(x[c], w);
                x[c] = lo + hi;
            } else if (OP == 10) {  // v_mul_u32_u24
                x[c] = __umul24((u32)x[c], lo32) + hi32;
            } else if (OP == 11) {  // Solinas-style: q = 2^60 - delta, delta < 2^24: reduce 128-bit product without w'
                u64 lo = x[c] * w, hi = __umul64hi(x[c], w);
                u64 delta = (1ULL << 60) - q;
                u64 zh = (hi << 4) | (lo >> 60), zl = lo & ((1ULL << 60) - 1);  // z = zh*2^60 + zl
                // zh*delta (zh < 2^62, delta < 2^24) -> up to 86 bits
                u64 p_lo = zh * delta, p_hi = __umul64hi(zh, delta);
                u64 s = zl + (p_lo & ((1ULL << 60) - 1));
                u64 top = (p_hi << 4) | (p_lo >> 60);
                x[c] = s + top * delta;
            }
        }
    }
    u64 r = 0;
    for (int c = 0; c < CH; c++) r ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static int g_blocks_per_cu = 8, g_threads = 256;
template <int OP>
static void run(const char *name, double ops_per_iter_chain, u64 *d)
{
    const int blocks = 256 * g_blocks_per_cu;
    u64 q = (1ULL << 60) - 33 * 32768 + 1, w = 0x0123456789abcdefULL % q, wsh = (u64)(((unsigned __int128)w << 64) / q);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(g_threads), 0, 0, d, 0x9E3779B97F4A7C15ULL, q, w, wsh);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(g_threads), 0, 0, d, 0x9E3779B97F4A7C15ULL + r, q, w, wsh);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double total = 5.0 * blocks * g_threads * (double)ITER * CH * ops_per_iter_chain;
    double gops = total / (ms * 1e-3) / 1e9;
    // lanes per clock per CU at 2.4 GHz (256 CUs)
    double per_cu_clk = gops * 1e9 / 256 / 2.4e9;
    printf("%-28s %8.3f ms  %10.1f Gop/s  %7.2f lane-ops/clk/CU  (%.1f cyc per wave64 op per SIMD)\n", name, ms / 5, gops,
           per_cu_clk, 64.0 * 4 / per_cu_clk);
}

int main(int argc, char **argv)
{
    if (argc > 2) { g_blocks_per_cu = atoi(argv[1]); g_threads = atoi(argv[2]); }
    printf("blocks/CU %d threads %d -> %d waves/SIMD\n", g_blocks_per_cu, g_threads, g_blocks_per_cu * g_threads / 256);
    u64 *d;
    hipMalloc((void **)&d, 256 * 8 * 512 * 8);
    run<0>("mad_u64_u32", 1, d);
    run<1>("mul_lo_u32", 1, d);
    run<2>("mul_hi_u32", 1, d);
    run<10>("mul_u32_u24", 1, d);
    run<3>("umul64hi(+add)", 1, d);
    run<4>("mul64lo(+add)", 1, d);
    run<9>("mul128 (lo+hi)", 1, d);
    run<5>("shoup_lazy modmul", 1, d);
    run<6>("harvey butterfly", 0.5, d);
    run<11>("solinas modmul", 1, d);
    run<12>("asm shoup4 (9 mad)", 1, d);
    run<13>("asm butterfly 8q", 0.5, d);
    run<14>("asm butterfly 8q signsel", 0.5, d);
    run<7>("add64 x2 + shift", 1, d);
    run<8>("fma_f64", 1, d);
    hipFree(d);
    return 0;
}
