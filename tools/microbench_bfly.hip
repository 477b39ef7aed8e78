// microbench_bfly.hip -- the NTT butterfly blocks of ntt16_bfly.inc in a register-only loop: cycles per butterfly per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o microbench_bfly tools/microbench_bfly.hip && ./microbench_bfly [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../nested_hashing_psi_amd/csrc/ntt16_kernel.h"
using namespace piehip;
using namespace piehip::ntt16;
#define ITER 512

template <bool INV, bool SC>
__global__ void __launch_bounds__(256) probe(u64 *out, u64 q, u64x2 twv)
{
    u64 x[16];
    for (int k = 0; k < 16; k++) x[k] = (threadIdx.x * 977 + k * 131 + 5) % q;
    ModC mc;
    mc.nql = (u32)(0 - q), mc.nqh = (u32)((0 - q) >> 32), mc.nq4 = 0 - 4 * q, mc.q4 = 4 * q;
    u64x2 tv = twv;
    if (!SC) tv.x += threadIdx.x & 1;  // per-lane twiddle: VGPR operands
    const Tw t = make_tw(tv);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) bfly<INV, SC>(x[k], x[k + 8], t, mc);
    }
    u64 r = 0;
    for (int k = 0; k < 16; k++) r ^= x[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <bool INV, bool SC>
static void run(const char *name, u64 *d, int wps)
{
    const u64 q = (1ULL << 60) - 33 * 32768 + 1;
    u64x2 tw;
    tw.x = 0x0123456789abcdefULL % q;
    tw.y = (u64)((((unsigned __int128)tw.x) << 63) / q);
    const int blocks = 256 * wps;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((probe<INV, SC>), dim3(blocks), dim3(256), 0, 0, d, q, tw);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((probe<INV, SC>), dim3(blocks), dim3(256), 0, 0, d, q, tw);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wb = 5.0 * wps * (double)ITER * 8;  // wave-butterflies per SIMD
    printf("%-34s %8.3f ms   %6.1f cycles per wave64 butterfly per SIMD (2.4 GHz nominal)\n", name, ms / 5, ms * 1e-3 * 2.4e9 / wb);
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    printf("%d waves per SIMD\n", wps);
    u64 *d;
    hipMalloc((void **)&d, 256 * 8 * 256 * 8);
    run<false, true>("forward, uniform twiddle (SGPR)", d, wps);
    run<false, false>("forward, per-lane twiddle (VGPR)", d, wps);
    run<true, true>("inverse, uniform twiddle (SGPR)", d, wps);
    run<true, false>("inverse, per-lane twiddle (VGPR)", d, wps);
    return 0;
}
