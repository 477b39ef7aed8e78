// Does stage A's access pattern cost HBM bandwidth?  Reads the same 7 x 14 plaintext limbs per (h, l) either in the
// library's layout ([bin][j][L][N]: 98 streams 512 KiB apart) or as coefficient tiles ([tile][bin][j][256 coefficients]:
// one contiguous 196 KiB run per block), with the same arithmetic intensity (a mad per word).  GB/s of each.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_layout.hip -o exp/stream_layout && exp/stream_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
static const int N = 16384, L = 4, E = 14, B = 7, K = 2;

__global__ void __launch_bounds__(256) strided(const u64 *__restrict__ db, u64 *__restrict__ out)
{
    const unsigned n = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y, h = blockIdx.z;
    const size_t LN = (size_t)L * N;
    const u64 *p = db + ((size_t)h * B * E) * LN + (size_t)l * N + n;
    u64 acc[B] = {0};
    for (int j = 0; j < E; j++)
#pragma unroll
        for (int t = 0; t < B; t++) acc[t] += p[((size_t)t * E + j) * LN] * (u64)(j + 3);
#pragma unroll
    for (int t = 0; t < B; t++) out[(((size_t)h * B + t) * L + l) * N + n] = acc[t];
}
__global__ void __launch_bounds__(256) tiled(const u64 *__restrict__ db, u64 *__restrict__ out)
{
    const unsigned tile = blockIdx.x, l = blockIdx.y, h = blockIdx.z, i = threadIdx.x;
    // [h][l][tile][bin][j][256]
    const u64 *p = db + ((((size_t)h * L + l) * (N / 256) + tile) * B * E) * 256 + i;
    u64 acc[B] = {0};
    for (int j = 0; j < E; j++)
#pragma unroll
        for (int t = 0; t < B; t++) acc[t] += p[((size_t)t * E + j) * 256] * (u64)(j + 3);
#pragma unroll
    for (int t = 0; t < B; t++) out[(((size_t)h * B + t) * L + l) * N + tile * 256 + i] = acc[t];
}
int main()
{
    const size_t words = (size_t)K * B * E * L * N;
    u64 *db, *out;
    hipMalloc(&db, words * 8);
    hipMalloc(&out, (size_t)K * B * L * N * 8);
    hipMemset(db, 1, words * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(N / 256, L, K);
    for (int mode = 0; mode < 2; mode++) {
        for (int w = 0; w < 3; w++) {
            if (mode) hipLaunchKernelGGL(tiled, grid, dim3(256), 0, 0, db, out);
            else hipLaunchKernelGGL(strided, grid, dim3(256), 0, 0, db, out);
        }
        hipEventRecord(e0);
        const int it = 50;
        for (int w = 0; w < it; w++) {
            if (mode) hipLaunchKernelGGL(tiled, grid, dim3(256), 0, 0, db, out);
            else hipLaunchKernelGGL(strided, grid, dim3(256), 0, 0, db, out);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.0f GB/s read\n", mode ? "tiled  " : "strided", ms / it * 1e3, words * 8.0 / (ms / it * 1e-3) / 1e9);
    }
    return 0;
}
