// What bandwidth does stage A's access pattern reach on gfx950 without its arithmetic?  (tools/, not shipped.)
// Each thread walks E terms; per term it loads 2 index words and BPT database words (8 or 16 bytes per lane) from
// streams LN / bin_stride apart, exactly as stage_a_mad_kernel does, and folds them into a checksum with MADS dependent
// 64-bit multiply-adds per database word.   hipcc --offload-arch=gfx950 -O3 tools/microbench_stream.hip -o /tmp/mbs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
struct alignas(16) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BPT, int CPT, int DEPTH, int MADS, int TPB>
__global__ void __launch_bounds__(TPB) stream_kernel(u32 N, u32 L, u32 b, u32 E, const u64 *__restrict__ idx, const u64 *__restrict__ db, u64 *out)
{
    const u32 n = CPT * (blockIdx.x * TPB + threadIdx.x);
    const u32 l = blockIdx.y;
    const u32 groups = b / BPT;
    const u32 h = blockIdx.z / groups, beta0 = (blockIdx.z % groups) * BPT;
    const size_t LN = (size_t)L * N;
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + n;
    const u64 *pd = db + (((size_t)h * b + beta0) * E) * LN + (size_t)l * N + n;
    const size_t bin_stride = (size_t)E * LN;
    u64 acc[BPT][CPT];
    for (int t = 0; t < BPT; t++) for (int e = 0; e < CPT; e++) acc[t][e] = 0;
    u64 qi[DEPTH][2][CPT], qd[DEPTH][BPT][CPT];
    auto load = [&](u32 j, u64 (&vi)[2][CPT], u64 (&vd)[BPT][CPT]) {
        if (CPT == 2) {
            const u64x2 i0 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN);
            const u64x2 i1 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN + LN);
            vi[0][0] = i0.x, vi[0][CPT - 1] = i0.y, vi[1][0] = i1.x, vi[1][CPT - 1] = i1.y;
#pragma unroll
            for (int t = 0; t < BPT; t++) {
                const u64x2 d = *reinterpret_cast<const u64x2 *>(pd + (size_t)t * bin_stride + (size_t)j * LN);
                vd[t][0] = d.x, vd[t][CPT - 1] = d.y;
            }
        } else {
            vi[0][0] = pi[(size_t)j * 2 * LN];
            vi[1][0] = pi[(size_t)j * 2 * LN + LN];
#pragma unroll
            for (int t = 0; t < BPT; t++) vd[t][0] = pd[(size_t)t * bin_stride + (size_t)j * LN];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if ((u32)d < E) load(d, qi[d], qd[d]);
    for (u32 j = 0; j < E; j++) {
        u64 iv[2][CPT], dv[BPT][CPT];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                iv[c][e] = qi[0][c][e];
#pragma unroll
                for (int d = 0; d + 1 < DEPTH; d++) qi[d][c][e] = qi[d + 1][c][e];
            }
#pragma unroll
        for (int t = 0; t < BPT; t++)
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                dv[t][e] = qd[0][t][e];
#pragma unroll
                for (int d = 0; d + 1 < DEPTH; d++) qd[d][t][e] = qd[d + 1][t][e];
            }
        if (j + DEPTH < E) load(j + DEPTH, qi[DEPTH - 1], qd[DEPTH - 1]);
#pragma unroll
        for (int t = 0; t < BPT; t++)
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                u64 v = dv[t][e];
#pragma unroll
                for (int k = 0; k < MADS; k++) v = (u64)(u32)v * (u32)(iv[k & 1][e]) + (v >> 32);
                acc[t][e] += v ^ iv[0][e] ^ iv[1][e];
            }
    }
    u64 s = 0;
    for (int t = 0; t < BPT; t++) for (int e = 0; e < CPT; e++) s += acc[t][e];
    if (s == 0x123456789abcdefull) out[0] = s;
}

// plain grid-stride read of the same bytes (the device's streaming ceiling for reads)
__global__ void __launch_bounds__(256) plain_kernel(const u64x2 *__restrict__ p, size_t n, u64 *out)
{
    u64 s = 0;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u64x2 v = p[i];
        s += v.x ^ v.y;
    }
    if (s == 0x123456789abcdefull) out[0] = s;
}

__global__ void __launch_bounds__(256) plain_write_kernel(u64x2 *p, size_t n)
{
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = u64x2{i, i};
}
__global__ void __launch_bounds__(256) plain_copy_kernel(const u64x2 *__restrict__ p, u64x2 *__restrict__ q, size_t n)
{
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
static u64 *d_idx, *d_db, *d_out;
static u32 N = 16384, L = 4, K = 2;

template <int BPT, int CPT, int DEPTH, int MADS, int TPB>
static void run(const char *name, u32 b, u32 E)
{
    dim3 grid(N / CPT / TPB, L, K * (b / BPT));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_kernel<BPT, CPT, DEPTH, MADS, TPB>), grid, dim3(TPB), 0, 0, N, L, b, E, d_idx, d_db, d_out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double dbb = (double)K * b * E * L * N * 8, idb = (double)K * E * 2 * L * N * 8 * (b / BPT);
    printf("%-34s b=%2u E=%2u wgs=%5u  %7.1f us  db %.2f TB/s  db+idx %.2f TB/s\n", name, b, E, grid.x * grid.y * grid.z, best * 1e3,
           dbb / best / 1e9, (dbb + idb) / best / 1e9);
}
#define RUN(BPT, CPT, DEPTH, MADS, TPB, b, E) run<BPT, CPT, DEPTH, MADS, TPB>("bpt" #BPT " cpt" #CPT " depth" #DEPTH " mads" #MADS " tpb" #TPB, b, E)

int main()
{
    const size_t LN = (size_t)L * N;
    const size_t dbw = (size_t)K * 40 * 14 * LN, idw = (size_t)K * 40 * 2 * LN;  // room for b*E up to 560
    CK(hipMalloc(&d_db, dbw * 8)); CK(hipMalloc(&d_idx, idw * 8)); CK(hipMalloc(&d_out, 8));
    CK(hipMemset(d_db, 1, dbw * 8)); CK(hipMemset(d_idx, 2, idw * 8));
    {
        // streaming ceilings: the same 196 MiB every launch (it stays in the 256 MiB infinity cache) against a rotation over
        // six buffers (1.2 GiB: every launch reads from HBM), and a plain write / copy of the same size
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t n = (size_t)K * 14 * 14 * LN / 2;  // 16-byte words
        u64x2 *big; CK(hipMalloc(&big, 7 * n * 16)); CK(hipMemset(big, 1, 7 * n * 16));
        for (int mode = 0; mode < 4; mode++) {
            float sum = 0;
            const int NREP = 13;
            for (int rep = 0; rep < NREP; rep++) {
                const u64x2 *src = big + (mode == 0 ? 0 : (size_t)(rep % 6) * n);
                CK(hipEventRecord(e0, 0));
                if (mode <= 1) hipLaunchKernelGGL(plain_kernel, dim3(4096), dim3(256), 0, 0, src, n, d_out);
                else if (mode == 2) hipLaunchKernelGGL(plain_write_kernel, dim3(4096), dim3(256), 0, 0, (u64x2 *)src, n);
                else hipLaunchKernelGGL(plain_copy_kernel, dim3(4096), dim3(256), 0, 0, src, big + 6 * n, n / 2);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) sum += ms;
            }
            const float avg = sum / (NREP - 1);
            const char *names[4] = {"read, same buffer", "read, rotating buffers", "write, rotating buffers", "copy (half size each way), rotating"};
            printf("plain %-38s %7.1f us  %.2f TB/s\n", names[mode], avg * 1e3, n * 16.0 / avg / 1e9);
        }
        CK(hipFree(big));
    }
    // stage A's access pattern without its arithmetic (the 196 MiB database stays in the infinity cache here)
    RUN(7, 1, 2, 0, 256, 14, 14);
    RUN(7, 1, 4, 0, 256, 14, 14);
    RUN(7, 2, 1, 0, 256, 14, 14);
    RUN(7, 1, 4, 16, 256, 14, 14);
    RUN(7, 1, 4, 0, 256, 7, 14);
    RUN(5, 1, 4, 0, 256, 5, 14);
    return 0;
}
