// Stage A's query-batch kernel in isolation (tools/, not shipped): the shipped kernel source over a rotation of databases
// (every launch streams the database from HBM), per (queries, layers per thread, terms in flight), with the
// queries' index matrices distinct or all the same array (how much of the time is index-matrix traffic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inested_hashing_psi_amd/csrc tools/microbench_stage_a_batch.hip -o build_lab/mbsab
#include "../nested_hashing_psi_amd/csrc/kernels_pie.hip"
#include <cstdio>
#include <cstdlib>
using namespace piehip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned __int128 u128;
static const int NBUF = 6;
static size_t DBW;
static const u32 N = 16384, L = 4, K = 2, B = 14, E = 14;
static u64 *g_idx[8], *g_minus, *g_db, *g_acc;
static DevConsts *g_dc;

template <int BPT, int Q, int DEPTH>
static void run(bool same_idx, u32 Bl = B)   // Bl: bin layers of this launch (<= B; the last group may be ragged)
{
    StageAQueries qs = {};
    for (int q = 0; q < Q; q++) qs.idx[q] = g_idx[same_idx ? 0 : q], qs.minus[q] = g_minus;
    const u32 nx = N / TPB, tiles = nx * L * K;
    const dim3 grid = stage_a_grid(nx, L, K, (Bl + BPT - 1) / BPT);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float sum = 0;
    const int NREP = 13;
    for (int rep = 0; rep < NREP; rep++) {
        const u64 *db = g_db + (size_t)(rep % NBUF) * DBW;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stage_a_mad_batch_kernel<BPT, Q, DEPTH>), grid, dim3(TPB), 0, 0, g_dc, N, L, K, B, E, qs, db, g_acc, B, 0u, (u32)Q, 0u, tiles, StageAXOut{});
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) sum += ms;
    }
    const float avg = sum / (NREP - 1) * 1e3f;
    const double bytes = 8.0 * L * N * ((double)K * Bl * E + Q * ((double)K * E * 2 + 2 + (double)Bl * K * 2));
    printf("layers=%u Q=%d BPT=%d depth=%d %s: %7.1f us per launch, %6.1f us per query, compulsory %.0f MiB -> %.2f TB/s\n", Bl, Q, BPT, DEPTH,
           same_idx ? "same idx    " : "distinct idx", avg, avg / Q, bytes / 1048576.0, bytes / avg / 1e6);
}
int main()
{
    const size_t LN = (size_t)L * N;
    static DevConsts h;
    const u64 qs[4] = {1152921504606830593ull, 1152921504606748673ull, 1152921504606683137ull, 1152921504606584833ull};
    for (int i = 0; i < 4; i++) {
        h.mod[i].q = qs[i];
        u128 R = ~(u128)0 / qs[i];
        h.mod[i].r0 = (u64)R, h.mod[i].r1 = (u64)(R >> 64);
    }
    CK(hipMalloc(&g_dc, sizeof h)); CK(hipMemcpy(g_dc, &h, sizeof h, hipMemcpyHostToDevice));
    const size_t dbw = (size_t)K * B * E * LN, idw = (size_t)K * E * 2 * LN;
    DBW = dbw;
    CK(hipMalloc(&g_db, NBUF * dbw * 8)); CK(hipMalloc(&g_minus, 2 * LN * 8)); CK(hipMalloc(&g_acc, (size_t)8 * B * K * 2 * LN * 8));
    CK(hipMemset(g_db, 1, NBUF * dbw * 8)); CK(hipMemset(g_minus, 3, 2 * LN * 8));
    for (int q = 0; q < 8; q++) { CK(hipMalloc(&g_idx[q], idw * 8)); CK(hipMemset(g_idx[q], 2 + q, idw * 8)); }
    for (int s = 0; s < 2; s++) {
        run<2, 3, 3>(s);          // r03-r04's tiling for three queries: seven groups of two layers
        run<3, 3, 2>(s);          // r05: five groups of three, the last one ragged (shipped)
        run<4, 3, 2>(s);          // four groups of four, the last one ragged (204 VGPRs: two waves per SIMD)
        run<2, 3, 3>(s, 12);
        run<3, 3, 2>(s, 12);
        run<3, 3, 3>(s, 12);
        run<4, 3, 2>(s, 12);
        run<2, 3, 3>(s, 2);       // a remainder launch of two layers: all latency
        run<4, 2, 3>(s);          // two queries: four groups of four, ragged (shipped)
        run<4, 2, 3>(s, 12);
        run<2, 4, 2>(s);          // four queries (shipped)
    }
    return 0;
}
