// Stage A's kernel in isolation (tools/, not shipped): the shipped kernel source, launched over a rotation of databases
// so that every launch streams from HBM, per (layers per thread, coefficients per thread, terms in flight).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inested_hashing_psi_amd/csrc tools/microbench_stage_a.hip -o /tmp/mbsa
#include "../nested_hashing_psi_amd/csrc/kernels_pie.hip"
#include <cstdio>
#include <cstdlib>
using namespace piehip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned __int128 u128;
static const int NBUF = 6;
static size_t DBW;
template <int BPT>
static void run(const char *name, const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus, const u64 *db0, u64 *acc)
{
    dim3 grid(N / TPB, L, K * (b / BPT));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int NREP = 13;
    for (int rep = 0; rep < NREP; rep++) {
        // a different copy of the database every launch (NBUF x 280 MiB >> the 256 MiB infinity cache): in a run() the
        // other kernels' traffic has evicted it by the time stage A comes round again
        const u64 *db = db0 + (size_t)(rep % NBUF) * DBW;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stage_a_mad_kernel<BPT>, grid, dim3(TPB), 0, 0, dc, N, L, K, b, E, idx, minus, db, acc, b, 0u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
        if (rep) sum += ms;
    }
    const float avg = sum / (NREP - 1);
    const double dbb = (double)K * b * E * L * N * 8;
    printf("%-22s b=%2u E=%2u wgs=%5u  best %7.1f avg %7.1f us  db %.2f TB/s\n", name, b, E, grid.x * grid.y * grid.z, best * 1e3, avg * 1e3, dbb / avg / 1e9);
}
int main()
{
    const u32 N = 16384, L = 4, K = 2;
    const size_t LN = (size_t)L * N;
    static DevConsts h;
    const u64 qs[4] = {1152921504606830593ull, 1152921504606748673ull, 1152921504606683137ull, 1152921504606584833ull};
    for (int i = 0; i < 4; i++) {
        h.mod[i].q = qs[i];
        u128 R = ~(u128)0 / qs[i];
        h.mod[i].r0 = (u64)R, h.mod[i].r1 = (u64)(R >> 64);
    }
    DevConsts *dc; CK(hipMalloc(&dc, sizeof h)); CK(hipMemcpy(dc, &h, sizeof h, hipMemcpyHostToDevice));
    u64 *db, *idx, *minus, *acc;
    const size_t dbw = (size_t)K * 40 * 14 * LN, idw = (size_t)K * 40 * 2 * LN;
    DBW = dbw;
    CK(hipMalloc(&db, NBUF * dbw * 8)); CK(hipMalloc(&idx, idw * 8)); CK(hipMalloc(&minus, 2 * LN * 8)); CK(hipMalloc(&acc, (size_t)40 * K * 2 * LN * 8));
    CK(hipMemset(db, 1, NBUF * dbw * 8)); CK(hipMemset(idx, 2, idw * 8)); CK(hipMemset(minus, 3, 2 * LN * 8));
#define RUN(B, b, E) run<B>(#B " layers per thread", dc, N, L, K, b, E, idx, minus, db, acc)
    RUN(7, 14, 14);
    RUN(2, 14, 14);
    RUN(7, 7, 14);
    RUN(5, 5, 14);
    RUN(4, 4, 14);
    RUN(2, 2, 14);
    RUN(5, 40, 14);
    RUN(4, 40, 14);
    RUN(5, 5, 40);
    return 0;
}
