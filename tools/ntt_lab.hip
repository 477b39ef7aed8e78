// ntt_lab.hip -- stand-alone bench + check of the 16-coefficients-per-thread NTT kernel (csrc/ntt16_kernel.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ntt_lab tools/ntt_lab.hip && ./ntt_lab [nitems ...]
// The check runs the same butterfly network on the host with the same tables (twiddles here are arbitrary residues:
// the lab verifies indexing, ranges and hand-offs, the library's tests verify the mathematics).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#ifdef LAB_STAMPS
// cycle stamps of the second slice of block 0, one row per wave: -DLAB_STAMPS
#include <stdint.h>
__device__ unsigned long long g_stamps[16 * 16];
#define NTT16_STAMP(i)                                                                                       \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (blockIdx.x == 0 && item == gridDim.x) {                                                          \
            unsigned long long t_;                                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
            if ((threadIdx.x & 63) == 0) g_stamps[(threadIdx.x >> 6) * 16 + (i)] = t_;                       \
            if ((i) == 0 || (i) == 8) {                                                                      \
                unsigned long long r_;                                                                       \
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_)::"memory");               \
                if ((threadIdx.x & 63) == 0) g_stamps[(threadIdx.x >> 6) * 16 + 9 + ((i) >> 3)] = r_;        \
            }                                                                                                \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#endif
#include "../nested_hashing_psi_amd/csrc/ntt16_kernel.h"

using namespace piehip;
using namespace piehip::ntt16;
// slice geometry under test: -DLAB_LOGNS=14 for the 2^14-coefficient slices (1024 threads, one workgroup per CU)
#ifndef LAB_LOGNS
#define LAB_LOGNS 13
#endif
typedef Geo<LAB_LOGNS> LG;
#define LOGN LG::LOGN
#define NS LG::NS
#define T LG::T
#define LDS_WORDS LG::LDS_WORDS
#define lane_to_std(p) lane_to_std_t((p), (1u << LAB_LOGNS) / 16)
static const unsigned LAB_SLOTS = (LAB_LOGNS == 13 ? 2u : 1u) * 256u;   // resident workgroups on 256 CUs

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static u64 rng_state = 0x9E3779B97F4A7C15ULL;
static u64 rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static u64 mulmod_h(u64 a, u64 b, u64 q) { return (u64)((unsigned __int128)a * b % q); }

int main(int argc, char **argv)
{
#ifndef LAB_NMOD
#define LAB_NMOD 3   // moduli the limbs cycle over: 4 = the run's Q launches (one table per XCD), 9 = its QP launches
#endif
    const u32 s0 = 1, N = NS << s0, nmod = LAB_NMOD;
    u64 qs[LAB_NMOD < 3 ? 3 : LAB_NMOD] = {(1ULL << 60) - 33 * 32768 + 1, (1ULL << 60) - 97 * 32768 + 1, (1ULL << 59) + 5 * 32768 + 1};
    for (u32 m = 3; m < nmod; m++) qs[m] = (1ULL << 60) - (161 + 64 * (u64)m) * 32768 + 1;
    static DevConsts dc;
    memset(&dc, 0, sizeof(dc));
    for (u32 m = 0; m < nmod; m++) dc.mod[m].q = qs[m];
    // natural tables: per modulus [fwd N][inv N] pairs
    std::vector<u64> twp((size_t)nmod * 2 * N * 2), twk;
    for (u32 m = 0; m < nmod; m++)
        for (u32 d = 0; d < 2; d++)
            for (u32 i = 0; i < N; i++) {
                const u64 w = rnd() % qs[m];
                twp[(((size_t)m * 2 + d) * N + i) * 2] = w;
                twp[(((size_t)m * 2 + d) * N + i) * 2 + 1] = (u64)((((unsigned __int128)w) << 63) / qs[m]);
            }
    {
        std::vector<u64> one;
        for (u32 m = 0; m < nmod; m++)
            for (u32 d = 0; d < 2; d++) {
                build_twk_table_t<LAB_LOGNS>(&twp[((size_t)m * 2 + d) * N * 2], s0, one);
                twk.insert(twk.end(), one.begin(), one.end());
            }
    }
    u64 *d_twp, *d_twk;
    DevConsts *d_dc;
    CK(hipMalloc((void **)&d_twp, twp.size() * 8));
    CK(hipMalloc((void **)&d_twk, twk.size() * 8));
    CK(hipMalloc((void **)&d_dc, sizeof(dc)));
    CK(hipMemcpy(d_twp, twp.data(), twp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_twk, twk.data(), twk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dc, &dc, sizeof(dc), hipMemcpyHostToDevice));
    const size_t lds = LDS_WORDS * 8;
    CK(hipFuncSetAttribute((const void *)ntt16_kernel_t<LAB_LOGNS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)ntt16_kernel_t<LAB_LOGNS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (ntt16_kernel_t<LAB_LOGNS, false>), T, lds));
    printf("occupancy: %d blocks/CU forward", occ);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (ntt16_kernel_t<LAB_LOGNS, true>), T, lds));
    printf(", %d inverse\n", occ);

    // ---- correctness on a small batch ------------------------------------------------------------------------------------
    int bad = 0;
    for (int inv = 0; inv < 2; inv++)
        for (int std_in = 0; std_in <= inv; std_in++) {
            const u32 nitems = 12;  // 6 limbs x 2 slices, moduli cycle over 3
            std::vector<u64> h((size_t)nitems * NS), ref;
            for (size_t i = 0; i < h.size(); i++) h[i] = rnd() % qs[((i / NS) >> s0) % nmod];
            ref = h;
            u64 *d;
            CK(hipMalloc((void **)&d, h.size() * 8));
            CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
            Args a;
            memset(&a, 0, sizeof(a));
            a.data = d, a.twp = (const u64x2 *)d_twp, a.twk = (const u64x2 *)d_twk, a.dc = d_dc, a.N = N, a.s0 = s0, a.nitems = nitems;
            a.mod_base = 0, a.mod_count = nmod;
            a.flags = inv ? (F_FOLDED | (std_in ? F_STD_IN : 0)) : 0;
            a.lift_first = ~0u;
            if (inv)
                hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, true>), dim3(5), dim3(T), lds, 0, a);   // 5 blocks: every block loops
            else
                hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, false>), dim3(5), dim3(T), lds, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            CK(hipFree(d));
            for (u32 it = 0; it < nitems; it++) {
                const u32 limb = it >> s0, blk = it & ((1u << s0) - 1), m = limb % nmod;
                const u64 q = qs[m];
                const u64 *tw = &twp[((size_t)m * 2 + inv) * N * 2];
                u64 *x = &ref[(size_t)it * NS];
                std::vector<u64> tmp(NS);
                if (inv && !std_in) {  // the input array is in lane order: bring the reference input to standard order
                    for (u32 p = 0; p < NS; p++) tmp[lane_to_std(p)] = x[p];
                    memcpy(x, tmp.data(), NS * 8);
                }
                if (!inv) {
                    for (u32 s = 0; s < LOGN; s++) {
                        const u32 ml = 1u << s, half = NS >> (s + 1);
                        for (u32 g = 0; g < ml; g++) {
                            const u64 w = tw[(((size_t)ml << s0) + (size_t)blk * ml + g) * 2];
                            for (u32 j = 0; j < half; j++) {
                                u64 &A = x[2 * half * g + j], &B = x[2 * half * g + j + half];
                                const u64 v = mulmod_h(B, w, q), u = A;
                                A = (u + v) % q;
                                B = (u + q - v) % q;
                            }
                        }
                    }
                    for (u32 p = 0; p < NS; p++)
                        if (h[(size_t)it * NS + p] != x[lane_to_std(p)]) {
                            if (bad < 5) printf("fwd mismatch item %u pos %u: %llx vs %llx\n", it, p, (unsigned long long)h[(size_t)it * NS + p], (unsigned long long)x[lane_to_std(p)]);
                            bad++;
                        }
                } else {
                    for (int s = LOGN - 1; s >= 0; s--) {
                        const u32 ml = 1u << s, half = NS >> (s + 1);
                        for (u32 g = 0; g < ml; g++) {
                            const u64 w = tw[(((size_t)ml << s0) + (size_t)blk * ml + g) * 2];
                            for (u32 j = 0; j < half; j++) {
                                u64 &A = x[2 * half * g + j], &B = x[2 * half * g + j + half];
                                const u64 u = A, v = B;
                                A = (u + v) % q;
                                B = mulmod_h((u + q - v) % q, w, q);
                            }
                        }
                    }
                    for (u32 p = 0; p < NS; p++)
                        if (h[(size_t)it * NS + p] % q != x[p] || h[(size_t)it * NS + p] >= 4 * q) {
                            if (bad < 5) printf("inv(std_in=%d) mismatch item %u pos %u\n", std_in, it, p);
                            bad++;
                        }
                }
            }
            printf("%s%s: %s\n", inv ? "inverse" : "forward", inv ? (std_in ? " (standard order in)" : " (lane order in)") : "", bad ? "MISMATCH" : "ok");
        }
#if !defined(LAB_NOLOAD) && !defined(LAB_NOSTORE)
    if (bad) return 1;
#endif

#ifdef LAB_STAMPS
    for (int inv = 0; inv < 2; inv++)
    for (u32 grid : {LAB_SLOTS / 2, LAB_SLOTS}) {
        const u32 nitems = 3 * grid;
        u64 *d;
        CK(hipMalloc((void **)&d, (size_t)nitems * NS * 8));
        CK(hipMemset(d, 1, (size_t)nitems * NS * 8));
        Args a;
        memset(&a, 0, sizeof(a));
        a.data = d, a.twp = (const u64x2 *)d_twp, a.twk = (const u64x2 *)d_twk, a.dc = d_dc, a.N = N, a.s0 = s0, a.nitems = nitems;
        a.mod_base = 0, a.mod_count = nmod;
        a.flags = inv ? F_FOLDED : F_LAZY_OUT;
        a.lift_first = ~0u;
        for (int rep = 0; rep < 400; rep++) {
            if (inv) hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, true>), dim3(grid), dim3(T), lds, 0, a);
            else hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, false>), dim3(grid), dim3(T), lds, 0, a);
        }
        CK(hipDeviceSynchronize());
        unsigned long long st[16 * 16];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
        static const char *names[2][9] = {{"load", "pass1", "barA", "ldsW", "barB", "rd+pass2", "pass3", "pass4", "store"},
                                          {"-", "ld+pass4'", "barA", "ldsW", "pass3'", "pass2'", "barB", "pass1'", "store"}};
        printf("%s, %u blocks (%.1f per CU): cycles per phase of the second slice (s_memtime ticks)\n", inv ? "inverse (lane order in)" : "forward", grid, grid / 256.0);
        for (int w = 0; w < (int)LG::W; w++) {
            printf("  wave %d:", w);
            for (int i = 1; i <= 8; i++) printf(" %s %llu", names[inv][i], st[w * 16 + i] - st[w * 16 + i - 1]);
            printf("  | total %llu cycles in %.2f us -> %.2f GHz\n", st[w * 16 + 8] - st[w * 16 + 0], (st[w * 16 + 10] - st[w * 16 + 9]) / 100.0,
                   (double)(st[w * 16 + 8] - st[w * 16 + 0]) / ((st[w * 16 + 10] - st[w * 16 + 9]) * 10.0));
        }
        CK(hipFree(d));
    }
    return 0;
#endif
    // ---- timing --------------------------------------------------------------------------------------------------------------
    std::vector<u32> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back((u32)atoi(argv[i]));
    if (sizes.empty()) sizes = {224, 448, 512, 756, 784, 1024, 2048, 4096};
    // inv = 2: the run()'s first inverse launch -- standard order in (stage A's accumulators), the lane-ordered copy of the X
    // operand's limbs written on the side (K = 2 ciphertexts of 2 x 3 limbs per row; rows of 12 limbs)
    // LAB_ROT=n: the launches cycle through n buffer sets (beyond the 256 MiB of the memory-side cache: every launch streams from HBM,
    // as in run(), instead of finding the previous launch's output)
    const int nrot = getenv("LAB_ROT") ? atoi(getenv("LAB_ROT")) : 1;
    // inv = 2: the run()'s first inverse launch -- standard order in (stage A's accumulators), the lane-ordered copy of the X
    // operand's limbs written on the side (K = 2 ciphertexts of 2 x 3 limbs per row; rows of 12 limbs)
    // inv = 3: the run()'s last forward launch -- a third plain items (d0, d1), two thirds key-switch digits lifted in the load phase
    for (int inv = 0; inv < 4; inv++)
        for (u32 nitems : sizes) {
            std::vector<u64 *> d(nrot), d_copy(nrot, nullptr), d_src(nrot, nullptr);
            Args a;
            memset(&a, 0, sizeof(a));
            a.twp = (const u64x2 *)d_twp, a.twk = (const u64x2 *)d_twk, a.dc = d_dc, a.N = N, a.s0 = s0, a.nitems = nitems;
            a.mod_base = 0, a.mod_count = nmod;
            a.flags = (inv == 1 || inv == 2) ? F_FOLDED : F_LAZY_OUT;
            a.lift_first = ~0u;
            const size_t rows = ((nitems >> s0) + 4 * nmod - 1) / (4 * nmod);
            u32 nlift_rows = 0;
            if (inv == 2) {
                a.flags |= F_STD_IN;
                a.copy_K = 2, a.copy_L = nmod, a.copy_M = 2 * nmod + 1;
            }
            if (inv == 3) {
                // nb ciphertexts: 2 L plain limbs and L * L digit limbs each
                const u32 per = (2 * nmod + nmod * nmod) << s0;
                nlift_rows = (nitems + per - 1) / per;
                a.lift_first = nlift_rows * ((2 * nmod) << s0);
                if (a.lift_first > nitems) a.lift_first = nitems;
                a.lift_L = nmod, a.lift_stride = (size_t)nmod * N;
            }
            for (int r = 0; r < nrot; r++) {
                CK(hipMalloc((void **)&d[r], (size_t)nitems * NS * 8));
                CK(hipMemset(d[r], 1, (size_t)nitems * NS * 8));
                if (inv == 2) CK(hipMalloc((void **)&d_copy[r], rows * 4 * a.copy_M * N * 8));
                if (inv == 3) {
                    CK(hipMalloc((void **)&d_src[r], (size_t)(nlift_rows + 1) * nmod * N * 8));
                    CK(hipMemset(d_src[r], 1, (size_t)(nlift_rows + 1) * nmod * N * 8));
                }
            }
            const u32 grid = nitems < LAB_SLOTS ? nitems : LAB_SLOTS;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            const int iters = 20;
            for (int it = 0; it < iters + 3; it++) {
                if (it == 3) CK(hipEventRecord(e0, 0));
                const int r = it % nrot;
                a.data = d[r], a.copy_out = d_copy[r];
                if (inv == 3) a.data2 = d[r] + (size_t)a.lift_first * NS, a.lift_src = d_src[r];
                if (inv == 1 || inv == 2)
                    hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, true>), dim3(grid), dim3(T), lds, 0, a);
                else if (inv == 3)
                    hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, false, true>), dim3(grid), dim3(T), lds, 0, a);
                else
                    hipLaunchKernelGGL((ntt16_kernel_t<LAB_LOGNS, false>), dim3(grid), dim3(T), lds, 0, a);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            static const char *names[4] = {"fwd", "inv", "inv std-in + copy", "fwd + digit lift"};
            printf("%s nitems=%5u  %8.2f us/launch  %6.2f slices/us  %7.1f GB/s alg (%.3f of 8 TB/s)\n", names[inv], nitems, us,
                   nitems / us, 16.0 * NS * nitems / (us * 1e-6) / 1e9, 16.0 * NS * nitems / (us * 1e-6) / 8e12);
            for (int r = 0; r < nrot; r++) {
                CK(hipFree(d[r]));
                if (d_copy[r]) CK(hipFree(d_copy[r]));
                if (d_src[r]) CK(hipFree(d_src[r]));
            }
        }
    return 0;
}
