// kernels_pie.hip -- the coefficient-wise kernels of BatchedFHEHIPPIE::run() for gfx950:
// fused ct x pt multiply-accumulate (stage A), HPS base conversions, tensor product, BV digit
// decomposition and key-switch accumulation, mask multiply, automorphism permutation, packed
// encoding.  All are streaming u64 modular arithmetic: one thread per coefficient, consecutive
// lanes on consecutive coefficients (coalesced 8-byte or 16-byte lanes), constants scalar-loaded
// from one DevConsts block.  Reference call sites: BatchedFHEHIPPIE.cpp:101-127.
#include <stdlib.h>

#include <algorithm>

#include "kernels.hpp"
#include "madasm.h"

namespace piehip {

static const u32 TPB = 256;

// set by the calling context right before its launches (all Q, P moduli in (2^59, 2^60)); per host thread, so
// contexts with different moduli can be driven from different threads
static thread_local bool g_small_moduli = false;
void set_small_moduli(bool v) { g_small_moduli = v; }

// ---------------------------------------------------------------------------------------------
// Stage A (rows A3+A4): acc[beta][h][c][l][n] = sum_j idx[h][j][c][l][n] * db[h][beta][j][l][n] + minus[c][l][n]
//
// HBM-bound: every database plaintext limb is read exactly once per run() (b K E L W bytes, 196 MiB at C3).
// A thread owns two adjacent coefficients (16-byte lanes) of one (h, limb) and BPT bin layers: the two index
// ciphertext components are loaded once per j and reused for all BPT layers, so index traffic is
// (b / BPT) K E 2L W instead of b K E 2L W; the BPT database loads per j are independent streams in flight.
// 128-bit lazy accumulation: moduli may be up to 61 bits (HostParams::init), so products are < 2^122 and the
// accumulator is reduced every 32 terms (2^61 + 32 * 2^122 < 2^128); one more Barrett reduction at the end.
// ---------------------------------------------------------------------------------------------
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// Block -> (coefficient block, limb, inner hash function, group of bin layers).  The `groups` blocks of one tile read the same
// index-ciphertext words (and different database words): they should run on the same XCD, one after the other, so that all
// but the first take the index words from that XCD's L2.  Workgroups go to the eight XCDs round-robin by their linear id, so
// a one-dimensional grid is cut as id = 8 * slot + xcd, slot = tile_of_this_xcd * groups + group.  (With the layer group in
// blockIdx.z the blocks of a tile were `tiles` apart in dispatch order: every group fetched the index matrix from HBM again.)
struct StageATile {
    u32 bx, l, hz, grp;
};
__device__ __forceinline__ bool stage_a_tile(u32 nx, u32 L, u32 tiles, u32 groups, StageATile &t)
{
    const u32 id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    t.grp = slot % groups;
    const u32 tile = (slot / groups) * 8 + xcd;
    if (tile >= tiles) return false;
    t.bx = tile % nx;
    t.l = (tile / nx) % L;
    t.hz = tile / (nx * L);
    return true;
}
static dim3 stage_a_grid(u32 nx, u32 L, u32 hn, u32 groups) { return dim3(8 * ((nx * L * hn + 7) / 8) * groups); }

template <int BPT>
__global__ void __launch_bounds__(TPB) stage_a_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                      const u64 *__restrict__ idx, const u64 *__restrict__ minus,
                                                      const u64 *__restrict__ db, u64 *__restrict__ acc, u32 bstride, u32 h0,
                                                      u32 nq, u32 q)
{
    const u32 n = 2 * (blockIdx.x * TPB + threadIdx.x);
    const u32 l = blockIdx.y;
    const u32 groups = b / BPT;
    const u32 h = h0 + blockIdx.z / groups, beta0 = (blockIdx.z % groups) * BPT;
    if (n >= N) return;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + n;
    const u64 *pd = db + (((size_t)h * bstride + beta0) * E) * LN + (size_t)l * N + n;  // db is [K][bstride][E][L][N]
    const size_t bin_stride = (size_t)E * LN;
    U128 a[BPT][2][2];
#pragma unroll
    for (int t = 0; t < BPT; t++)
#pragma unroll
        for (int c = 0; c < 2; c++) a[t][c][0] = a[t][c][1] = U128{0, 0};
    for (u32 j = 0; j < E; j++) {
        const u64x2 i0 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN);
        const u64x2 i1 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN + LN);
        u64x2 d[BPT];
#pragma unroll
        for (int t = 0; t < BPT; t++) {  // streamed once per run(): non-temporal (see stage_a_mad_kernel)
            typedef u64 u64v2 __attribute__((ext_vector_type(2)));
            const u64v2 v = __builtin_nontemporal_load(reinterpret_cast<const u64v2 *>(pd + (size_t)t * bin_stride + (size_t)j * LN));
            d[t].x = v.x, d[t].y = v.y;
        }
#pragma unroll
        for (int t = 0; t < BPT; t++) {
            mac128(a[t][0][0], i0.x, d[t].x);
            mac128(a[t][0][1], i0.y, d[t].y);
            mac128(a[t][1][0], i1.x, d[t].x);
            mac128(a[t][1][1], i1.y, d[t].y);
        }
        if ((j & 31) == 31) {
#pragma unroll
            for (int t = 0; t < BPT; t++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int e = 0; e < 2; e++) a[t][c][e] = U128{reduce128(a[t][c][e], m), 0};
        }
    }
    const u64x2 m0 = *reinterpret_cast<const u64x2 *>(minus + (size_t)l * N + n);
    const u64x2 m1 = *reinterpret_cast<const u64x2 *>(minus + LN + (size_t)l * N + n);
#pragma unroll
    for (int t = 0; t < BPT; t++) {
        u64 *po = acc + ((((size_t)(beta0 + t) * nq + q) * K + h) * 2) * LN + (size_t)l * N + n;
        u64x2 r0, r1;
        r0.x = addmod(reduce128(a[t][0][0], m), m0.x, m.q);
        r0.y = addmod(reduce128(a[t][0][1], m), m0.y, m.q);
        r1.x = addmod(reduce128(a[t][1][0], m), m1.x, m.q);
        r1.y = addmod(reduce128(a[t][1][1], m), m1.y, m.q);
        *reinterpret_cast<u64x2 *>(po) = r0;
        *reinterpret_cast<u64x2 *>(po + LN) = r1;
    }
}

// Same work with carry-free column accumulators on v_mad_u64_u32 (madasm.h): the 128-bit form above is bound by
// its 64x64->128 multiplies (46 SIMD cycles each, 3.9 TB/s at C3), this one by HBM.  Needs every modulus
// < 2^60 (a carry sweep every 8 terms keeps the low columns from overflowing, a reduction every 15 the top one).
// One coefficient per thread and SA_DEPTH terms in flight behind the one being accumulated: a term is 2 + BPT loads of
// 8 bytes per lane against microseconds of HBM latency and ~80 instructions of arithmetic.  tools/microbench_stage_a.hip
// runs this kernel over a rotation of databases (every launch from HBM): 49 us for the 196 MiB of the headline workload
// with seven layers per thread (168 registers, three waves per SIMD), the same with 16-byte lanes and one term ahead, 63 us
// when it is forced to four waves per SIMD (spills) -- and 35 us for a plain read of the same bytes.
// Lane-ordered home of coefficient n of a limb (ntt16_kernel.h: slices of 2^logns coefficients, T = 2^logns / 16 threads, thread tau
// holds elements 16 tau .. 16 tau + 15 and stores pair j at 2 (T j + tau)): the offset inside the limb
__device__ __forceinline__ u32 lane_home(u32 n, u32 logns)
{
    const u32 ns = 1u << logns, e = n & (ns - 1);
    return (n & ~(ns - 1)) + 2 * ((ns >> 4) * ((e >> 1) & 7) + (e >> 4)) + (e & 1);
}

// a + b mod q for canonical a, b and q < 2^62, without the compare / select pair hipcc makes of addmod(): v_cndmask_b32 on VCC
// issues at a sixth of the rate of the other vector instructions here (profiles/r03/microbench_operands.txt: 23.7 cycles per
// wave against 4.2), and the epilogue of stage A has 24 of these sums per thread
__device__ __forceinline__ u64 addmod_nb(u64 a, u64 b, u64 q)
{
    // (as an instruction block: written in C the compiler turns the mask back into a compare and a select)
    const u64 s = a + b;
    u32 lo, hi;
    asm("v_lshl_add_u64 v[60:61], %[x], 0, %[nm]\n\t"  // s - q: negative iff s < q
        "v_ashrrev_i32 v62, 31, v61\n\t"
        "v_bfi_b32 %[lo], v62, %[xl], v60\n\t"
        "v_bfi_b32 %[hi], v62, %[xh], v61"
        : [lo] "=&v"(lo), [hi] "=&v"(hi)
        : [x] "v"(s), [xl] "v"((u32)s), [xh] "v"((u32)(s >> 32)), [nm] "s"(0 - q)
        : "v62", "v60", "v61");
    return ((u64)hi << 32) | lo;
}

static const int SA_DEPTH = 4;
template <bool W124>
__device__ __forceinline__ u64 colacc_reduce(const ColAcc &a, const Mod &m, u64 nq);  // (instruction block, defined below)
__device__ __forceinline__ u64 colacc_reduce123_lazy(const ColAcc &a, const Mod &m, u64 nq);
template <int BPT>
__global__ void __launch_bounds__(TPB) stage_a_mad_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                          const u64 *__restrict__ idx, const u64 *__restrict__ minus,
                                                          const u64 *__restrict__ db, u64 *__restrict__ acc, u32 bstride, u32 h0,
                                                          u32 nq, u32 q, StageAXOut xo)
{
    const u32 nl = threadIdx.x;  // lane part of the coefficient index: every stream is a uniform base plus this
    const u32 n = blockIdx.x * TPB + nl;
    const u32 l = blockIdx.y;
    const u32 groups = b / BPT;
    const u32 h = h0 + blockIdx.z / groups, beta0 = (blockIdx.z % groups) * BPT;
    if (n >= N) return;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    // uniform stream bases (scalar registers where the compiler can; the loads take base + lane offset)
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + blockIdx.x * TPB;
    const u64 *pd = db + (((size_t)h * bstride + beta0) * E) * LN + (size_t)l * N + blockIdx.x * TPB;  // db is [K][bstride][E][L][N]
    const size_t bin_stride = (size_t)E * LN;
    ColAcc a[BPT][2];
#pragma unroll
    for (int t = 0; t < BPT; t++) a[t][0] = a[t][1] = ColAcc{0, 0, 0};
    auto load_term = [&](u32 j, u64 (&vi)[2], u64 (&vd)[BPT]) {
        const u64 *pij = pi + (size_t)j * 2 * LN, *pdj = pd + (size_t)j * LN;
        vi[0] = pij[nl];
        vi[1] = (pij + LN)[nl];
#pragma unroll
        // the database is read once per run() and is as large as the infinity cache: non-temporal loads keep it from evicting
        // the (dirty) intermediate arrays the neighbouring kernels hand to each other there -- 59 -> 50 us for this kernel
        // inside a run(), where it otherwise also paid for those write-backs (alone on the GPU it took 49 us either way)
        for (int t = 0; t < BPT; t++) vd[t] = __builtin_nontemporal_load(pdj + (size_t)t * bin_stride + nl);
    };
    // a ring of SA_DEPTH term buffers; the term loop is unrolled SA_DEPTH times so that the ring needs no register moves
    u64 qiv[SA_DEPTH][2], qdv[SA_DEPTH][BPT];
#pragma unroll
    for (int d = 0; d < SA_DEPTH; d++)
        if ((u32)d < E) load_term(d, qiv[d], qdv[d]);
    auto term = [&](u32 j, u64 (&vi)[2], u64 (&vd)[BPT]) {
        const Split30 i0 = split30(vi[0]), i1 = split30(vi[1]);
#pragma unroll
        for (int t = 0; t < BPT; t++) colacc_mac2(a[t][0], a[t][1], i0, i1, vd[t]);
        if (j + SA_DEPTH < E) load_term(j + SA_DEPTH, vi, vd);  // refill this ring slot (the other slots are in flight)
        if ((j % COLACC_MAX_TERMS) == COLACC_MAX_TERMS - 1 && j + 1 < E) {  // more terms follow: make room in the low columns
#pragma unroll
            for (int t = 0; t < BPT; t++) colacc_carry(a[t][0]), colacc_carry(a[t][1]);
        }
        if ((j % COLACC_MAX_TOTAL) == COLACC_MAX_TOTAL - 1 && j + 1 < E) {  // the top column is full: reduce and start over
#pragma unroll
            for (int t = 0; t < BPT; t++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const u64 r = reduce124(colacc_value(a[t][c]), m);  // (the epilogue's instruction block here would cost this kernel, at seven
                                                                       // layers per thread, its third wave per SIMD: 176 registers)
                    a[t][c] = ColAcc{r & 0x3FFFFFFFull, r >> 30, 0};
                }
        }
    };
    for (u32 j = 0; j < E; j += SA_DEPTH) {
#pragma unroll
        for (int d = 0; d < SA_DEPTH; d++)
            if (j + d < E) term(j + d, qiv[d], qdv[d]);
    }
    const bool xdir = h == 0 && xo.out != nullptr;
    const u32 noff = n + ((xdir ? ~0u : 0u) & (lane_home(n, xo.logns) - n));
#pragma unroll
    for (int t = 0; t < BPT; t++) {
        const size_t row = (size_t)(beta0 + t) * nq + q;
        // operand X of the first product goes straight to the QP operand array, lane-ordered (StageAXOut).  Uniform choices of
        // base and stride, and the lane offset blended with a mask: no per-lane selects (v_cndmask on VCC: see addmod_nb)
        u64 *const base = xdir ? xo.out + ((row * 4) * xo.M + l) * N : acc + ((row * K + h) * 2) * LN + (size_t)l * N;
        const size_t cstride = xdir ? (size_t)xo.M * N : LN;
        u64 *const po = base + noff;
#pragma unroll
        for (int c = 0; c < 2; c++)
            po[(size_t)c * cstride] = addmod_nb(colacc_reduce<true>(a[t][c], m, 0 - m.q), minus[(size_t)c * LN + (size_t)l * N + n], m.q);
    }
}

// A batch of Q queries against one database (piehip_set_query_batch): the thread holds the accumulators of Q queries x BPT bin
// layers, so a database word is loaded once for the whole batch -- the database is 196 of the 253 MiB stage A moves for one
// query at the headline shape -- and an index-ciphertext word once per BPT layers as before.  A term is 2 Q + BPT loads for
// 2 Q BPT multiply-accumulates (one query, seven layers: 9 for 14; two queries, four layers: 8 for 16).
// acc rows: [bin layer][query][K][2][L][N] (the product chain treats (layer, query) as one batch index).
// 27-31 us per query at the headline shape against 46-49 for one query per launch; what bounds it then is issue time plus
// memory latency at three waves per SIMD, not HBM bytes -- profiles/r03/stage_a_batch_microbench.txt, with the cuts that were
// tried on top (index words shared through L1 or LDS) and dropped.
template <int BPT, int Q, int DEPTH>
__global__ void __launch_bounds__(TPB) stage_a_mad_batch_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                                StageAQueries qs, const u64 *__restrict__ db, u64 *__restrict__ acc,
                                                                u32 bstride, u32 h0, u32 nq, u32 q0, u32 tiles, StageAXOut xo)
{
    StageATile tl;
    if (!stage_a_tile((N + TPB - 1) / TPB, L, tiles, (b + BPT - 1) / BPT, tl)) return;
    const u32 nl = threadIdx.x, n0 = tl.bx * TPB, l = tl.l, h = h0 + tl.hz, beta0 = tl.grp * BPT;
    const u32 n = n0 + nl;
    if (n >= N) return;
    // the last group of a launch may hold fewer than BPT layers (b no multiple of BPT): its missing layers repeat its last one --
    // the same loads again (L1 hits) and multiply-adds nobody stores -- so that the term loop stays free of per-layer branches
    // (skipping them under uniform branches instead was measured: 4-17 % slower, profiles/r05/stage_a_batch_layer_groups.txt)
    const u32 tmax = __builtin_amdgcn_readfirstlane(min((u32)BPT, b - beta0) - 1);
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    const size_t ioff = ((size_t)h * E) * 2 * LN + (size_t)l * N + n0;
    const u64 *pd = db + (((size_t)h * bstride + beta0) * E) * LN + (size_t)l * N + n0;
    const size_t bin_stride = (size_t)E * LN;
    ColAcc a[Q][BPT][2];
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
        for (int t = 0; t < BPT; t++) a[q][t][0] = a[q][t][1] = ColAcc{0, 0, 0};
    auto load_term = [&](u32 j, u64 (&vi)[Q][2], u64 (&vd)[BPT]) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const u64 *pij = qs.idx[q] + ioff + (size_t)j * 2 * LN;
            vi[q][0] = pij[nl];
            vi[q][1] = (pij + LN)[nl];
        }
        const u64 *pdj = pd + (size_t)j * LN;
#pragma unroll
        for (int t = 0; t < BPT; t++) vd[t] = __builtin_nontemporal_load(pdj + (size_t)min((u32)t, tmax) * bin_stride + nl);
    };
    u64 qiv[DEPTH][Q][2], qdv[DEPTH][BPT];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if ((u32)d < E) load_term(d, qiv[d], qdv[d]);
    auto term = [&](u32 j, u64 (&vi)[Q][2], u64 (&vd)[BPT]) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const Split30 i0 = split30(vi[q][0]), i1 = split30(vi[q][1]);
#pragma unroll
            for (int t = 0; t < BPT; t++) colacc_mac2(a[q][t][0], a[q][t][1], i0, i1, vd[t]);
        }
        if (j + DEPTH < E) load_term(j + DEPTH, vi, vd);
        if ((j % COLACC_MAX_TERMS) == COLACC_MAX_TERMS - 1 && j + 1 < E) {
#pragma unroll
            for (int q = 0; q < Q; q++)
#pragma unroll
                for (int t = 0; t < BPT; t++) colacc_carry(a[q][t][0]), colacc_carry(a[q][t][1]);
        }
        if ((j % COLACC_MAX_TOTAL) == COLACC_MAX_TOTAL - 1 && j + 1 < E) {
#pragma unroll
            for (int q = 0; q < Q; q++)
#pragma unroll
                for (int t = 0; t < BPT; t++)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const u64 r = colacc_reduce<true>(a[q][t][c], m, 0 - m.q);  // (the instruction block of the epilogue: no compare / select pairs)
                        a[q][t][c] = ColAcc{r & 0x3FFFFFFFull, r >> 30, 0};
                    }
        }
    };
    for (u32 j = 0; j < E; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (j + d < E) term(j + d, qiv[d], qdv[d]);
    }
    const bool xdir = h == 0 && xo.out != nullptr;
    const u32 noff = n + ((xdir ? ~0u : 0u) & (lane_home(n, xo.logns) - n));
#pragma unroll
    for (int q = 0; q < Q; q++) {
        u64 mi[2];
#pragma unroll
        for (int c = 0; c < 2; c++) mi[c] = qs.minus[q][(size_t)c * LN + (size_t)l * N + n];
#pragma unroll
        for (int t = 0; t < BPT; t++) {
            if ((u32)t > tmax) continue;   // (uniform: a layer the ragged last group does not have)
            const size_t row = (size_t)(beta0 + t) * nq + q0 + q;
            // operand X of the first product goes straight to the QP operand array, lane-ordered (StageAXOut).  Uniform choices
            // of base and stride, and the lane offset blended with a mask: no per-lane selects (v_cndmask on VCC: see addmod_nb)
            u64 *const base = xdir ? xo.out + ((row * 4) * xo.M + l) * N : acc + ((row * K + h) * 2) * LN + (size_t)l * N;
            const size_t cstride = xdir ? (size_t)xo.M * N : LN;
            u64 *const po = base + noff;
#pragma unroll
            for (int c = 0; c < 2; c++) po[(size_t)c * cstride] = addmod_nb(colacc_reduce<true>(a[q][t][c], m, 0 - m.q), mi[c], m.q);
        }
    }
}

// one launch over b bin layers whose count is a multiple of the per-thread layer count `bpt`
static void launch_stage_a_uniform(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus,
                                   const u64 *db, u64 *acc, hipStream_t st, bool mad, u32 bstride, u32 h0, u32 hn, int bpt, u32 nq,
                                   u32 q, StageAXOut xo)
{
    const int cpt = mad ? 1 : 2;
    dim3 grid((N / cpt + TPB - 1) / TPB, L, hn * (b / bpt));
#define SA(B_)                                                                                                       \
    do {                                                                                                             \
        if (mad) hipLaunchKernelGGL(stage_a_mad_kernel<B_>, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc, bstride, h0, nq, q, xo); \
        else hipLaunchKernelGGL(stage_a_kernel<B_>, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc, bstride, h0, nq, q);  \
    } while (0)
    switch (bpt) {
        case 8: hipLaunchKernelGGL(stage_a_kernel<8>, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc, bstride, h0, nq, q); break;
        case 7: SA(7); break;
        case 6: SA(6); break;
        case 5: SA(5); break;
        case 4: SA(4); break;
        case 3: SA(3); break;
        case 2: SA(2); break;
        default: SA(1); break;
    }
#undef SA
}

void launch_stage_a(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus,
                    const u64 *db, u64 *acc, hipStream_t st, bool small_moduli, u32 bstride, u32 h0, u32 hn, u32 nq, u32 q,
                    const StageAXOut *xop)
{
    StageAXOut xo;
    if (xop && small_moduli) xo = *xop;  // (the 128-bit kernel keeps writing acc: callers check small_moduli before they rely on xo)
    if (!bstride) bstride = b;
    if (!hn) hn = K - h0;
    if (!nq) nq = 1;
    const bool mad = small_moduli;
    const u32 cap = mad ? 7 : 8;
    // Bin layers per thread: a thread re-reads the two index-ciphertext limbs once per group of layers, so groups should be
    // as large as the accumulators allow.  A divisor of b in [4, cap] gives one uniform launch; otherwise (b = 17, 23, ...:
    // one queue's share of an odd split) groups of `cap` layers plus one launch for the remainder -- never one layer per
    // thread, which read the index matrix b times.
    u32 bpt = 0;
    for (u32 c = cap; c >= 4; c--)
        if (b % c == 0) {
            bpt = c;
            break;
        }
    if (b <= cap) bpt = b;
    const size_t LN = (size_t)L * N;
    if (bpt) {
        launch_stage_a_uniform(dc, N, L, K, b, E, idx, minus, db, acc, st, mad, bstride, h0, hn, (int)bpt, nq, q, xo);
        return;
    }
    const u32 full = (b / cap) * cap, rest = b - full;
    launch_stage_a_uniform(dc, N, L, K, full, E, idx, minus, db, acc, st, mad, bstride, h0, hn, (int)cap, nq, q, xo);
    StageAXOut xr = xo;
    if (xr.out) xr.out += (size_t)full * nq * 4 * xr.M * N;
    launch_stage_a_uniform(dc, N, L, K, rest, E, idx, minus, db + (size_t)full * E * LN, acc + (size_t)full * nq * K * 2 * LN, st, mad,
                           bstride, h0, hn, (int)rest, nq, q, xr);
}

// Stage A of a query batch: acc[b][nq][K][2][L][N].  Column-accumulator kernel (every modulus < 2^60): groups of two to four
// queries per launch (plus one launch for the layers left over by the per-thread layer count), each reading the database once;
// otherwise one launch series per query.
template <int Q, int BPT>
static void launch_stage_a_batch_qb(const DevConsts *dc, u32 N, u32 L, u32 K, u32 nb, u32 E, const StageAQueries &qs, const u64 *db,
                                    u64 *acc, hipStream_t st, u32 bstride, u32 h0, u32 hn, u32 nq, u32 q0, StageAXOut xo)
{
    const u32 nx = (N + TPB - 1) / TPB;
    // terms in flight behind the one being accumulated (profiles/r03/stage_a_batch_microbench.txt; r05: two for the wide tilings)
    constexpr int DEPTH = (Q == 4 || Q * BPT >= 9) ? 2 : 3;
    hipLaunchKernelGGL((stage_a_mad_batch_kernel<BPT, Q, DEPTH>), stage_a_grid(nx, L, hn, (nb + BPT - 1) / BPT), dim3(TPB), 0, st, dc, N, L, K,
                       nb, E, qs, db, acc, bstride, h0, nq, q0, nx * L * hn, xo);
}
// Bin layers per thread.  What bounds this kernel is the traffic through the L1s (profiles/r05/stage_a_batch_prefetch_really_in_flight.txt:
// ~8.6 TB/s of L1 misses chip-wide, three quarters of them index words that a thread re-reads from the L2 once per GROUP of layers), so
// groups should be as large as the registers allow: four layers for two queries, three for three queries (164 VGPRs either way:
// three waves per SIMD), three for four (208: two waves).  r03-r04 had two layers for three queries: 81.7 us for twelve layers against 69.9 with
// groups of three (tools/microbench_stage_a_batch.hip, r05).  ONE launch whatever the layer count: the last group is ragged (the
// kernel repeats its last layer) -- a remainder launch of one or two layers is all latency (25 us for two layers alone), and two
// launches of half the groups each leave the chip a partial round of waves twice.
template <int Q>
static void launch_stage_a_batch_q(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const StageAQueries &qs, const u64 *db,
                                   u64 *acc, hipStream_t st, u32 bstride, u32 h0, u32 hn, u32 nq, u32 q0, StageAXOut xo)
{
    constexpr u32 cap = Q == 2 ? 4 : 3;  // bin layers per thread (Q = 4, three layers: 208 VGPRs, two waves -- still 4 % ahead of two layers)
    // the group size that issues the fewest loads per term over the launch: ceil(b / g) groups of 2 Q index words + g database words
    // (a ragged last group loads and multiplies its padding too: six layers of two queries are better off as 3 + 3 than as 4 + 2)
    u32 bpt = 1, best = ~0u;
    for (u32 g = 1; g <= cap && g <= b; g++) {
        const u32 loads = ((b + g - 1) / g) * (2 * Q + g);
        if (loads <= best) best = loads, bpt = g;
    }
    if constexpr (cap >= 4)
        if (bpt == 4) return launch_stage_a_batch_qb<Q, 4>(dc, N, L, K, b, E, qs, db, acc, st, bstride, h0, hn, nq, q0, xo);
    if constexpr (cap >= 3)
        if (bpt == 3) return launch_stage_a_batch_qb<Q, 3>(dc, N, L, K, b, E, qs, db, acc, st, bstride, h0, hn, nq, q0, xo);
    if (bpt == 2) return launch_stage_a_batch_qb<Q, 2>(dc, N, L, K, b, E, qs, db, acc, st, bstride, h0, hn, nq, q0, xo);
    return launch_stage_a_batch_qb<Q, 1>(dc, N, L, K, b, E, qs, db, acc, st, bstride, h0, hn, nq, q0, xo);
}
void launch_stage_a_batch(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const StageAQueries &qs, u32 nq, const u64 *db,
                          u64 *acc, hipStream_t st, bool small_moduli, u32 bstride, u32 h0, u32 hn, const StageAXOut *xop)
{
    StageAXOut xo;
    if (xop && small_moduli) xo = *xop;
    if (!bstride) bstride = b;
    if (!hn) hn = K - h0;
    u32 q0 = 0;
    while (small_moduli && nq - q0 >= 2) {
        // groups of four, three or two queries (five: 3 + 2; six: 3 + 3; seven: 4 + 3)
        const u32 left = nq - q0, g = left == 5 || left == 6 ? 3 : std::min(left, 4u);
        StageAQueries sub = {};
        for (u32 q = 0; q < g; q++) sub.idx[q] = qs.idx[q0 + q], sub.minus[q] = qs.minus[q0 + q];
        if (g == 2) launch_stage_a_batch_q<2>(dc, N, L, K, b, E, sub, db, acc, st, bstride, h0, hn, nq, q0, xo);
        else if (g == 3) launch_stage_a_batch_q<3>(dc, N, L, K, b, E, sub, db, acc, st, bstride, h0, hn, nq, q0, xo);
        else launch_stage_a_batch_q<4>(dc, N, L, K, b, E, sub, db, acc, st, bstride, h0, hn, nq, q0, xo);
        q0 += g;
    }
    for (; q0 < nq; q0++)
        launch_stage_a(dc, N, L, K, b, E, qs.idx[q0], qs.minus[q0], db, acc, st, small_moduli, bstride, h0, hn, nq, q0, xop);
}

// ---------------------------------------------------------------------------------------------
// Base conversions (row A6).  One thread per coefficient reads its L (or 2L+1) residues at limb
// stride N and writes every output limb.  Rounding terms use the 60-bit fixed-point rule of
// modarith.h (identical to oracle/pie_oracle.c: po_expand_q_to_qp, po_scale_pq_expand, po_scale_round_tp).
// ---------------------------------------------------------------------------------------------
// The per-limb iterations of the base conversions are unrolled (the residue arrays must stay in registers), and each
// iteration needs its own dozen table entries: moduli, Barrett and Shoup constants, one column of a conversion matrix.
// Left alone, the compiler loads all of them at the top of the kernel -- 200+ scalar registers' worth -- and spills them
// into vector lanes (a third of the instructions of these kernels were v_readlane / v_writelane / v_mov).  So every
// iteration starts with a scheduling fence and reads its constants through a freshly laundered pointer into the
// constant address space: scalar loads, issued in that iteration, dead at its end.
#define PIE_ITER_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef const __attribute__((address_space(4))) DevConsts *DcC;
__device__ __forceinline__ DcC dc_iter(const DevConsts *dc)
{
    u64 v = (u64)dc;
    asm volatile("" : "+s"(v));
    return (DcC)v;
}
__device__ __forceinline__ Mod ld_mod(DcC c, u32 i)
{
    Mod m;
    m.q = c->mod[i].q, m.r0 = c->mod[i].r0, m.r1 = c->mod[i].r1, m.n_inv = 0, m.n_inv_sh = 0;
    m.fconst = c->mod[i].fconst, m.fshift = c->mod[i].fshift, m.pad = 0;
    return m;
}

// Conditional subtraction with a wave-uniform modulus, without VCC: t = x - m as x + (2^64 - m) (one v_lshl_add_u64 with the
// negated modulus in scalar registers), then a select on the sign of t (v_ashrrev_i32 + two v_bfi_b32).  hipcc's lowering of
// `x >= m ? x - m : x` is compare + two moves of the modulus into vector registers + two v_cndmask + a two-instruction
// subtraction through VCC with its wait state: 8 instructions against 4, ~120 times per coefficient pair of a base conversion.
// Needs x < 2^63 and x - m > -2^63 (all residues here are below 2^62).
__device__ __forceinline__ u64 neg_u(u64 m)  // 2^64 - m, kept opaque so that x + neg_u(m) stays an addition
{
    u64 n = 0 - m;
    asm("" : "+s"(n));
    return n;
}
__device__ __forceinline__ u64 sel_neg(u64 x, u64 t, u32 &mask)  // t < 0 ? x : t;  mask = t < 0 ? ~0 : 0
{
    u32 lo, hi, k;
    asm("v_ashrrev_i32 %[k], 31, %[th]\n\t"
        "v_bfi_b32 %[lo], %[k], %[xl], %[tl]\n\t"
        "v_bfi_b32 %[hi], %[k], %[xh], %[th]"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [k] "=&v"(k)
        : [xl] "v"((u32)x), [xh] "v"((u32)(x >> 32)), [tl] "v"((u32)t), [th] "v"((u32)(t >> 32)));
    mask = k;
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 csub_u(u64 x, u64 negm)
{
    u32 k;
    return sel_neg(x, x + negm, k);
}
// the modarith.h formulas with this subtraction (same values; moduli wave-uniform: they come from ld_mod / scalar loads)
__device__ __forceinline__ u64 mul_shoup_lazy_u(u64 a, u64 w, u64 wsh, u64 q) { return a * w - mulhi(a, wsh) * q; }
__device__ __forceinline__ u64 mul_shoup_u(u64 a, u64 w, u64 wsh, u64 q, u64 negq) { return csub_u(mul_shoup_lazy_u(a, w, wsh, q), negq); }
__device__ __forceinline__ u64 fixfrac_u(u64 y, const Mod &m) { return mulhi(y << m.fshift, m.fconst) >> 3; }
__device__ __forceinline__ void divmod_shoup_u(u64 a, u64 w, u64 wsh, u64 q, u64 negq, u64 &quot, u64 &rem)
{
    const u64 qe = mulhi(a, wsh);
    const u64 r = a * w - qe * q;
    u32 k;
    rem = sel_neg(r, r + negq, k);
    quot = qe + 1 + (u64)(int64_t)(int32_t)k;  // + 1 unless r < q
}
__device__ __forceinline__ u64 reduce123_u(U128 z, const Mod &m, u64 negq, u64 neg2q)
{
    const u64 mu = (m.r1 << 59) | (m.r0 >> 5);
    const u64 zh = (z.hi << 5) | (z.lo >> 59);
    const u64 qhat = mulhi(zh, mu);
    return csub_u(csub_u(z.lo - qhat * m.q, neg2q), negq);
}

// ---- constant-operand modular products as instruction blocks (every modulus < 2^60) ---------------------------------------------
// A mixed VALU stream issues one instruction per ~4 cycles per wave on gfx950 whatever the instruction (a 2-cycle VOP1/VOP2
// that follows a 4-cycle VOP3 costs 4: profiles/r03/microbench_operands.txt), so these kernels are bound by their instruction
// COUNT.  hipcc spends 28 instructions on a Shoup product with a wave-uniform constant (an exact __umul64hi with six register
// moves, three 64-bit low products through v_mul_lo_u32 + v_add3_u32, a subtraction through VCC with its wait state); the NTT
// butterfly's formulation does it in 12 -- quotient estimate from the 63-bit constant floor(w 2^63 / q) with three multiplier
// operations (error <= 3), remainder a w + qe (2^64 - q) on two v_mad_u64_u32 chains -- plus two sign-mask subtractions for a
// canonical result.  Temporaries whose halves are needed live in fixed registers v60-v71 (low enough that scale_round stays at
// 76 VGPRs = six waves per SIMD: its 5.25 waves per SIMD are then one round; an asm operand cannot name the
// halves of a 64-bit pair); constants are SGPR operands, one per instruction (constant-bus limit of VOP3 on gfx9).
#define PIE_ASM_CLOB "vcc", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
// a < 2^63, w < q < 2^60: v[64:65] <- all but the last product of a w mod q + {0..3} q, v[62:63] <- the quotient estimate
#define PIE_SHOUP63_HEAD                                        \
    "v_mad_u64_u32 v[60:61], vcc, %[al], %[sh], 0\n\t"          \
    "v_mad_u64_u32 v[64:65], vcc, %[al], %[wl], 0\n\t"          \
    "v_mad_u64_u32 v[60:61], vcc, %[ah], %[sl], v[60:61]\n\t"   \
    "v_mad_u64_u32 v[66:67], vcc, %[al], %[wh], 0\n\t"          \
    "v_lshlrev_b32 v70, 1, %[ah]\n\t"                           \
    "v_mad_u64_u32 v[66:67], vcc, %[ah], %[wl], v[66:67]\n\t"   \
    "v_lshrrev_b64 v[60:61], 31, v[60:61]\n\t"                  \
    "v_mad_u64_u32 v[62:63], vcc, v70, %[sh], v[60:61]\n\t"     \
    "v_mad_u64_u32 v[66:67], vcc, v62, %[nqh], v[66:67]\n\t"    \
    "v_mad_u64_u32 v[66:67], vcc, v63, %[nql], v[66:67]\n\t"    \
    "v_add_u32 v65, v65, v66\n\t"
#define PIE_SHOUP63_IN(a, w, wsh, nq)                                                                                        \
    [al] "v"((u32)(a)), [ah] "v"((u32)((a) >> 32)), [wl] "s"((u32)(w)), [wh] "s"((u32)((w) >> 32)), [sl] "s"((u32)((wsh) >> 1)), \
        [sh] "s"((u32)((wsh) >> 33)), [nql] "s"((u32)(nq)), [nqh] "s"((u32)((nq) >> 32))
// a w mod q + {0,1,2,3} q for a < 2^63 (wsh = floor(w 2^64 / q), nq = 2^64 - q): 12 instructions
__device__ __forceinline__ u64 shoup63_lazy(u64 a, u64 w, u64 wsh, u64 nq)
{
    u64 r;
    asm(PIE_SHOUP63_HEAD
        "v_mad_u64_u32 %[r], vcc, v62, %[nql], v[64:65]"
        : [r] "=v"(r)
        : PIE_SHOUP63_IN(a, w, wsh, nq)
        : PIE_ASM_CLOB);
    return r;
}
// ... reduced to [0, q): 20 instructions
__device__ __forceinline__ u64 shoup63(u64 a, u64 w, u64 wsh, u64 nq)
{
    u32 lo, hi;
    const u64 n2q = 2 * nq;  // 2^64 - 2q
    asm(PIE_SHOUP63_HEAD
        "v_mad_u64_u32 v[64:65], vcc, v62, %[nql], v[64:65]\n\t"
        "v_lshl_add_u64 v[68:69], v[64:65], 0, %[n2q]\n\t"
        "v_ashrrev_i32 v71, 31, v69\n\t"
        "v_bfi_b32 v64, v71, v64, v68\n\t"
        "v_bfi_b32 v65, v71, v65, v69\n\t"
        "v_lshl_add_u64 v[68:69], v[64:65], 0, %[n1q]\n\t"
        "v_ashrrev_i32 v71, 31, v69\n\t"
        "v_bfi_b32 %[lo], v71, v64, v68\n\t"
        "v_bfi_b32 %[hi], v71, v65, v69"
        : [lo] "=v"(lo), [hi] "=v"(hi)
        : PIE_SHOUP63_IN(a, w, wsh, nq), [n2q] "s"(n2q), [n1q] "s"(nq)
        : PIE_ASM_CLOB);
    return ((u64)hi << 32) | lo;
}
// exact floor(a w / q) and a w mod q for a < q: the estimate is at most 3 short, and each of the two conditional
// subtractions reports whether it subtracted (64-bit sign masks M: quotient = qe + 3 + 2 M1 + M2).  23 instructions
__device__ __forceinline__ void divmod63(u64 a, u64 w, u64 wsh, u64 nq, u64 &quot, u64 &rem)
{
    u32 lo, hi;
    u64 qt;
    const u64 n2q = 2 * nq;
    asm(PIE_SHOUP63_HEAD
        "v_mad_u64_u32 v[64:65], vcc, v62, %[nql], v[64:65]\n\t"
        "v_lshl_add_u64 v[68:69], v[64:65], 0, %[n2q]\n\t"
        "v_ashrrev_i64 v[60:61], 63, v[68:69]\n\t"
        "v_bfi_b32 v64, v60, v64, v68\n\t"
        "v_bfi_b32 v65, v60, v65, v69\n\t"
        "v_lshl_add_u64 v[62:63], v[60:61], 1, v[62:63]\n\t"
        "v_lshl_add_u64 v[68:69], v[64:65], 0, %[n1q]\n\t"
        "v_ashrrev_i64 v[60:61], 63, v[68:69]\n\t"
        "v_bfi_b32 %[lo], v60, v64, v68\n\t"
        "v_bfi_b32 %[hi], v60, v65, v69\n\t"
        "v_lshl_add_u64 v[62:63], v[60:61], 0, v[62:63]\n\t"
        "v_lshl_add_u64 %[qt], v[62:63], 0, 3"
        : [lo] "=v"(lo), [hi] "=v"(hi), [qt] "=v"(qt)
        : PIE_SHOUP63_IN(a, w, wsh, nq), [n2q] "s"(n2q), [n1q] "s"(nq)
        : PIE_ASM_CLOB);
    quot = qt;
    rem = ((u64)hi << 32) | lo;
}
// exact floor(a b / 2^64), b wave-uniform: the cross products are summed with the carry kept (VCC is read two instructions
// after it is written: the wait states a VALU read of a VALU-written VCC needs on gfx950).  8 instructions (hipcc: 11)
__device__ __forceinline__ u64 mulhi_sb(u64 a, u64 b)
{
    u64 r;
    asm("v_mul_hi_u32 v60, %[al], %[bl]\n\t"
        "v_mov_b32 v61, 0\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[al], %[bh], v[60:61]\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[ah], %[bl], v[62:63]\n\t"
        "v_mad_u64_u32 v[66:67], s[96:97], %[ah], %[bh], 0\n\t"
        "v_lshrrev_b64 v[64:65], 32, v[62:63]\n\t"
        "v_addc_co_u32 v65, vcc, 0, v65, vcc\n\t"
        "v_lshl_add_u64 %[r], v[66:67], 0, v[64:65]"
        : [r] "=v"(r)
        : [al] "v"((u32)a), [ah] "v"((u32)(a >> 32)), [bl] "s"((u32)b), [bh] "s"((u32)(b >> 32))
        : PIE_ASM_CLOB, "s96", "s97");
    return r;
}
// Column accumulator -> residue in one block (2^59 < q < 2^60).  The columns are carry-normalised (c1' = c1 + (c0 >> 30),
// c2' = c2 + (c1' >> 30)), after which z >> 59 = (c2' << 1) | bit 29 of c1' and the low word of z is three disjoint bit fields;
// one-word Barrett as reduce123 / reduce124 of modarith.h (same quotient estimate, same remainder), mulhi as mulhi_sb, the
// remainder z + qhat (2^64 - q) on one v_mad_u64_u32 chain, sign-mask subtractions.  32 instructions (35 with W124) where the
// compiler's colacc_value + reduce123 take ~65 (128-bit additions through v_cmp / v_cndmask carries, an 11-instruction mulhi).
// W124: z < 2^124 (eight products), otherwise z < 2^123 (seven).
// a three-column accumulator below 2^123 to v[66:67] in [0, 4q) (v60-v71, vcc, s[96:97] as scratch): the body of colacc_reduce<false>
#define PIE_COLACC123_TO_4Q \
    "v_lshrrev_b64 v[60:61], 30, %[c0]\n\t" \
    "v_lshl_add_u64 v[60:61], v[60:61], 0, %[c1]\n\t" \
    "v_lshrrev_b64 v[62:63], 30, v[60:61]\n\t" \
    "v_lshl_add_u64 v[62:63], v[62:63], 0, %[c2]\n\t" \
    "v_lshlrev_b64 v[64:65], 1, v[62:63]\n\t" \
    "v_bfe_u32 v68, v60, 29, 1\n\t" \
    "v_and_b32 v66, 0x3fffffff, %[c0l]\n\t" \
    "v_bfe_u32 v67, v60, 2, 28\n\t" \
    "v_or_b32 v64, v64, v68\n\t" \
    "v_lshl_or_b32 v66, v60, 30, v66\n\t" \
    "v_lshl_or_b32 v67, v62, 28, v67\n\t" \
    "v_mul_hi_u32 v68, v64, %[mul]\n\t" \
    "v_mov_b32 v69, 0\n\t" \
    "v_mad_u64_u32 v[68:69], vcc, v64, %[muh], v[68:69]\n\t" \
    "v_mad_u64_u32 v[68:69], vcc, v65, %[mul], v[68:69]\n\t" \
    "v_mad_u64_u32 v[70:71], s[96:97], v65, %[muh], 0\n\t" \
    "v_lshrrev_b64 v[68:69], 32, v[68:69]\n\t" \
    "v_addc_co_u32 v69, vcc, 0, v69, vcc\n\t" \
    "v_lshl_add_u64 v[68:69], v[70:71], 0, v[68:69]\n\t" \
    "v_mad_u64_u32 v[70:71], vcc, v68, %[nqh], 0\n\t" \
    "v_mad_u64_u32 v[70:71], vcc, v69, %[nql], v[70:71]\n\t" \
    "v_add_u32 v67, v67, v70\n\t" \
    "v_mad_u64_u32 v[66:67], vcc, v68, %[nql], v[66:67]\n\t"
// ... left there: [0, 4q).  For values that go on into a folded forward transform (fold_store adds a [0, 4q) product to them and the
// transform takes anything below 8q) or into a Shoup product (any operand below 2^63): eight instructions less than the canonical form
__device__ __forceinline__ u64 colacc_reduce123_lazy(const ColAcc &a, const Mod &m, u64 nq)
{
    const u64 mu = (m.r1 << 59) | (m.r0 >> 5);   // floor(2^123 / q)
    u64 r;
    asm(PIE_COLACC123_TO_4Q
        "v_lshl_add_u64 %[r], v[66:67], 0, 0"
        : [r] "=v"(r)
        : [c0] "v"(a.c0), [c1] "v"(a.c1), [c2] "v"(a.c2), [c0l] "v"((u32)a.c0), [mul] "s"((u32)mu), [muh] "s"((u32)(mu >> 32)),
          [nql] "s"((u32)nq), [nqh] "s"((u32)(nq >> 32))
        : PIE_ASM_CLOB, "s96", "s97");
    return r;
}
template <bool W124>
__device__ __forceinline__ u64 colacc_reduce(const ColAcc &a, const Mod &m, u64 nq)
{
    const u64 mu = (m.r1 << 59) | (m.r0 >> 5);   // floor(2^123 / q)
    const u64 n2q = 2 * nq, n4q = 4 * nq;
    u64 r;
    if (!W124) {
        asm(PIE_COLACC123_TO_4Q
            "v_lshl_add_u64 v[60:61], v[66:67], 0, %[n2q]\n\t"
            "v_ashrrev_i32 v62, 31, v61\n\t"
            "v_bfi_b32 v66, v62, v66, v60\n\t"
            "v_bfi_b32 v67, v62, v67, v61\n\t"
            "v_lshl_add_u64 v[60:61], v[66:67], 0, %[n1q]\n\t"
            "v_ashrrev_i32 v62, 31, v61\n\t"
            "v_and_b32 v64, %[ql], v62\n\t"
            "v_and_b32 v65, %[qh], v62\n\t"
            "v_lshl_add_u64 %[r], v[60:61], 0, v[64:65]"
            : [r] "=v"(r)
            : [c0] "v"(a.c0), [c1] "v"(a.c1), [c2] "v"(a.c2), [c0l] "v"((u32)a.c0), [mul] "s"((u32)mu), [muh] "s"((u32)(mu >> 32)),
              [nql] "s"((u32)nq), [nqh] "s"((u32)(nq >> 32)), [n2q] "s"(n2q), [n1q] "s"(nq), [ql] "s"((u32)m.q), [qh] "s"((u32)(m.q >> 32))
            : PIE_ASM_CLOB, "s96", "s97");
    } else {
        // z >> 60 = c2' after the normalisation; qhat = 2 floor(zh mu / 2^64); remainder in [0, 7q): one more subtraction
        asm("v_lshrrev_b64 v[60:61], 30, %[c0]\n\t"
            "v_lshl_add_u64 v[60:61], v[60:61], 0, %[c1]\n\t"
            "v_lshrrev_b64 v[64:65], 30, v[60:61]\n\t"
            "v_lshl_add_u64 v[64:65], v[64:65], 0, %[c2]\n\t"
            "v_and_b32 v66, 0x3fffffff, %[c0l]\n\t"
            "v_bfe_u32 v67, v60, 2, 28\n\t"
            "v_lshl_or_b32 v66, v60, 30, v66\n\t"
            "v_lshl_or_b32 v67, v64, 28, v67\n\t"
            "v_mul_hi_u32 v68, v64, %[mul]\n\t"
            "v_mov_b32 v69, 0\n\t"
            "v_mad_u64_u32 v[68:69], vcc, v64, %[muh], v[68:69]\n\t"
            "v_mad_u64_u32 v[68:69], vcc, v65, %[mul], v[68:69]\n\t"
            "v_mad_u64_u32 v[70:71], s[96:97], v65, %[muh], 0\n\t"
            "v_lshrrev_b64 v[68:69], 32, v[68:69]\n\t"
            "v_addc_co_u32 v69, vcc, 0, v69, vcc\n\t"
            "v_lshl_add_u64 v[68:69], v[70:71], 0, v[68:69]\n\t"
            "v_lshlrev_b64 v[68:69], 1, v[68:69]\n\t"
            "v_mad_u64_u32 v[70:71], vcc, v68, %[nqh], 0\n\t"
            "v_mad_u64_u32 v[70:71], vcc, v69, %[nql], v[70:71]\n\t"
            "v_add_u32 v67, v67, v70\n\t"
            "v_mad_u64_u32 v[66:67], vcc, v68, %[nql], v[66:67]\n\t"
            "v_lshl_add_u64 v[60:61], v[66:67], 0, %[n4q]\n\t"
            "v_ashrrev_i32 v62, 31, v61\n\t"
            "v_bfi_b32 v66, v62, v66, v60\n\t"
            "v_bfi_b32 v67, v62, v67, v61\n\t"
            "v_lshl_add_u64 v[60:61], v[66:67], 0, %[n2q]\n\t"
            "v_ashrrev_i32 v62, 31, v61\n\t"
            "v_bfi_b32 v66, v62, v66, v60\n\t"
            "v_bfi_b32 v67, v62, v67, v61\n\t"
            "v_lshl_add_u64 v[60:61], v[66:67], 0, %[n1q]\n\t"
            "v_ashrrev_i32 v62, 31, v61\n\t"
            "v_and_b32 v64, %[ql], v62\n\t"
            "v_and_b32 v65, %[qh], v62\n\t"
            "v_lshl_add_u64 %[r], v[60:61], 0, v[64:65]"
            : [r] "=v"(r)
            : [c0] "v"(a.c0), [c1] "v"(a.c1), [c2] "v"(a.c2), [c0l] "v"((u32)a.c0), [mul] "s"((u32)mu), [muh] "s"((u32)(mu >> 32)),
              [nql] "s"((u32)nq), [nqh] "s"((u32)(nq >> 32)), [n4q] "s"(n4q), [n2q] "s"(n2q), [n1q] "s"(nq), [ql] "s"((u32)m.q),
              [qh] "s"((u32)(m.q >> 32))
            : PIE_ASM_CLOB, "s96", "s97");
    }
    return r;
}
// a small integer v (< 2^30) times a residue c, into the columns
__device__ __forceinline__ void colacc_mac_small(ColAcc &a, u32 v, u64 c)
{
    const Split30 s = split30(c);
    a.c0 = mad_u(v, s.lo, a.c0);
    a.c1 = mad_u(v, s.hi, a.c1);
}

// the instruction-block forms of the helpers above where every modulus is below 2^60 (ASM), the compiler's otherwise
template <bool ASM>
__device__ __forceinline__ u64 mshoup(u64 a, u64 w, u64 wsh, u64 q, u64 negq)
{
    return ASM ? shoup63(a, w, wsh, negq) : mul_shoup_u(a, w, wsh, q, negq);
}
template <bool ASM>
__device__ __forceinline__ void mdivmod(u64 a, u64 w, u64 wsh, u64 q, u64 negq, u64 &quot, u64 &rem)
{
    if (ASM)
        divmod63(a, w, wsh, negq, quot, rem);
    else
        divmod_shoup_u(a, w, wsh, q, negq, quot, rem);
}
template <bool ASM>
__device__ __forceinline__ u64 mfixfrac(u64 y, const Mod &m)
{
    return ASM ? (mulhi_sb(y << m.fshift, m.fconst) >> 3) : fixfrac_u(y, m);
}

// sum_i y[i] * c[i] as a 128-bit integer.  MAD: carry-free column accumulators on v_mad_u64_u32 (madasm.h; needs all
// operands < 2^60 and NS <= 8), otherwise 64x64->128 multiplies.
template <u32 NS, bool MAD>
__device__ __forceinline__ U128 dot128(const u64 *y, const u64 (&c)[NS])
{
    if (MAD) {
        ColAcc a = {0, 0, 0};
#pragma unroll
        for (u32 i = 0; i < NS; i++) colacc_mac(a, split30(y[i]), split30(c[i]));
        return colacc_value(a);
    }
    U128 acc = {0, 0};
#pragma unroll
    for (u32 i = 0; i < NS; i++) mac128(acc, y[i], c[i]);
    return acc;
}

// centred CRT lift of y_i-weighted residues from a source basis into target modulus `tm`:
//   sum_i y_i * hat[i] - v * prodmod
// LZ (MAD only): the result stays in [0, 4q) (colacc_reduce123_lazy)
template <u32 NS, bool MAD, bool LZ = false>
__device__ __forceinline__ u64 crt_out(const u64 *y, const u64 (&hat)[NS], u64 v, u64 prodmod, const Mod &tm, u64 negq, u64 neg2q)
{
    if (MAD && NS <= 7) {
        // MAD implies 2^59 < q < 2^60 for every modulus: up to 7 products plus the small term stay below 2^123; everything
        // goes through the column accumulator (v <= NS is one more, partial, term) and one reduction block
        ColAcc a = {0, 0, 0};
#pragma unroll
        for (u32 i = 0; i < NS; i++) colacc_mac(a, split30(y[i]), split30(hat[i]));
        colacc_mac_small(a, (u32)v, tm.q - prodmod);  // - v * prodmod (mod tm)
        return LZ ? colacc_reduce123_lazy(a, tm, negq) : colacc_reduce<false>(a, tm, negq);
    }
    U128 acc = dot128<NS, MAD>(y, hat);
    mac128(acc, v, tm.q - prodmod);  // - v * prodmod (mod tm); v <= ns: one Barrett reduction for the whole sum
    return reduce128(acc, tm);
}

// Outer-stage folding.  For N >= 2^14 the transforms next to these kernels run as two half-size slices per limb
// (two workgroups per CU, finer scheduling granularity) and the outermost radix-2 stage moves here, where both
// coefficients n and n + N/2 of every limb pass through registers anyway:
//   load side  (input comes from an inverse transform):  (u, v) -> ((u + v) N^-1, (u - v) psi^{-N/2} N^-1)
//   store side (output goes to a forward transform):     (u, v) -> (u + v psi^{N/2}, u - v psi^{N/2})
// The folded inverse transform hands over unnormalised residues in [0, 4q) and the forward transform takes anything
// below 8q, so neither side reduces sums or differences: the Shoup product accepts any 64-bit operand.
__device__ __forceinline__ void fold_load(const DevConsts *dc, u32 a, u64 u, u64 v, u64 &x0, u64 &x1)
{
    PIE_ITER_FENCE();
    const DcC c = dc_iter(dc);
    const u64 q = c->mod[a].q, nq = neg_u(q);
    x0 = shoup63(u + v, c->fold_ia[a], c->fold_ia_sh[a], nq);            // (folding implies q < 2^60)
    x1 = shoup63(u + (4 * q - v), c->fold_ib[a], c->fold_ib_sh[a], nq);
}
// u in [0, 4q) (canonical, or left unreduced by the base conversions: LZ); results in (0, 8q)
__device__ __forceinline__ void fold_store(const DevConsts *dc, u32 a, u64 u, u64 v, u64 &y0, u64 &y1)
{
    const DcC c = dc_iter(dc);
    const u64 q = c->mod[a].q;
    const u64 t = shoup63_lazy(v, c->fold_w[a], c->fold_w_sh[a], neg_u(q));  // [0, 4q)
    y0 = u + t;
    y1 = u + (4 * q - t);
}

// The RNS width L is a template parameter: with run-time trip counts hipcc indexes the per-coefficient residue
// arrays dynamically and spills them to scratch.  NP coefficients (1, or the 2 of a folded pair) go through every
// iteration together and share its constants.
// x[L] (mod Q) -> out[M]: Q limbs copied, P limbs = centred CRT lift
// YIN: x[] already holds the CRT digits y_i = [x_i (Q/q_i)^-1]_{q_i} (folded load with the merged constants); the Q
// limbs of the output are then not produced
template <u32 L, bool MAD, bool YIN, int NP, bool LZ = false>
__device__ __forceinline__ void expand_core(const DevConsts *dc, const u64 (&x)[NP][L], u64 (&out)[NP][2 * L + 1])
{
    constexpr u32 Lp = L + 1;
    u64 y[NP][L];
    u64 fsum[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) fsum[p] = 0;
#pragma unroll
    for (u32 i = 0; i < L; i++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod mi = ld_mod(c, i);
        const u64 w = c->qhat_inv[i], wsh = c->qhat_inv_sh[i], nq = neg_u(mi.q);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            out[p][i] = x[p][i];
            y[p][i] = YIN ? x[p][i] : mshoup<MAD>(x[p][i], w, wsh, mi.q, nq);
            fsum[p] += mfixfrac<MAD>(y[p][i], mi);
        }
    }
#pragma unroll
    for (u32 j = 0; j < Lp; j++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod pj = ld_mod(c, L + j);
        u64 hat[L];
#pragma unroll
        for (u32 i = 0; i < L; i++) hat[i] = c->qhat_modp[i][j];
        const u64 prodmod = c->Q_modp[j], nq = neg_u(pj.q), n2q = neg_u(2 * pj.q);
#pragma unroll
        for (int p = 0; p < NP; p++) out[p][L + j] = crt_out<L, MAD, LZ>(y[p], hat, (fsum[p] + FIX_HALF) >> 60, prodmod, pj, nq, n2q);
    }
}

// x[L] (mod Q) -> out[M]: P limbs = round(P x / Q), Q limbs = centred CRT lift of that
template <u32 L, bool MAD, bool YIN, int NP, bool LZ = false>
__device__ __forceinline__ void scale_pq_core(const DevConsts *dc, const u64 (&x)[NP][L], u64 (&out)[NP][2 * L + 1])
{
    constexpr u32 Lp = L + 1;
    u64 y[NP][L];
    u64 fsum[NP];
    U128 itot[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) fsum[p] = 0, itot[p] = U128{0, 0};
#pragma unroll
    for (u32 i = 0; i < L; i++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod mi = ld_mod(c, i);
        const u64 w = c->qhat_inv[i], wsh = c->qhat_inv_sh[i], pw = c->P_modq[i], pwsh = c->P_modq_sh[i], nq = neg_u(mi.q);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            y[p][i] = YIN ? x[p][i] : mshoup<MAD>(x[p][i], w, wsh, mi.q, nq);
            // y_i P / q_i = y_i floor(P/q_i) + floor(y_i w_i / q_i) + (y_i w_i mod q_i) / q_i
            u64 fl, z;
            mdivmod<MAD>(y[p][i], pw, pwsh, mi.q, nq, fl, z);
            add128(itot[p], U128{fl, 0});
            fsum[p] += mfixfrac<MAD>(z, mi);
        }
    }
    u64 yp[NP][Lp];
    u64 fs2[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        add128(itot[p], U128{(fsum[p] + FIX_HALF) >> 60, 0});
        fs2[p] = 0;
    }
#pragma unroll
    for (u32 j = 0; j < Lp; j++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod pj = ld_mod(c, L + j);
        u64 col[L];
#pragma unroll
        for (u32 i = 0; i < L; i++) col[i] = c->PI_modp[i][j];
        const u64 w = c->phat_inv[j], wsh = c->phat_inv_sh[j], nq = neg_u(pj.q);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            u64 r;
            if (MAD) {   // L <= 7 products + the integer parts (< (L + 1) 2^60, in column 0): below 2^123
                ColAcc a = {itot[p].lo, 0, 0};
#pragma unroll
                for (u32 i = 0; i < L; i++) colacc_mac(a, split30(y[p][i]), split30(col[i]));
                r = LZ ? colacc_reduce123_lazy(a, pj, nq) : colacc_reduce<false>(a, pj, nq);  // (LZ: feeds a Shoup product and fold_store)
            } else {
                U128 acc = dot128<L, MAD>(y[p], col);
                add128(acc, itot[p]);
                r = reduce128(acc, pj);
            }
            out[p][L + j] = r;
            yp[p][j] = mshoup<MAD>(r, w, wsh, pj.q, nq);
            fs2[p] += mfixfrac<MAD>(yp[p][j], pj);
        }
    }
#pragma unroll
    for (u32 i = 0; i < L; i++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod mi = ld_mod(c, i);
        u64 hat[Lp];
#pragma unroll
        for (u32 j = 0; j < Lp; j++) hat[j] = c->phat_modq[j][i];
        const u64 prodmod = c->P_modq[i], nq = neg_u(mi.q), n2q = neg_u(2 * mi.q);
#pragma unroll
        for (int p = 0; p < NP; p++) out[p][i] = crt_out<Lp, MAD, LZ>(yp[p], hat, (fs2[p] + FIX_HALF) >> 60, prodmod, mi, nq, n2q);
    }
}

// SKIPQ (centred lift only): leave the Q limbs of the output alone, they already hold the operand's EVALUATION form
template <bool SCALE, bool FOLD, u32 L, bool MAD, bool SKIPQ>
__device__ __forceinline__ void expand_body(const DevConsts *__restrict__ dc, u32 N, const u64 *__restrict__ in, size_t so, size_t si,
                                            u64 *__restrict__ out, u32 out_polys, u32 out_slot, u32 by)
{
    // the residues x_i themselves are not needed: the folded load multiplies by N^-1 (Q/q_i)^-1 in one go
    constexpr bool YIN = FOLD && (SCALE || SKIPQ);
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    const u32 H = N / 2;
    if (n >= (FOLD ? H : N)) return;
    const u32 o = by >> 1, c = by & 1;
    constexpr u32 M = 2 * L + 1;
    const u64 *pin = in + (size_t)o * so + (size_t)c * si + n;
    u64 *pout = out + ((size_t)(o * out_polys + out_slot + c) * M) * N + n;
    constexpr int NP = FOLD ? 2 : 1;
    u64 x[NP][L], y[NP][M];
#pragma unroll
    for (u32 i = 0; i < L; i++) {
        if (YIN) {
            PIE_ITER_FENCE();
            const DcC c = dc_iter(dc);
            const u64 q = c->mod[i].q, nq = neg_u(q), u = pin[(size_t)i * N], v = pin[(size_t)i * N + H];
            x[0][i] = shoup63(u + v, c->fold_iaq[i], c->fold_iaq_sh[i], nq);            // u, v in [0, 4q)
            x[NP - 1][i] = shoup63(u + (4 * q - v), c->fold_ibq[i], c->fold_ibq_sh[i], nq);
        } else if (FOLD) {
            fold_load(dc, i, pin[(size_t)i * N], pin[(size_t)i * N + H], x[0][i], x[NP - 1][i]);
        } else {
            x[0][i] = pin[(size_t)i * N];
        }
    }
    // folded output: every result passes through fold_store into a transform that takes [0, 8q) -- [0, 4q) will do (LZ)
    constexpr bool LZ = FOLD && MAD;
    if (SCALE)
        scale_pq_core<L, MAD, YIN, NP, LZ>(dc, x, y);
    else
        expand_core<L, MAD, YIN, NP, LZ>(dc, x, y);
#pragma unroll
    for (u32 a = 0; a < M; a++) {
        if (!SCALE && a < L && SKIPQ) continue;  // (NttExtra)
        PIE_ITER_FENCE();
        if (FOLD) {
            u64 y0, y1;
            fold_store(dc, a, y[0][a], y[NP - 1][a], y0, y1);
            pout[(size_t)a * N] = y0;
            pout[(size_t)a * N + H] = y1;
        } else {
            pout[(size_t)a * N] = y[0][a];
        }
    }
}

template <bool SCALE, bool FOLD, u32 L, bool MAD, bool SKIPQ>
__global__ void __launch_bounds__(TPB) expand_kernel(const DevConsts *__restrict__ dc, u32 N, const u64 *__restrict__ in,
                                                     size_t so, size_t si, u64 *__restrict__ out, u32 out_polys, u32 out_slot)
{
    expand_body<SCALE, FOLD, L, MAD, SKIPQ>(dc, N, in, so, si, out, out_polys, out_slot, blockIdx.y);
}
// Both operands of a ciphertext multiplication in one launch: rows [0, 2 n) of the grid scale operand Y (P/Q scaling, the
// longer body: first, so that the shorter rows fill in behind it), rows [2 n, 4 n) lift operand X.  As two launches each was
// a single round of 3.5 / 1.75 waves per SIMD with every wave in the same load - compute - store phase; together the rounds
// interleave (12.4 + 20.6 us -> one launch).  Writes out[n][4][M][N]: X into slots 0, 1, Y into slots 2, 3.
template <bool FOLD, u32 L, bool MAD, bool SKIPQ>
__global__ void __launch_bounds__(TPB) expand_both_kernel(const DevConsts *__restrict__ dc, u32 N, const u64 *__restrict__ x, size_t sx,
                                                          const u64 *__restrict__ y, size_t sy, size_t si, u64 *__restrict__ out,
                                                          u32 n_outer)
{
    if (blockIdx.y < 2 * n_outer)
        expand_body<true, FOLD, L, MAD, false>(dc, N, y, sy, si, out, 4, 2, blockIdx.y);
    else
        expand_body<false, FOLD, L, MAD, SKIPQ>(dc, N, x, sx, si, out, 4, 0, blockIdx.y - 2 * n_outer);
}
void launch_expand_both(const DevConsts *dc, u32 N, u32 L, const u64 *x, size_t sx, const u64 *y, size_t sy, size_t si, u32 n_outer,
                        u64 *out, hipStream_t st, bool fold, bool skip_q)
{
    dim3 grid(((fold ? N / 2 : N) + TPB - 1) / TPB, n_outer * 4);
#define EB(F_, L_, M_, Q_) hipLaunchKernelGGL((expand_both_kernel<F_, L_, M_, Q_>), grid, dim3(TPB), 0, st, dc, N, x, sx, y, sy, si, out, n_outer)
#define EBL(L_)                                                                          \
    case L_:                                                                             \
        if (fold && g_small_moduli) { if (skip_q) EB(true, L_, true, true); else EB(true, L_, true, false); }      \
        else if (fold) { if (skip_q) EB(true, L_, false, true); else EB(true, L_, false, false); }                 \
        else if (g_small_moduli) { if (skip_q) EB(false, L_, true, true); else EB(false, L_, true, false); }       \
        else { if (skip_q) EB(false, L_, false, true); else EB(false, L_, false, false); }                         \
        break;
    switch (L) { EBL(1) EBL(2) EBL(3) EBL(4) EBL(5) EBL(6) EBL(7) }
#undef EBL
#undef EB
}

static void launch_expand_common(bool scale, bool fold, const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si,
                                 u32 n_outer, u64 *out, u32 out_polys, u32 out_slot, hipStream_t st, bool skip_q = false)
{
    dim3 grid(((fold ? N / 2 : N) + TPB - 1) / TPB, n_outer * 2);
#define EX(S_, F_, L_, Q_)                                                                                                    \
    do {                                                                                                                      \
        if (g_small_moduli)                                                                                                   \
            hipLaunchKernelGGL((expand_kernel<S_, F_, L_, true, Q_>), grid, dim3(TPB), 0, st, dc, N, in, so, si, out, out_polys, out_slot); \
        else                                                                                                                  \
            hipLaunchKernelGGL((expand_kernel<S_, F_, L_, false, Q_>), grid, dim3(TPB), 0, st, dc, N, in, so, si, out, out_polys, out_slot); \
    } while (0)
#define EXL(L_)                                                         \
    case L_:                                                            \
        if (scale) {                                                    \
            if (fold) EX(true, true, L_, false); else EX(true, false, L_, false); \
        } else if (skip_q) {                                            \
            if (fold) EX(false, true, L_, true); else EX(false, false, L_, true); \
        } else {                                                        \
            if (fold) EX(false, true, L_, false); else EX(false, false, L_, false); \
        }                                                               \
        break;
    switch (L) { EXL(1) EXL(2) EXL(3) EXL(4) EXL(5) EXL(6) EXL(7) }
#undef EXL
#undef EX
}
void launch_expand_q_to_qp(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si, u32 n_outer, u64 *out,
                           u32 out_polys, u32 out_slot, hipStream_t st, bool fold, bool skip_q)
{
    launch_expand_common(false, fold, dc, N, L, in, so, si, n_outer, out, out_polys, out_slot, st, skip_q);
}
void launch_scale_pq_expand(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si, u32 n_outer, u64 *out,
                            u32 out_polys, u32 out_slot, hipStream_t st, bool fold)
{
    launch_expand_common(true, fold, dc, N, L, in, so, si, n_outer, out, out_polys, out_slot, st);
}

// ---------------------------------------------------------------------------------------------
// Tensor product over QP (row A5 step 4): d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1
// ---------------------------------------------------------------------------------------------
// MAD (every modulus in (2^59, 2^60)): the operands arrive lazily reduced (< 8q, from the forward transform); two sign-mask
// subtractions bring them below 2q < 2^61, the four products go through the carry-free column accumulators (30-bit low halves,
// <= 31-bit high halves: d1's two products keep every column below 2^63) and one reduction block each (z < 8 q^2 < 2^123):
// 170 VALU instructions per thread where three 128-bit products with two-word Barrett reductions took 297 -- the kernel
// was as much bound by them as by its 115 MB of traffic.
template <bool MAD>
__global__ void __launch_bounds__(TPB) tensor_kernel(const DevConsts *dc, u32 N, u32 M, const u64 *__restrict__ e,
                                                     u64 *__restrict__ d)
{
    // two adjacent coefficients per thread (16-byte lanes; the arrays are point-wise, any order): half the workgroups and waves of the
    // one-coefficient form for the same bytes -- beside the other queue group's transforms that is what counts (r05, relin_mac likewise)
    const u32 n = 2 * (blockIdx.x * TPB + threadIdx.x);
    if (n >= N) return;
    const u32 a = blockIdx.y, bin = blockIdx.z;
    const Mod m = dc->mod[a];
    const size_t MN = (size_t)M * N;
    const u64 *pe = e + (size_t)bin * 4 * MN + (size_t)a * N + n;
    u64 *pd = d + (size_t)bin * 3 * MN + (size_t)a * N + n;
    const u64x2 va0 = *reinterpret_cast<const u64x2 *>(pe), va1 = *reinterpret_cast<const u64x2 *>(pe + MN),
                vb0 = *reinterpret_cast<const u64x2 *>(pe + 2 * MN), vb1 = *reinterpret_cast<const u64x2 *>(pe + 3 * MN);
    u64x2 r0, r1, r2;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        u64 a0 = k ? va0.y : va0.x, a1 = k ? va1.y : va1.x, b0 = k ? vb0.y : vb0.x, b1 = k ? vb1.y : vb1.x;
        u64 d0, d1, d2;
        if (MAD) {
            const u64 nq = neg_u(m.q), n2q = neg_u(2 * m.q), n4q = neg_u(4 * m.q);
            a0 = csub_u(csub_u(a0, n4q), n2q), a1 = csub_u(csub_u(a1, n4q), n2q);
            b0 = csub_u(csub_u(b0, n4q), n2q), b1 = csub_u(csub_u(b1, n4q), n2q);
            const Split30 sa0 = split30(a0), sa1 = split30(a1), sb0 = split30(b0), sb1 = split30(b1);
            ColAcc c = {0, 0, 0};
            colacc_mac(c, sa0, sb0);
            d0 = colacc_reduce<false>(c, m, nq);
            c = ColAcc{0, 0, 0};
            colacc_mac(c, sa0, sb1);
            colacc_mac(c, sa1, sb0);
            d1 = colacc_reduce<false>(c, m, nq);
            c = ColAcc{0, 0, 0};
            colacc_mac(c, sa1, sb1);
            d2 = colacc_reduce<false>(c, m, nq);
        } else {
            d0 = mulmod(a0, b0, m);
            U128 x = mul128(a0, b1);
            mac128(x, a1, b0);
            d1 = reduce128(x, m);
            d2 = mulmod(a1, b1, m);
        }
        if (k) r0.y = d0, r1.y = d1, r2.y = d2;
        else r0.x = d0, r1.x = d1, r2.x = d2;
    }
    *reinterpret_cast<u64x2 *>(pd) = r0;
    *reinterpret_cast<u64x2 *>(pd + MN) = r1;
    *reinterpret_cast<u64x2 *>(pd + 2 * MN) = r2;
}
void launch_tensor(const DevConsts *dc, u32 N, u32 M, const u64 *e, u64 *d, u32 nb, hipStream_t st)
{
    dim3 grid((N / 2 + TPB - 1) / TPB, M, nb);   // (N is a power of two >= 8)
    if (g_small_moduli)
        hipLaunchKernelGGL(tensor_kernel<true>, grid, dim3(TPB), 0, st, dc, N, M, e, d);
    else
        hipLaunchKernelGGL(tensor_kernel<false>, grid, dim3(TPB), 0, st, dc, N, M, e, d);
}

// ---------------------------------------------------------------------------------------------
// Scale-and-round by t/P from QP into Q (row A6)
// ---------------------------------------------------------------------------------------------
// d[M] (mod QP) -> out[L]: round(t d / P) mod Q
// lz (MAD, L <= 5): the results stay in [0, 4q) (colacc_reduce123_lazy): for components that go through fold_store into a
// transform of the 16-coefficient kernel
// PRE: the inputs come from a folded load with the merged constants (DevConsts::fold_iap / fold_iat): the P limbs already hold
// yp_j = [d_j (QP/p_j)^-1]_{p_j}, the Q limbs [d_k t P^-1]_{q_k} -- the products this routine would form first
template <u32 L, bool MAD, int NP, bool PRE = false>
__device__ __forceinline__ void scale_round_core(const DevConsts *dc, const u64 (&d)[NP][2 * L + 1], u64 (&out)[NP][L], bool lz = false)
{
    constexpr u32 Lp = L + 1;
    u64 yp[NP][Lp];
    u64 fsum[NP];
    U128 itot[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) fsum[p] = 0, itot[p] = U128{0, 0};
#pragma unroll
    for (u32 j = 0; j < Lp; j++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod pj = ld_mod(c, L + j);
        const u64 w = c->qp_hat_inv[L + j], wsh = c->qp_hat_inv_sh[L + j], tw = c->tQ_modp[j], twsh = c->tQ_modp_sh[j], nq = neg_u(pj.q);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            yp[p][j] = PRE ? d[p][L + j] : mshoup<MAD>(d[p][L + j], w, wsh, pj.q, nq);
            u64 fl, z;
            mdivmod<MAD>(yp[p][j], tw, twsh, pj.q, nq, fl, z);
            add128(itot[p], U128{fl, 0});
            fsum[p] += mfixfrac<MAD>(z, pj);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; p++) add128(itot[p], U128{(fsum[p] + FIX_HALF) >> 60, 0});
#pragma unroll
    for (u32 k = 0; k < L; k++) {
        PIE_ITER_FENCE();
        const DcC c = dc_iter(dc);
        const Mod qk = ld_mod(c, k);
        u64 col[Lp];
#pragma unroll
        for (u32 j = 0; j < Lp; j++) col[j] = c->tQF_modq[j][k];
        const u64 tp = c->tPinv_modq[k], nq = neg_u(qk.q);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            if (MAD && L <= 6) {   // L + 2 products + the integer parts (< (L + 2) 2^60, in column 0)
                ColAcc a = {itot[p].lo, 0, 0};
#pragma unroll
                for (u32 j = 0; j < Lp; j++) colacc_mac(a, split30(yp[p][j]), split30(col[j]));
                if (PRE)
                    a.c0 += d[p][k];  // (canonical, < 2^60: column 0 holds L + 2 products' low parts and the integer parts besides)
                else
                    colacc_mac(a, split30(d[p][k]), split30(tp));
                out[p][k] = (L <= 5 && lz) ? colacc_reduce123_lazy(a, qk, nq) : colacc_reduce<(L > 5)>(a, qk, nq);
            } else {
                U128 acc = dot128<Lp, MAD>(yp[p], col);
                if (PRE)
                    add128(acc, U128{d[p][k], 0});
                else
                    mac128(acc, d[p][k], tp);
                add128(acc, itot[p]);
                out[p][k] = reduce128(acc, qk);
            }
        }
    }
}

// FOLD: inverse outer stage on load; forward outer stage on the store of components 0 and 1 (they go to a
// forward transform); component 2 is stored as plain coefficients (the digit kernel folds it per target modulus)
template <bool FOLD, u32 L, bool MAD>
__global__ void __launch_bounds__(TPB) scale_round_kernel(const DevConsts *__restrict__ dc, u32 N, const u64 *__restrict__ d,
                                                          u64 *__restrict__ out01, size_t stride01,
                                                          u64 *__restrict__ out2, size_t stride2, u32 fold_comp2)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    const u32 H = N / 2;
    if (n >= (FOLD ? H : N)) return;
    const u32 comp = blockIdx.y, bin = blockIdx.z;
    constexpr u32 M = 2 * L + 1;
    const u64 *pin = d + ((size_t)(bin * 3 + comp) * M) * N + n;
    u64 *pout = comp < 2 ? out01 + (size_t)bin * stride01 + (size_t)comp * L * N + n : out2 + (size_t)bin * stride2 + n;
    constexpr int NP = FOLD ? 2 : 1;
    u64 x[NP][M], y[NP][L];
#pragma unroll
    for (u32 a = 0; a < M; a++) {
        if (FOLD) {
            // the folded load and the first product of scale-and-round in one Shoup multiplication (merged constants)
            PIE_ITER_FENCE();
            const DcC c = dc_iter(dc);
            const u32 ak = a < L ? a : 0, aj = a < L ? 0 : a - L;
            const u64 q = c->mod[a].q, nq = neg_u(q), u = pin[(size_t)a * N], v = pin[(size_t)a * N + H];
            const u64 wa = a < L ? c->fold_iat[ak] : c->fold_iap[aj], was = a < L ? c->fold_iat_sh[ak] : c->fold_iap_sh[aj];
            const u64 wb = a < L ? c->fold_ibt[ak] : c->fold_ibp[aj], wbs = a < L ? c->fold_ibt_sh[ak] : c->fold_ibp_sh[aj];
            x[0][a] = shoup63(u + v, wa, was, nq);                // u, v in [0, 4q)
            x[NP - 1][a] = shoup63(u + (4 * q - v), wb, wbs, nq);
        } else {
            x[0][a] = pin[(size_t)a * N];
        }
    }
    // d0, d1 of the key-switch path go on through fold_store into the forward transform ([0, 8q) in): [0, 4q) will do.  d2 feeds the
    // digit lift (canonical), and with fold_comp2 the three components leave the library's lane-ordered world
    scale_round_core<L, MAD, NP, FOLD>(dc, x, y, FOLD && MAD && comp < 2 && !fold_comp2);
#pragma unroll
    for (u32 k = 0; k < L; k++) {
        PIE_ITER_FENCE();
        if (FOLD) {
            u64 y0 = y[0][k], y1 = y[NP - 1][k];
            if (comp < 2 || fold_comp2) fold_store(dc, k, y[0][k], y[NP - 1][k], y0, y1);
            pout[(size_t)k * N] = y0;
            pout[(size_t)k * N + H] = y1;
        } else {
            pout[(size_t)k * N] = y[0][k];
        }
    }
}
void launch_scale_round(const DevConsts *dc, u32 N, u32 L, const u64 *d, u32 nb, u64 *out01, size_t stride01, u64 *out2,
                        size_t stride2, hipStream_t st, bool fold, bool fold_comp2)
{
    dim3 grid(((fold ? N / 2 : N) + TPB - 1) / TPB, 3, nb);
#define SRL(L_)                                                                                                              \
    case L_:                                                                                                                 \
        if (fold && g_small_moduli)                                                                                          \
            hipLaunchKernelGGL((scale_round_kernel<true, L_, true>), grid, dim3(TPB), 0, st, dc, N, d, out01, stride01, out2, stride2, \
                               fold_comp2 ? 1u : 0u);                                                                        \
        else if (fold)                                                                                                       \
            hipLaunchKernelGGL((scale_round_kernel<true, L_, false>), grid, dim3(TPB), 0, st, dc, N, d, out01, stride01, out2, stride2, \
                               fold_comp2 ? 1u : 0u);                                                                        \
        else if (g_small_moduli)                                                                                             \
            hipLaunchKernelGGL((scale_round_kernel<false, L_, true>), grid, dim3(TPB), 0, st, dc, N, d, out01, stride01, out2, stride2, 0u); \
        else                                                                                                                 \
            hipLaunchKernelGGL((scale_round_kernel<false, L_, false>), grid, dim3(TPB), 0, st, dc, N, d, out01, stride01, out2, stride2, 0u); \
        break;
    switch (L) { SRL(1) SRL(2) SRL(3) SRL(4) SRL(5) SRL(6) SRL(7) }
#undef SRL
}

// ---------------------------------------------------------------------------------------------
// BV relinearisation (row A7): digit decomposition and key-switch accumulation
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 digit_lift(const DevConsts *dc, u32 i, u32 j, u64 v)
{
    const Mod &mj = dc->mod[j];
    const u64 qi = dc->mod[i].q;
    u64 r = (qi < 2 * mj.q) ? (v >= mj.q ? v - mj.q : v) : barrett128(0, v, mj);
    if (v > qi / 2) r = submod(r, dc->qi_modqj[i][j], mj.q);  // centred lift of the residue mod q_i
    return r;
}
template <bool FOLD>
__global__ void __launch_bounds__(TPB) digits_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, const u64 *__restrict__ d2,
                                                     size_t stride2, u64 *__restrict__ dig)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    const u32 H = N / 2;
    if (n >= (FOLD ? H : N)) return;
    const u32 i = blockIdx.y / L, j = blockIdx.y % L, bin = blockIdx.z;
    const u64 *pin = d2 + (size_t)bin * stride2 + (size_t)i * N + n;
    u64 *pout = dig + (((size_t)bin * L + i) * L + j) * N + n;
    if (FOLD) {
        u64 y0, y1;
        fold_store(dc, j, digit_lift(dc, i, j, pin[0]), digit_lift(dc, i, j, pin[H]), y0, y1);
        pout[0] = y0;
        pout[H] = y1;
    } else {
        pout[0] = digit_lift(dc, i, j, pin[0]);
    }
}
void launch_digits(const DevConsts *dc, u32 N, u32 L, const u64 *d2, size_t stride2, u32 nb, u64 *dig, hipStream_t st, bool fold)
{
    dim3 grid(((fold ? N / 2 : N) + TPB - 1) / TPB, L * L, nb);
    if (fold)
        hipLaunchKernelGGL(digits_kernel<true>, grid, dim3(TPB), 0, st, dc, N, L, d2, stride2, dig);
    else
        hipLaunchKernelGGL(digits_kernel<false>, grid, dim3(TPB), 0, st, dc, N, L, d2, stride2, dig);
}

// One thread: two adjacent coefficients (16-byte lanes) of one (bin, limb j), both ciphertext components, so a
// digit is loaded once for its two key products.  out_map (lane order -> standard) keeps pairs adjacent.
// KP > 0 (lane-ordered inputs of a register-blocked transform with T threads per slice and KP coefficient pairs per thread:
// pair k of thread tau sits at k T + tau and belongs at KP tau + k): a block takes the (256 / KP) x KP pairs of threads
// tau0 .. tau0 + 256 / KP - 1, i.e. KP contiguous runs of the lane order, and hands the results through LDS so that they leave
// as one contiguous 4 KiB run of the standard order; with the plain out_map scatter every 16-byte store lands in its own
// stretch.  KP = 16: kernels_ntt_fast.hip (32 coefficients per thread), KP = 8: ntt16_kernel.h.
// RPT = 2 (column-accumulator path only): one thread takes the same coefficient pair of TWO ciphertext rows that use the same key -- rows
// r and r + key_group -- and loads every key word once for both.  The kernel is bound by the traffic through the L1s (330 MB per step at
// the headline shape, more than half of it key words that every row re-reads from the L2: profiles/r05/stage_a_batch_layer_groups.txt
// has the model); a launch's last block of rows may have no second row (uniform branch).
template <bool MAD, int KP, int RPT = 1>
__global__ void __launch_bounds__(TPB) relin_mac_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, const u64 *__restrict__ d01,
                                                        size_t stride01, const u64 *__restrict__ dig,
                                                        const u64 *__restrict__ key0, const u64 *__restrict__ mask,
                                                        u64 *__restrict__ out, const u32 *__restrict__ out_map,
                                                        size_t key_stride, u32 key_group, u32 T, u32 mask_div, u32 nb)
{
    static_assert(RPT == 1 || MAD, "two rows per thread: the column-accumulator path");
    constexpr bool TILE = KP > 0;
    constexpr u32 TT = TILE ? TPB / (KP ? KP : 1) : 1;  // threads of the transform per tile
    __shared__ u64x2 s_tile[TILE ? 2 * RPT : 1][TILE ? TPB : 1];
    u32 n = 2 * (blockIdx.x * TPB + threadIdx.x);
    u32 std_pair = 0;  // TILE: first standard-order pair of this block's tile
    if (TILE) {
        const u32 tiles = T / TT, slice = blockIdx.x / tiles, tau0 = (blockIdx.x % tiles) * TT;
        const u32 k = threadIdx.x / TT, tt = threadIdx.x % TT;
        n = 2 * (slice * KP * T + k * T + tau0 + tt);
        std_pair = slice * KP * T + KP * tau0;
    }
    if (n >= N) return;
    const u32 j = blockIdx.y;
    // rows of this thread: RPT = 1: row blockIdx.z; RPT = 2: rows 2 k G + q and (2 k + 1) G + q of key q (k = z / G, q = z % G)
    u32 rows[RPT];
    bool has[RPT];
    if (RPT == 1) {
        rows[0] = blockIdx.z, has[0] = true;
    } else {
        const u32 k = blockIdx.z / key_group, q = blockIdx.z % key_group;
        rows[0] = 2 * k * key_group + q, has[0] = rows[0] < nb;
        rows[RPT - 1] = rows[0] + key_group, has[RPT - 1] = rows[RPT - 1] < nb;
        if (!has[0]) return;
    }
    const u32 bin = rows[0];
    const u64 *key = key0 + (size_t)(bin % key_group) * key_stride;  // one key per position in a group (EvalMerge)
    const Mod m = dc->mod[j];
    const size_t LN = (size_t)L * N;
    const u32 po = TILE ? 0 : (out_map ? out_map[n] : n);
    u64x2 res[RPT][2];
    if (MAD) {
        // column accumulators end to end: the L products, then d01 (it may arrive unnormalised, < 2^63, from the forward
        // transform) into column 0 -- (L + 8) 2^60 < 2^64 -- and one reduction block; the mask product likewise
        ColAcc a[RPT][2][2];
#pragma unroll
        for (int r = 0; r < RPT; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) a[r][c][0] = a[r][c][1] = ColAcc{0, 0, 0};
        for (u32 i = 0; i < L; i++) {
            const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)i * 2 + 0) * L + j) * N + n);
            const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)i * 2 + 1) * L + j) * N + n);
            const Split30 k0x = split30(k0.x), k0y = split30(k0.y), k1x = split30(k1.x), k1y = split30(k1.y);
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                if (r && !has[r]) continue;
                const u64x2 d = *reinterpret_cast<const u64x2 *>(dig + (((size_t)rows[r] * L + i) * L + j) * N + n);
                const Split30 dx = split30(d.x), dy = split30(d.y);
                colacc_mac(a[r][0][0], dx, k0x);
                colacc_mac(a[r][0][1], dy, k0y);
                colacc_mac(a[r][1][0], dx, k1x);
                colacc_mac(a[r][1][1], dy, k1y);
            }
        }
        const u64 nq = 0 - m.q;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            if (r && !has[r]) continue;
            u64x2 mk;
            if (mask) mk = *reinterpret_cast<const u64x2 *>(mask + (size_t)(rows[r] / mask_div) * LN + (size_t)j * N + n);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const u64x2 s = *reinterpret_cast<const u64x2 *>(d01 + (size_t)rows[r] * stride01 + (size_t)c * LN + (size_t)j * N + n);
                a[r][c][0].c0 += s.x;
                a[r][c][1].c0 += s.y;
                res[r][c].x = colacc_reduce<false>(a[r][c][0], m, nq);
                res[r][c].y = colacc_reduce<false>(a[r][c][1], m, nq);
                if (mask) {
                    ColAcc p0 = {0, 0, 0}, p1 = {0, 0, 0};
                    colacc_mac(p0, split30(res[r][c].x), split30(mk.x));
                    colacc_mac(p1, split30(res[r][c].y), split30(mk.y));
                    res[r][c].x = colacc_reduce<false>(p0, m, nq);
                    res[r][c].y = colacc_reduce<false>(p1, m, nq);
                }
            }
        }
    } else {
        u64x2 mk;
        if (mask) mk = *reinterpret_cast<const u64x2 *>(mask + (size_t)(bin / mask_div) * LN + (size_t)j * N + n);
        U128 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < 2; e++) acc[c][e] = U128{0, 0};
        for (u32 i = 0; i < L; i++) {
            const u64x2 d = *reinterpret_cast<const u64x2 *>(dig + (((size_t)bin * L + i) * L + j) * N + n);
            const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)i * 2 + 0) * L + j) * N + n);
            const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)i * 2 + 1) * L + j) * N + n);
            mac128(acc[0][0], d.x, k0.x);
            mac128(acc[0][1], d.y, k0.y);
            mac128(acc[1][0], d.x, k1.x);
            mac128(acc[1][1], d.y, k1.y);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const u64x2 s = *reinterpret_cast<const u64x2 *>(d01 + (size_t)bin * stride01 + (size_t)c * LN + (size_t)j * N + n);
            // d01 joins the sum before the reduction: it may arrive unnormalised (< 2^63) from the forward transform
            add128(acc[c][0], U128{s.x, 0});
            add128(acc[c][1], U128{s.y, 0});
            res[0][c].x = reduce128(acc[c][0], m);
            res[0][c].y = reduce128(acc[c][1], m);
            if (mask) {
                res[0][c].x = mulmod(res[0][c].x, mk.x, m);
                res[0][c].y = mulmod(res[0][c].y, mk.y, m);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (r && !has[r]) continue;
            const u64x2 v = res[r][c];
            if (TILE)
                s_tile[2 * r + c][KP * (threadIdx.x % TT) + threadIdx.x / TT] = v;  // standard offset inside the tile: KP (tau - tau0) + k
            else
                *reinterpret_cast<u64x2 *>(out + ((size_t)rows[r] * 2 + c) * LN + (size_t)j * N + po) = v;
        }
    if (TILE) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (r && !has[r]) continue;
                *reinterpret_cast<u64x2 *>(out + ((size_t)rows[r] * 2 + c) * LN + (size_t)j * N + 2 * (std_pair + threadIdx.x)) = s_tile[2 * r + c][threadIdx.x];
            }
    }
}
void launch_relin_mac(const DevConsts *dc, u32 N, u32 L, const u64 *d01, size_t stride01, const u64 *dig, const u64 *key,
                      const u64 *mask, u64 *out, u32 nb, hipStream_t st, const u32 *out_map, size_t key_stride, u32 key_group,
                      u32 sigma_T, u32 sigma_kp, u32 mask_div)
{
    if (!key_group) key_group = 1;
    if (!mask_div) mask_div = 1;
    // sigma_T: out_map is the lane order of a register-blocked transform with sigma_T threads per slice, sigma_kp pairs per thread
    const bool tile = out_map && (sigma_kp == 16 || sigma_kp == 8) && sigma_T >= TPB / sigma_kp && sigma_T % (TPB / sigma_kp) == 0 &&
                      (N / 2) % TPB == 0;
    const int kp = tile ? (int)sigma_kp : 0;
    // two rows per thread where at least two rows share a key (the column-accumulator path)
    const bool two = g_small_moduli && nb >= 2 * key_group;
    const u32 zrows = two ? ((nb + 2 * key_group - 1) / (2 * key_group)) * key_group : nb;
    dim3 grid((N / 2 + TPB - 1) / TPB, L, zrows);
#define RM(M_, K_, R_)                                                                                                              \
    hipLaunchKernelGGL((relin_mac_kernel<M_, K_, R_>), grid, dim3(TPB), 0, st, dc, N, L, d01, stride01, dig, key, mask, out, out_map, \
                       key_stride, key_group, sigma_T, mask_div, nb)
    if (two) {
        if (kp == 16) RM(true, 16, 2); else if (kp == 8) RM(true, 8, 2); else RM(true, 0, 2);
    } else if (g_small_moduli) {
        if (kp == 16) RM(true, 16, 1); else if (kp == 8) RM(true, 8, 1); else RM(true, 0, 1);
    } else {
        if (kp == 16) RM(false, 16, 1); else if (kp == 8) RM(false, 8, 1); else RM(false, 0, 1);
    }
#undef RM
}

// One word for the host: `value` lands in page-locked host memory when everything queued on the stream before it is done
// (piehip_host.cpp: "the uploads of this query have left host memory" -- an event cannot say that once kernels are queued
// behind the uploads: hipEventSynchronize on an earlier event waits for the stream's whole backlog).
__global__ void host_flag_kernel(volatile u64 *flag, u64 value)
{
    *flag = value;
    __threadfence_system();
}
void launch_host_flag(u64 *flag_dev, u64 value, hipStream_t st)
{
    hipLaunchKernelGGL(host_flag_kernel, dim3(1), dim3(1), 0, st, (volatile u64 *)flag_dev, value);
}

// ---------------------------------------------------------------------------------------------
// EvalAdd / EvalMult(ct,pt) as stand-alone element-wise kernels (rows A3, A4)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) ct_add_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ x,
                                                     const u64 *__restrict__ y, u64 *__restrict__ out)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y % L;
    const size_t o = ((size_t)blockIdx.z * 2 * L + blockIdx.y) * N + n;
    out[o] = addmod(x[o], y[o], dc->mod[l].q);
}
void launch_ct_add(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u64 *y, u64 *out, u32 nct, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, nct);
    hipLaunchKernelGGL(ct_add_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, y, out);
}
__global__ void __launch_bounds__(TPB) ct_mul_plain_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ x,
                                                           const u64 *__restrict__ pt, size_t pt_stride,
                                                           u64 *__restrict__ out, u32 pt_div)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y % L;
    const size_t o = ((size_t)blockIdx.z * 2 * L + blockIdx.y) * N + n;
    out[o] = mulmod(x[o], pt[(size_t)(blockIdx.z / pt_div) * pt_stride + (size_t)l * N + n], dc->mod[l]);
}
void launch_ct_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u64 *pt, size_t pt_stride, u64 *out,
                         u32 nct, hipStream_t st, u32 pt_div)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, nct);
    hipLaunchKernelGGL(ct_mul_plain_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, pt, pt_stride, out, pt_div ? pt_div : 1);
}

// ---------------------------------------------------------------------------------------------
// Automorphism permutation in EVALUATION format (row A9): out[r][p] = in[r][map[p]]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) permute_kernel(u32 N, const u64 *__restrict__ in, const u32 *__restrict__ map,
                                                      u64 *__restrict__ out)
{
    const u32 p = blockIdx.x * TPB + threadIdx.x;
    if (p >= N) return;
    const size_t r = (size_t)blockIdx.y * N;
    out[r + p] = in[r + map[p]];
}
void launch_permute(u32 N, const u64 *in, const u32 *map, u64 *out, u32 nrows, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, nrows);
    hipLaunchKernelGGL(permute_kernel, grid, dim3(TPB), 0, st, N, in, map, out);
}

// ---------------------------------------------------------------------------------------------
// Rotation-based PIE (FHEHIPPIE.cpp:61-77): EvalInnerProduct = ct x pt + log2 rotate-and-add steps, EvalMerge =
// mask slot 0, rotate ciphertext i by -i, add.  All steps batched over (PIE, bin).
// ---------------------------------------------------------------------------------------------
// out[i] = x[i / group] (.) pt[i / group][i % group]      (one index ciphertext against a group of plaintexts)
__global__ void __launch_bounds__(TPB) bcast_mul_plain_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ x,
                                                              size_t xs, u32 group, const u64 *__restrict__ pt, size_t ps_outer,
                                                              size_t ps_inner, u64 *__restrict__ out)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y % L, i = blockIdx.z, g = i / group, r = i % group;
    const size_t LN = (size_t)L * N;
    const size_t o = (size_t)blockIdx.y * N + n;
    out[(size_t)i * 2 * LN + o] = mulmod(x[(size_t)g * xs + o], pt[(size_t)g * ps_outer + (size_t)r * ps_inner + (size_t)l * N + n], dc->mod[l]);
}
void launch_bcast_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, size_t xs, u32 group, const u64 *pt, size_t ps_outer,
                            size_t ps_inner, u64 *out, u32 nct, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, nct);
    hipLaunchKernelGGL(bcast_mul_plain_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, xs, group, pt, ps_outer, ps_inner, out);
}
// Automorphism of ciphertext i with the EVALUATION index map maps[i % map_group]:
//   d01[i] = (acc ? x.c0 : 0) + x.c0 o map,  (acc ? x.c1 : 0)       (stays in EVALUATION format)
//   d2[i]  = x.c1 o map                                              (goes through the key switch)
// identity_first: position 0 of every group is not rotated (EvalMerge's first term): d01 = x, d2 = 0.
__global__ void __launch_bounds__(TPB) rot_prepare_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ x,
                                                          const u32 *__restrict__ maps, u32 map_group, u32 acc, u32 identity_first,
                                                          u64 *__restrict__ d01, u64 *__restrict__ d2)
{
    const u32 p = blockIdx.x * TPB + threadIdx.x;
    if (p >= N) return;
    const u32 l = blockIdx.y, i = blockIdx.z, r = i % map_group;
    const size_t LN = (size_t)L * N;
    const u64 *c0 = x + (size_t)i * 2 * LN + (size_t)l * N, *c1 = c0 + LN;
    u64 *o0 = d01 + (size_t)i * 2 * LN + (size_t)l * N, *o1 = o0 + LN, *o2 = d2 + (size_t)i * LN + (size_t)l * N;
    if (identity_first && r == 0) {
        o0[p] = c0[p];
        o1[p] = c1[p];
        o2[p] = 0;
        return;
    }
    const u32 s = maps[(size_t)r * N + p];
    const u64 q = dc->mod[l].q;
    o0[p] = acc ? addmod(c0[p], c0[s], q) : c0[s];
    o1[p] = acc ? c1[p] : 0;
    o2[p] = c1[s];
}
void launch_rot_prepare(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u32 *maps, u32 map_group, bool acc, bool identity_first,
                        u64 *d01, u64 *d2, u32 nct, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, L, nct);
    hipLaunchKernelGGL(rot_prepare_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, maps, map_group ? map_group : 1u, acc ? 1u : 0u,
                       identity_first ? 1u : 0u, d01, d2);
}
// out[g] = (sum_{r < group} x[g * group + r]) (.) pt[g]
__global__ void __launch_bounds__(TPB) sum_mul_plain_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ x, u32 group,
                                                            const u64 *__restrict__ pt, size_t pt_stride, u64 *__restrict__ out,
                                                            size_t out_stride)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y % L, g = blockIdx.z;
    const size_t LN = (size_t)L * N;
    const size_t o = (size_t)blockIdx.y * N + n;
    const Mod m = dc->mod[l];
    u64 s = 0;
    for (u32 r = 0; r < group; r++) s = addmod(s, x[((size_t)g * group + r) * 2 * LN + o], m.q);
    out[(size_t)g * out_stride + o] = mulmod(s, pt[(size_t)g * pt_stride + (size_t)l * N + n], m);
}
void launch_sum_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, u32 group, const u64 *pt, size_t pt_stride, u64 *out,
                          size_t out_stride, u32 ngroups, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, ngroups);
    hipLaunchKernelGGL(sum_mul_plain_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, group, pt, pt_stride, out, out_stride);
}

// ---------------------------------------------------------------------------------------------
// Packed encoding (row A2; MakePackedPlaintext at BatchedFHEHIPPIE.cpp:68,81)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) encode_scatter_kernel(const DevConsts *dc, u32 N, u32 M,
                                                             const int64_t *__restrict__ slots, u32 B,
                                                             const u32 *__restrict__ inv_pos, u64 *__restrict__ u)
{
    const u32 p = blockIdx.x * TPB + threadIdx.x;
    if (p >= N) return;
    const u64 t = dc->mod[M].q;
    const u32 s = inv_pos[p];
    u64 val = 0;
    if (s < B) {
        const int64_t v = slots[(size_t)blockIdx.y * B + s];
        const u64 mag = v < 0 ? (u64)(-v) : (u64)v;
        val = v < 0 ? (mag ? t - mag : 0) : mag;
    }
    u[(size_t)blockIdx.y * N + p] = val;
}
void launch_encode_scatter(const DevConsts *dc, u32 N, u32 M, const int64_t *slots, u32 B, const u32 *inv_pos, u64 *u,
                           u32 npt, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, npt);
    hipLaunchKernelGGL(encode_scatter_kernel, grid, dim3(TPB), 0, st, dc, N, M, slots, B, inv_pos, u);
}
__global__ void __launch_bounds__(TPB) encode_lift_kernel(const DevConsts *dc, u32 N, u32 L, u32 M,
                                                          const u64 *__restrict__ u, u64 *__restrict__ out)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u64 t = dc->mod[M].q;
    const u64 v = u[(size_t)blockIdx.z * N + n];
    const u64 q = dc->mod[blockIdx.y].q;
    out[((size_t)blockIdx.z * L + blockIdx.y) * N + n] = v > t / 2 ? q - (t - v) : v;  // centred lift
}
void launch_encode_lift(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *u, u64 *out, u32 npt, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, L, npt);
    hipLaunchKernelGGL(encode_lift_kernel, grid, dim3(TPB), 0, st, dc, N, L, M, u, out);
}

}  // namespace piehip
