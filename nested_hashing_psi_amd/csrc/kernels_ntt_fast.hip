// kernels_ntt_fast.hip -- register-blocked negacyclic NTT for slices of 2^12 .. 2^14 coefficients.
//
// One persistent workgroup of T = n/32 threads walks over limbs; each thread owns 32 coefficients
// (64 VGPRs) and the transform runs as three register passes with two LDS transposes between them:
//
//   forward (Cooley-Tukey)   pass A: 4 stages, thread holds rows k (stride n/16) of a column PAIR
//                                    (2 tau, 2 tau + 1): loaded straight from HBM with 16-byte lanes;
//                                    twiddles depend on k only -> scalar loads
//                            pass B: 5 stages inside blocks of n/16, stride n/512
//                            pass C: log2(n) - 9 stages inside 32 contiguous coefficients
//   inverse (Gentleman-Sande) is the mirror image: C', B', A', then the N^-1 scaling.
//
// HBM traffic is exactly one coalesced read and one coalesced write of the limb (16 n bytes).  The
// next limb's loads are issued before the current limb's passes start, so with a single resident
// workgroup per CU (a 2^14 limb fills 136 of the 160 KiB LDS) HBM latency still hides behind ALU work.
// LDS image: element e lives at e + 2 (e >> 5) (a 16-byte pad per 32 elements): the 32-contiguous
// pass-C rows (ds_read_b128, lane stride 272 B), the stride-n/512 pass-B columns and the pair-wise
// pass-A rows are all bank-conflict-free or at worst 2-way.
//
// Lane order ("sigma" layout).  The evaluation side of a transform ends (forward) or starts (inverse) with
// 32 contiguous coefficients per thread, which is uncoalesced in memory; the standard order therefore costs
// one more LDS transpose and barrier.  For arrays that never leave the library (the QP operands and the
// tensor result between the forward and inverse transforms, the key-switch digits) the kernel can instead
// store pair k of thread tau at 2 (k T + tau): coalesced, no transpose.  Point-wise kernels do not care
// about the order; key and mask get a sigma-ordered copy at load time (ntt_sigma_inverse_map).
//
// The ALU, not HBM, bounds this kernel: a 60-bit Shoup butterfly is ~10 v_mad_u64_u32 plus ~10 32-bit
// adds/selects, 7 butterflies per 16 bytes moved (DESIGN.md "NTT roofline").
#include <vector>

#include "kernels.hpp"
#include "madasm.h"

namespace piehip {

struct NttFastArgs {
    u64 *data;
    const u64 *twp;  // interleaved {w, w_shoup} pairs: per modulus 2 tables (fwd, inv) of N pairs
    const u64 *twc;  // the same pairs in pass-C kernel order (see build_twc_table)
    const DevConsts *dc;
    u32 N, logN;
    u32 s0;       // log2 slices per limb (stages below 2^s0 groups were done in global memory)
    u32 nitems;   // limbs << s0
    u32 sigma;    // EVALUATION side stored in lane order (see below) instead of standard bit-reversed order
    // forward only, s0 = 0: BV digit lift fused into the load (replaces the digits kernel).  Item (bin, i, j)
    // reads residue limb i of the COEFFICIENT polynomial lift_src[bin][i], lifts it (centred) into q_j and
    // transforms it; lift_L = 0 disables.
    const u64 *lift_src;
    u32 lift_L;
    u32 sigma_split;  // forward, lane order: store as if the limb were 2^sigma_split slices (the folded layout)
    u32 mod_base, mod_count;
    // inverse, standard order in: also write the lane-ordered EVALUATION input of operand polynomials to the Q limbs of
    // the QP operand array (the tensor product then needs no forward transform for them).  Limb li of the input
    // [nb][copy_K][2][copy_L] is copied iff it belongs to operand 0 of its bin: -> copy_out[bin][4][copy_M][N], slot c.
    u64 *copy_out;
    u32 copy_K, copy_L, copy_M;
    // forward: the transformed limbs are a compact enumeration of [nb][4][skip_M] without limbs < skip_L of slots 0, 1
    u32 skip_L, skip_M;
    u32 folded;  // s0 != 0 because the caller's neighbouring kernels apply the outermost stage (not the global-memory stages)
};

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 phi(u32 e) { return e + 2 * (e >> 5); }

// ---- 60-bit Shoup multiplication on the 32-bit multiplier ------------------------------------------
// v_mad_u64_u32 (32x32+64 -> 64) issues at twice the rate of v_mul_lo_u32 / v_mul_hi_u32 on gfx950, and
// hipcc narrows every 64-bit product whose high half is unused to v_mul_lo_u32; inline asm keeps
// the whole butterfly on the mad.  9 mads per modular multiplication:
//   quotient estimate  qe = floor(b ws / 2^64) - {0,1,2}   from 3 partial products (b_lo ws_lo dropped)
//   remainder          b w + qe (2^64 - q)  mod 2^64       as two accumulation chains (low word, cross terms)
// ws here is the 63-bit constant floor(w 2^63 / q) (= the usual Shoup constant >> 1) so that the sum of the
// two cross products cannot overflow 64 bits for b < 2^63; qe = 2 bh sh + (bh sl + bl sh) >> 31 under-
// estimates floor(b w / q) by at most 3.  The result lies in [0, 4q); with q < 2^60 the butterflies keep
// residues in [0, 8q) (forward) or [0, 4q) (inverse) and normalise once at the end of the transform.
// b < 2^63, w < q, ws = floor(w 2^63 / q), nq = 2^64 - q: returns b w mod q + {0,1,2,3} q
__device__ __forceinline__ u64 shoup4(u64 b, u64 w, u64 ws, u64 nq)
{
    const u32 bl = (u32)b, bh = (u32)(b >> 32), wl = (u32)w, wh = (u32)(w >> 32);
    const u32 sl = (u32)ws, sh = (u32)(ws >> 32), nql = (u32)nq, nqh = (u32)(nq >> 32);
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

// Harvey butterflies on lazy residues ------------------------------------------------------------------------
// forward: inputs in [0, 8q), outputs in [0, 8q)
__device__ __forceinline__ void ct_bfly(u64 &a, u64 &b, u64 w, u64 ws, u64 nq, u64 q4)
{
    const u64 u = a >= q4 ? a - q4 : a;
    const u64 v = shoup4(b, w, ws, nq);
    a = u + v;
    b = u - v + q4;
}
// inverse: inputs in [0, 4q), outputs in [0, 4q)
__device__ __forceinline__ void gs_bfly(u64 &a, u64 &b, u64 w, u64 ws, u64 nq, u64 q4)
{
    const u64 s = a + b;
    const u64 d = a - b + q4;
    a = s >= q4 ? s - q4 : s;
    b = shoup4(d, w, ws, nq);
}
// m-th index in [0,32) whose bit `d` (a power of two) is clear
__device__ __forceinline__ constexpr int bfly_lo(int m, int d) { return ((m & ~(d - 1)) << 1) | (m & (d - 1)); }

// LDS hand-off inside one wave: DS operations of a wave execute in issue order, so only the compiler has to
// be kept from moving the reads above the writes
#define WAVE_LOCAL_SYNC()                                   \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        __builtin_amdgcn_wave_barrier();                    \
    } while (0)

// keeps hipcc from interleaving more than a few butterflies (each carries ~12 VGPRs of temporaries)
#define FENCE_EVERY 16
// in-kernel cycle stamps are a tooling build (tools/ntt_stamps.hip defines STAMP before including this file)
#ifndef STAMP
#define STAMP(i)
#endif
#define BFLY_FENCE(cnt, every)                                         \
    do {                                                               \
        if (((cnt) % (every)) == (every) - 1) __builtin_amdgcn_sched_barrier(0); \
    } while (0)

// centred lift of a residue mod q_i into q_j (same rule as digits_kernel / oracle keyswitch_acc)
__device__ __forceinline__ u64 lift_digit(u64 v, u64 qi, u64 qi_mod_qj, const Mod &mj)
{
    // v < q_i; when q_i < 2 q_j (every chain of equal-width primes) one conditional subtraction reduces it
    u64 r = (qi < 2 * mj.q) ? (v >= mj.q ? v - mj.q : v) : barrett128(0, v, mj);
    if (v > qi / 2) r = submod(r, qi_mod_qj, mj.q);
    return r;
}

// LAZY (forward, lane order): residues leave in [0, 8q) instead of [0, q) -- for consumers that reduce anyway (the
// tensor product's 128-bit Barrett, the key-switch MAC's accumulator); saves three conditional subtractions per coefficient
template <int LOGN, bool INV, bool SIGMA, bool LAZY = false>
__global__ void __launch_bounds__((1 << LOGN) / 32)
ntt_fast_kernel(u64 *__restrict__ gdata, const u64x2 *__restrict__ gtw, const u64x2 *__restrict__ gtwc,
                const DevConsts *__restrict__ gdc, NttFastArgs a)
{
    constexpr u32 n = 1u << LOGN;   // coefficients in this slice
    constexpr u32 T = n / 32;       // threads
    constexpr u32 NB = n / 16;      // pass-B block size = pass-A row stride
    constexpr u32 SB = NB / 32;     // pass-B stride
    constexpr int C = LOGN - 9;     // pass-C stages
    extern __shared__ __attribute__((aligned(16))) u64 lds_real[];
    u64 *lds = lds_real;

    const u32 tau = threadIdx.x;
    const u32 beta = tau / SB, rho = tau % SB;
    u64 x[32], y[32];

    u32 item = blockIdx.x;
    if (item >= a.nitems) return;
    // slice index in memory of work item `it` (forward: compact enumeration that skips the Q limbs of slots 0 and 1)
    auto slice_of = [&](u32 it) -> size_t {
        u32 lb = it >> a.s0;
        if (!INV && a.skip_L) {
            const u32 P = a.skip_M - a.skip_L, per = 2 * P + 2 * a.skip_M;
            const u32 cb = lb / per, r = lb % per;
            lb = cb * 4 * a.skip_M + (r < 2 * P ? (r / P) * a.skip_M + a.skip_L + r % P : 2 * a.skip_M + (r - 2 * P));
        }
        return ((size_t)lb << a.s0) + (it & ((1u << a.s0) - 1));
    };
    // prefetch the first slice (A-layout addresses are also the coalesced copy-in/out order)
    {
        const u64 *g = gdata + slice_of(item) * n;
        const u32 st = (INV && SIGMA) ? 2 * T : NB;  // sigma order: pair k of thread tau at 2 (k T + tau)
        if (!INV && a.lift_L) g = a.lift_src + (size_t)(item / a.lift_L) * n;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64x2 v = *reinterpret_cast<const u64x2 *>(g + 2 * tau + st * k);
            y[2 * k] = v.x;
            y[2 * k + 1] = v.y;
        }
        if (!INV && a.lift_L) {
            const u32 li = (item / a.lift_L) % a.lift_L, lj = item % a.lift_L;
            const Mod mj = gdc->mod[lj];
            const u64 qi = gdc->mod[li].q, qq = gdc->qi_modqj[li][lj];
#pragma unroll
            for (int k = 0; k < 32; k++) y[k] = lift_digit(y[k], qi, qq, mj);
        }
    }
    int iter = 0;
    for (; item < a.nitems; item += gridDim.x, iter++) {
        STAMP(0);
        if (!INV && a.lift_L && iter > 0) {  // digits prefetched during the previous slice: lift them now
            const u32 li = (item / a.lift_L) % a.lift_L, lj = item % a.lift_L;
            const Mod mj = gdc->mod[lj];
            const u64 qi = gdc->mod[li].q, qq = gdc->qi_modqj[li][lj];
#pragma unroll
            for (int k = 0; k < 32; k++) y[k] = lift_digit(y[k], qi, qq, mj);
        }
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = y[k];
        const u32 next = item + gridDim.x;
        // forward: the next slice's loads fly during the whole transform (pass A has scalar twiddles, so
        // the 64 extra VGPRs fit); inverse: they are issued before pass A' instead (see below)
        const size_t slice = slice_of(item);
        const u32 limb = (u32)(slice >> a.s0), blk = item & ((1u << a.s0) - 1);
        const u32 mod = a.mod_base + limb % a.mod_count;
        const Mod m = gdc->mod[mod];
        const u64 q = m.q, q2 = 2 * m.q, q4 = 4 * m.q, nq = 0 - m.q;
        // twiddle pairs of this modulus and direction; global group index of local group i at a stage
        // with ml local groups: (ml << s0) + blk * ml + i
        const u64x2 *__restrict__ tw = gtw + ((size_t)mod * 2 + (INV ? 1 : 0)) * a.N;
        // pass-C twiddles in kernel order [blk][stage][j][tau]: lane-consecutive 16-byte loads (the natural
        // order would touch 64 cache lines per wave instruction in the last stage)
        constexpr u32 JT = 32 - (32 >> C);  // twiddles per thread over the C stages
        const u64x2 *__restrict__ twc = gtwc + (((size_t)mod * 2 + (INV ? 1 : 0)) * (1u << a.s0) + blk) * (T * JT) + tau;
#define TWL(idx) tw[idx]
#define TWC(sc, j) twc[T * ((32u >> C) * ((1u << (sc)) - 1) + (j))]
        u64 *g = gdata + slice * n;

        STAMP(1);
        if (!INV) {
            // ---- pass A: stages with 1, 2, 4, 8 local groups; rows k, k + d ------------------------
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int d = 8 >> s;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k & d) continue;
                    const u32 ml = 1u << s, i = (u32)k >> (4 - s);
                    const u64x2 w = TWL((ml << a.s0) + blk * ml + i);  // uniform: scalar load
#pragma unroll
                    for (int j = 0; j < 2; j++) ct_bfly(x[2 * k + j], x[2 * (k + d) + j], w.x, w.y, nq, q4);
                }
                if (s == 0) {
                    // Prefetch the next slice here: the previous slice's stores have had one stage to drain (the
                    // vector-memory queue is in order, loads issued right behind 128 KiB of stores stall at issue),
                    // pass A needs no vector loads (scalar twiddles), and the data is home before pass B's twiddle
                    // loads queue up behind it.
                    __builtin_amdgcn_sched_barrier(0);
                    if (next < a.nitems) {
                        const u64 *gn = a.lift_L ? a.lift_src + (size_t)(next / a.lift_L) * n : gdata + slice_of(next) * n;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const u64x2 v = *reinterpret_cast<const u64x2 *>(gn + 2 * tau + NB * k);
                            y[2 * k] = v.x;
                            y[2 * k + 1] = v.y;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            STAMP(2);
            __syncthreads();  // every wave has finished the previous slice's copy-out reads
#pragma unroll
            for (int k = 0; k < 16; k++) {
                u64x2 v;
                v.x = x[2 * k];
                v.y = x[2 * k + 1];
                *reinterpret_cast<u64x2 *>(&lds[phi(2 * tau + NB * k)]) = v;
            }
            STAMP(3);
            __syncthreads();
            STAMP(4);
            // ---- pass B: 16 << sb local groups; columns rho + SB k of block beta ---------------------
#pragma unroll
            for (int k = 0; k < 32; k++) x[k] = lds[phi(NB * beta + rho + SB * k)];
            STAMP(5);
#pragma unroll
            for (int sb = 0; sb < 5; sb++) {
                const int d = 16 >> sb;
#pragma unroll
                for (int mm = 0; mm < 16; mm++) {
                    const int k = bfly_lo(mm, d);
                    const u32 ml = 16u << sb;
                    const u64x2 w = TWL((ml << a.s0) + blk * ml + (beta << sb) + ((u32)k >> (5 - sb)));
                    ct_bfly(x[k], x[k + d], w.x, w.y, nq, q4);
                }
            }
            STAMP(6);
#pragma unroll
            for (int k = 0; k < 32; k++) lds[phi(NB * beta + rho + SB * k)] = x[k];
            STAMP(7);
            WAVE_LOCAL_SYNC();  // a wave's pass-B blocks are exactly its pass-C rows (2048 contiguous elements)
            STAMP(8);
            // ---- pass C: 512 << sc local groups; 32 contiguous coefficients ---------------------------
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u64x2 v = *reinterpret_cast<const u64x2 *>(&lds[phi(32 * tau + 2 * k)]);
                x[2 * k] = v.x;
                x[2 * k + 1] = v.y;
            }
            STAMP(9);
#pragma unroll
            for (int sc = 0; sc < C; sc++) {
                const int d = 1 << (C - 1 - sc);
#pragma unroll
                for (int mm = 0; mm < 16; mm++) {
                    const int k = bfly_lo(mm, d);
                    const u64x2 w = TWC(sc, (u32)k >> (C - sc));
                    ct_bfly(x[k], x[k + d], w.x, w.y, nq, q4);
                }
            }
            STAMP(10);
            if (SIGMA) {  // lane order: straight from registers, coalesced
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    u64x2 v;
                    u64 r0 = x[2 * k], r1 = x[2 * k + 1];
                    if (!LAZY) {
                        r0 = r0 >= q4 ? r0 - q4 : r0;
                        r1 = r1 >= q4 ? r1 - q4 : r1;
                        r0 = r0 >= q2 ? r0 - q2 : r0;
                        r1 = r1 >= q2 ? r1 - q2 : r1;
                        r0 = r0 >= q ? r0 - q : r0;
                        r1 = r1 >= q ? r1 - q : r1;
                    }
                    v.x = r0;
                    v.y = r1;
                    // lane order of a limb transformed as 2^sigma_split slices: slice tau / Ts, thread tau % Ts
                    const u32 Ts = T >> a.sigma_split;
                    *reinterpret_cast<u64x2 *>(g + (size_t)(tau / Ts) * (n >> a.sigma_split) + 2 * (k * Ts + tau % Ts)) = v;
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                u64x2 v;
                u64 r0 = x[2 * k], r1 = x[2 * k + 1];
                r0 = r0 >= q4 ? r0 - q4 : r0;
                r1 = r1 >= q4 ? r1 - q4 : r1;
                r0 = r0 >= q2 ? r0 - q2 : r0;
                r1 = r1 >= q2 ? r1 - q2 : r1;
                v.x = r0 >= q ? r0 - q : r0;
                v.y = r1 >= q ? r1 - q : r1;
                *reinterpret_cast<u64x2 *>(&lds[phi(32 * tau + 2 * k)]) = v;
            }
            STAMP(11);
            __syncthreads();
            STAMP(12);
            // ---- coalesced copy-out -------------------------------------------------------------------
#pragma unroll
            for (int k = 0; k < 16; k++)
                *reinterpret_cast<u64x2 *>(g + 2 * tau + NB * k) = *reinterpret_cast<const u64x2 *>(&lds[phi(2 * tau + NB * k)]);
            STAMP(13);
            STAMP(14);
        } else {
            // ---- standard order: copy-in through LDS to reach the 32-contiguous layout (lane order has it) -
            if (!SIGMA) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    u64x2 v;
                    v.x = x[2 * k];
                    v.y = x[2 * k + 1];
                    *reinterpret_cast<u64x2 *>(&lds[phi(2 * tau + NB * k)]) = v;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u64x2 v = *reinterpret_cast<const u64x2 *>(&lds[phi(32 * tau + 2 * k)]);
                    x[2 * k] = v.x;
                    x[2 * k + 1] = v.y;
                }
                // the registers now hold this slice of the EVALUATION input in lane order: operand-0 polynomials keep a
                // copy as the Q limbs of the QP operand array
                if (a.copy_out && (limb / (2 * a.copy_L)) % a.copy_K == 0) {
                    const u32 bin = limb / (2 * a.copy_L * a.copy_K), c = (limb / a.copy_L) & 1, i = limb % a.copy_L;
                    u64 *co = a.copy_out + ((((size_t)bin * 4 + c) * a.copy_M + i) << a.s0) * n + (size_t)blk * n;
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        u64x2 v;
                        v.x = x[2 * k];
                        v.y = x[2 * k + 1];
                        *reinterpret_cast<u64x2 *>(co + 2 * (k * T + tau)) = v;
                    }
                }
            }
            // ---- pass C': distances 1, 2, .. 2^(C-1) -----------------------------------------------------------
#pragma unroll
            for (int sc = C - 1; sc >= 0; sc--) {
                const int d = 1 << (C - 1 - sc);
#pragma unroll
                for (int mm = 0; mm < 16; mm++) {
                    const int k = bfly_lo(mm, d);
                    const u64x2 w = TWC(sc, (u32)k >> (C - sc));
                    gs_bfly(x[k], x[k + d], w.x, w.y, nq, q4);
                    BFLY_FENCE(mm, FENCE_EVERY);
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                u64x2 v;
                v.x = x[2 * k];
                v.y = x[2 * k + 1];
                *reinterpret_cast<u64x2 *>(&lds[phi(32 * tau + 2 * k)]) = v;
            }
            WAVE_LOCAL_SYNC();
            // ---- pass B' -----------------------------------------------------------------------------------------
#pragma unroll
            for (int k = 0; k < 32; k++) x[k] = lds[phi(NB * beta + rho + SB * k)];
#pragma unroll
            for (int sb = 4; sb >= 0; sb--) {
                const int d = 16 >> sb;
#pragma unroll
                for (int mm = 0; mm < 16; mm++) {
                    const int k = bfly_lo(mm, d);
                    const u32 ml = 16u << sb;
                    const u64x2 w = TWL((ml << a.s0) + blk * ml + (beta << sb) + ((u32)k >> (5 - sb)));
                    gs_bfly(x[k], x[k + d], w.x, w.y, nq, q4);
                    BFLY_FENCE(mm, FENCE_EVERY);
                }
            }
#pragma unroll
            for (int k = 0; k < 32; k++) lds[phi(NB * beta + rho + SB * k)] = x[k];
            __syncthreads();
            // ---- pass A' + scaling, stored straight to HBM -------------------------------------------------------
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u64x2 v = *reinterpret_cast<const u64x2 *>(&lds[phi(2 * tau + NB * k)]);
                x[2 * k] = v.x;
                x[2 * k + 1] = v.y;
            }
            __syncthreads();  // the next slice's copy-in overwrites the image
            if (next < a.nitems) {  // prefetch: latency hides behind pass A' (scalar twiddles, low pressure)
                const u64 *gn = gdata + (size_t)next * n;
                const u32 st = SIGMA ? 2 * T : NB;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u64x2 v = *reinterpret_cast<const u64x2 *>(gn + 2 * tau + st * k);
                    y[2 * k] = v.x;
                    y[2 * k + 1] = v.y;
                }
            }
#pragma unroll
            for (int s = 3; s >= 0; s--) {
                const int d = 8 >> s;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k & d) continue;
                    const u32 ml = 1u << s, i = (u32)k >> (4 - s);
                    const u64x2 w = TWL((ml << a.s0) + blk * ml + i);
#pragma unroll
                    for (int j = 0; j < 2; j++) gs_bfly(x[2 * k + j], x[2 * (k + d) + j], w.x, w.y, nq, q4);
                    BFLY_FENCE(k, FENCE_EVERY / 2);
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                u64x2 v;
                if (a.s0 == 0) {
                    v.x = mul_shoup(x[2 * k], m.n_inv, m.n_inv_sh, q);
                    v.y = mul_shoup(x[2 * k + 1], m.n_inv, m.n_inv_sh, q);
                } else if (a.folded) {  // the consumer applies the outermost stage and N^-1 with a Shoup product, which
                                        // takes the unnormalised [0, 4q) residues as they are (kernels_pie.hip fold_load)
                    v.x = x[2 * k];
                    v.y = x[2 * k + 1];
                } else {  // split transform of a ring above 2^14: the global-memory stages expect canonical residues
                    u64 r0 = x[2 * k], r1 = x[2 * k + 1];
                    r0 = r0 >= q2 ? r0 - q2 : r0;
                    r1 = r1 >= q2 ? r1 - q2 : r1;
                    v.x = r0 >= q ? r0 - q : r0;
                    v.y = r1 >= q ? r1 - q : r1;
                }
                *reinterpret_cast<u64x2 *>(g + 2 * tau + NB * k) = v;
            }
        }
    }
}

template <int LOGN, bool INV, bool SIGMA, bool LAZY = false>
static void launch_one(const NttFastArgs &a, u32 max_groups, hipStream_t st)
{
    constexpr u32 n = 1u << LOGN;
    constexpr size_t lds = (size_t)(n + n / 16) * sizeof(u64);
    static PerDeviceOnce attr;
    if (attr.first_on_current_device())
        (void)hipFuncSetAttribute((const void *)ntt_fast_kernel<LOGN, INV, SIGMA, LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    u32 grid = a.nitems < max_groups ? a.nitems : max_groups;
    hipLaunchKernelGGL((ntt_fast_kernel<LOGN, INV, SIGMA, LAZY>), dim3(grid), dim3(n / 32), lds, st, a.data, reinterpret_cast<const u64x2 *>(a.twp), reinterpret_cast<const u64x2 *>(a.twc), a.dc, a);
}

// returns false if this slice size has no register-blocked kernel
// Host side: pass-C twiddles of one (modulus, direction) in kernel order [blk][stage][j][tau].
// nat[k] = {w, w_shoup} pairs in natural (bit-reversed exponent) order, N pairs.
void build_twc_table(const u64 *nat, u32 logN, u32 s0, std::vector<u64> &out)
{
    const u32 logn = logN - s0, n = 1u << logn, T = n / 32;
    const int C = (int)logn - 9;
    const u32 JT = 32 - (32 >> C);
    out.assign((size_t)(1u << s0) * T * JT * 2, 0);
    for (u32 blk = 0; blk < (1u << s0); blk++)
        for (int sc = 0; sc < C; sc++) {
            const u32 ml = 512u << sc, J = 1u << (5 - C + sc);
            for (u32 j = 0; j < J; j++)
                for (u32 tau = 0; tau < T; tau++) {
                    const u32 i = (tau << (5 - C + sc)) + j;
                    const size_t src = (size_t)((ml << s0) + blk * ml + i) * 2;
                    const size_t dst = (((size_t)blk * T * JT) + (size_t)T * ((32u >> C) * ((1u << sc) - 1) + j) + tau) * 2;
                    out[dst] = nat[src];
                    out[dst + 1] = nat[src + 1];
                }
        }
}

// lane-order position p (within the limb) -> standard position; identity when the register-blocked kernel
// does not apply to this ring dimension
void ntt_sigma_inverse_map(u32 logN, u32 s0, std::vector<u32> &map)
{
    const u32 N = 1u << logN;
    map.resize(N);
    if (s0 == ~0u) {
        for (u32 p = 0; p < N; p++) map[p] = p;
        return;
    }
    const u32 n = N >> s0, T = n / 32;
    for (u32 p = 0; p < N; p++) {
        const u32 blk = p / n, pl = p % n;
        const u32 tau = (pl >> 1) % T, k = (pl >> 1) / T;
        map[p] = blk * n + 32 * tau + 2 * k + (pl & 1);
    }
}

bool launch_ntt_fast(const u64 *twp, const u64 *twc, const DevConsts *dc, u32 N, u32 logN, u32 s0, u64 *data, u32 nlimbs, u32 mod_base,
                     u32 mod_count, bool inverse, bool sigma, u32 num_cus, hipStream_t st, const u64 *lift_src, u32 lift_L,
                     u32 sigma_split, const NttExtra *ex)
{
    const u32 logn = logN - s0;
    if (logn < 12 || logn > 14) return false;
    NttFastArgs a;
    a.data = data;
    a.twp = twp;
    a.twc = twc;
    a.dc = dc;
    a.N = N;
    a.logN = logN;
    a.s0 = s0;
    a.nitems = nlimbs << s0;
    a.sigma = sigma ? 1u : 0u;
    a.lift_src = lift_src;
    a.lift_L = (lift_src && !inverse && s0 == 0) ? lift_L : 0;
    a.sigma_split = (!inverse && s0 == 0) ? sigma_split : 0;
    a.mod_base = mod_base;
    a.mod_count = mod_count;
    a.copy_out = (ex && inverse && !sigma) ? ex->copy_out : nullptr;
    a.copy_K = ex ? ex->copy_K : 1;
    a.copy_L = ex ? ex->copy_L : 1;
    a.copy_M = ex ? ex->copy_M : 1;
    a.skip_L = (ex && !inverse) ? ex->skip_L : 0;
    a.skip_M = ex ? ex->skip_M : 0;
    a.folded = (ex && ex->folded) ? 1u : 0u;
    const bool lazy = ex && ex->lazy_out && !inverse && sigma && !a.lift_L;
    // resident workgroups per CU by LDS: 136 KiB -> 1, 68 KiB -> 2, 34 KiB -> 4
    const u32 per_cu = logn == 14 ? 1 : (logn == 13 ? 2 : 4);
    const u32 maxg = num_cus * per_cu;
#define NTT_DISPATCH(LG)                                                          \
    do {                                                                          \
        if (inverse) {                                                            \
            if (sigma) launch_one<LG, true, true>(a, maxg, st);                   \
            else launch_one<LG, true, false>(a, maxg, st);                        \
        } else {                                                                  \
            if (sigma && lazy) launch_one<LG, false, true, true>(a, maxg, st);    \
            else if (sigma) launch_one<LG, false, true>(a, maxg, st);             \
            else launch_one<LG, false, false>(a, maxg, st);                       \
        }                                                                         \
    } while (0)
    if (logn == 14) NTT_DISPATCH(14);
    else if (logn == 13) NTT_DISPATCH(13);
    else NTT_DISPATCH(12);
#undef NTT_DISPATCH
    return true;
}

}  // namespace piehip
