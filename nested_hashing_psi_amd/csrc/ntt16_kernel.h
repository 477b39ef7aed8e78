// ntt16_kernel.h -- negacyclic NTT of 2^13-coefficient slices, 16 coefficients per thread (device code, gfx950).
//
// The 60-bit Shoup butterfly is bound by the integer ALU (~100 SIMD cycles per wave64 butterfly), and a
// wave of a transform spends 40-50 % of its life parked at LDS hand-offs, barriers and twiddle loads.  A SIMD issues one VALU
// instruction every 2 cycles only when at least two of its waves have one ready, so two waves per SIMD (all that 32
// coefficients per thread leave room for: the LDS image of a slice is 8 bytes per coefficient) keep it about half busy.
// This kernel halves the per-thread footprint: 512 threads (8 waves) per slice, two slices per CU = 4 waves per SIMD.
//
// Forward (Cooley-Tukey, natural order in, bit-reversed out), stage s pairs elements that differ in bit 12 - s:
//   pass 1  stages 0-2   thread tau holds rows r (stride 1024) of the column pair (2 tau, 2 tau + 1): loaded straight from
//                        HBM with 16-byte lanes; twiddles depend on r only -> scalar loads
//   -- LDS transpose, the only workgroup barrier pair of a slice --
//   pass 2  stages 3-6   wave w, lane l holds e = 1024 w + 64 k + l: twiddles depend on (w, k) only -> scalar loads
//   -- LDS hand-off inside the wave (a wave's pass-2 elements are its pass-3 and pass-4 elements) --
//   pass 3  stages 7-10  lane (a, c) holds e = 1024 w + 64 a + 4 k + c; per-lane twiddles from a kernel-ordered table
//   -- LDS hand-off inside the wave --
//   pass 4  stages 11-12 lane l holds the 16 contiguous coefficients e = 1024 w + 16 l + k
// Inverse (Gentleman-Sande) is the mirror image.  HBM traffic: one coalesced read and one coalesced write of the slice.
// LDS image: element e at e + 2 (e >> 5) (16 bytes of padding per 256): every access pattern above is conflict-free.
//
// Lane order.  The evaluation side ends (forward) / starts (inverse) with 16 contiguous coefficients per thread; arrays
// that never leave the library keep them as the threads hold them: pair j (coefficients 2j, 2j + 1 of the 16) of thread
// tau at 2 (512 j + tau) -- coalesced 16-byte lanes, no transpose.  Standard (bit-reversed) order costs one more LDS
// round trip and is offered on the inverse's input side only (the accumulators of stage A arrive in it).
#pragma once
#include <vector>

#include "madasm.h"
#include "params.hpp"

namespace piehip {
namespace ntt16 {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// Slices of 2^13 (512 threads, two workgroups per CU) or 2^14 coefficients (1024 threads, one workgroup per CU: rings of
// 2^15 as two folded slices per limb).  The wave-local passes 2-4 are the same; pass 1 covers the stages above a wave's 1024
// elements: 3 stages on 8 rows x 2 columns per thread (16-byte lanes), or 4 stages on 16 rows x 1 column (8-byte lanes).
template <u32 LOGNS>
struct Geo {
    static constexpr u32 LOGN = LOGNS;
    static constexpr u32 NS = 1u << LOGNS;     // coefficients per slice
    static constexpr u32 T = NS / 16;          // threads per slice
    static constexpr u32 W = T / 64;           // waves per slice = 1024-element blocks
    static constexpr u32 R = NS / 1024;        // rows of pass 1 (stride 1024)
    static constexpr u32 LOGR = LOGNS - 10;    // stages of pass 1
    static constexpr u32 CPT = 16 / R;         // columns per thread in pass 1
    static constexpr u32 LDS_WORDS = NS + NS / 16;
    static constexpr u32 TWK_PER_SLICE = W * 15 * 16 + W * 12 * 64;  // kernel-ordered twiddle pairs of passes 3 and 4
};
// (the 2^13 geometry under its old names: tools/ntt_lab.hip)
static constexpr u32 LOGN = Geo<13>::LOGN, NS = Geo<13>::NS, T = Geo<13>::T, LDS_WORDS = Geo<13>::LDS_WORDS,
                     TWK_PER_SLICE = Geo<13>::TWK_PER_SLICE;

__device__ __forceinline__ u32 phi(u32 e) { return e + 2 * (e >> 5); }

struct Tw {  // {w, floor(w 2^63 / q)} split into 32-bit halves; sh2 = 2 sh (sh < 2^31) for the quotient estimate
    u32 wl, wh, sl, sh, sh2;
};
__device__ __forceinline__ Tw make_tw(u64x2 p)
{
    Tw t;
    t.wl = (u32)p.x, t.wh = (u32)(p.x >> 32), t.sl = (u32)p.y, t.sh = (u32)(p.y >> 32);
    t.sh2 = t.sh << 1;  // wave-uniform twiddles: a scalar shift; per-lane ones: one shift per twiddle, not per butterfly
    return t;
}

// per-modulus constants of the butterflies (wave-uniform: SGPRs)
struct ModC {
    u32 nql, nqh;  // 2^64 - q
    u64 nq4;       // 2^64 - 4q
    u64 q4;        // 4q
};

// ---- the 60-bit lazy butterfly as hand-scheduled instruction blocks ---------------------------------------------------------
// Shoup product b w mod q + {0,1,2,3} q with the 63-bit constant ws = floor(w 2^63 / q) (b < 2^63, w < q < 2^60):
//   quotient estimate  qe = 2 bh sh + ((bh sl + bl sh) >> 31)           3 multiplier ops (bl sl dropped, error <= 3)
//   remainder          b w + qe (2^64 - q)  mod 2^64                    6 multiplier ops on two accumulation chains
// (derivation in kernels_ntt_fast.hip).  Why assembly blocks (tools/gen_ntt16_bfly.py writes them): hipcc narrows multiplier
// ops whose high half is dead to v_mul_lo_u32 (half the rate), lowers conditional subtractions to compare + select chains
// through VCC (a VALU write of VCC or an SGPR needs two wait states before a VALU may read it on gfx950) and pads every short
// asm statement with s_nop.  19-20 instructions per butterfly in both directions, 9 of them on the multiplier, 11 fixed scratch
// registers at the top of the 128-register budget.  Conditional subtraction x in [0, 2m) -> [0, m) as t = x - m followed by a
// select on the sign of t (v_bfi_b32 under an arithmetic-shift mask); the two 64-bit differences (u + 4q - v forward,
// a + 4q - b inverse) as v_sub_co_u32 / v_subb_co_u32 with two independent instructions between the halves (the wait
// states a VALU read of VCC needs after a VALU write); every other carry-out is discarded into VCC.
#define NTT16_S(x) "s"(x)
#define NTT16_V(x) "v"(x)
#include "ntt16_bfly.inc"

// forward (Cooley-Tukey): a, b in [0, 8q) -> a' = u + v, b' = u - v + 4q with u = a mod+ 4q in [0, 4q), v = b w in [0, 4q)
// inverse (Gentleman-Sande): a, b in [0, 4q) -> a' = (a + b) mod+ 4q, b' = (a - b + 4q) w, both in [0, 4q)
// SC: the twiddles are wave-uniform (SGPR operands).  H2: the block takes 2 sh as an operand (t.sh2) instead of doubling the
// high word of the multiplicand itself -- one instruction fewer; worth it where a twiddle serves several butterflies (every
// wave-uniform one: the doubling is a scalar instruction; per-lane ones of the stages with fewer twiddles than butterflies).
#ifndef NTT16_INV_H2
#define NTT16_INV_H2 1
#endif
template <bool INV, bool SC, bool H2 = SC && (!INV || NTT16_INV_H2)>
__device__ __forceinline__ void bfly(u64 &x0, u64 &y0, const Tw &t0, const ModC &m)
{
    const u64 a0 = x0, b0 = y0;
    if (INV) {
        u32 aol0, aoh0;
        u64 bo0;
        if (SC && H2)
            NTT16_GS1H(NTT16_S);
        else if (SC)
            NTT16_GS1(NTT16_S);
        else if (H2)
            NTT16_GS1H(NTT16_V);
        else
            NTT16_GS1(NTT16_V);
        x0 = ((u64)aoh0 << 32) | aol0, y0 = bo0;
    } else {
        u64 ao0;
        u32 bol0, boh0;
        if (SC && H2)
            NTT16_CT1H(NTT16_S);
        else if (SC)
            NTT16_CT1(NTT16_S);
        else if (H2)
            NTT16_CT1H(NTT16_V);
        else
            NTT16_CT1(NTT16_V);
        x0 = ao0, y0 = ((u64)boh0 << 32) | bol0;
    }
}
// two butterflies (call sites pair them; one block each: see tools/gen_ntt16_bfly.py on why they are not interleaved)
template <bool INV, bool SC, bool H2 = SC && !INV>
__device__ __forceinline__ void bfly2(u64 &x0, u64 &y0, const Tw &t0, u64 &x1, u64 &y1, const Tw &t1, const ModC &m)
{
    bfly<INV, SC, H2>(x0, y0, t0, m);
    bfly<INV, SC, H2>(x1, y1, t1, m);
}
// m-th index in [0, 16) whose bit `d` (a power of two) is clear
__device__ __forceinline__ constexpr int bfly_lo(int m, int d) { return ((m & ~(d - 1)) << 1) | (m & (d - 1)); }

// Wave-uniform tables (twiddles of passes 1 and 2, modulus constants) are read through the constant address space: a
// plain global pointer gives vector loads of a uniform address (the compiler cannot prove that the kernel's own stores leave
// the tables alone) -- 22 global_load_dwordx4 per slice queued in order behind the slice's HBM loads, four VGPRs per
// twiddle, VGPR operands in the butterfly blocks.  Through address space 4 they are s_load_dwordx4 into SGPRs.
typedef const __attribute__((address_space(4))) u64x2 *TwS;
typedef const __attribute__((address_space(4))) DevConsts *DcS;
__device__ __forceinline__ u64 uniform_addr(const void *p)  // (uniform integer divisions leave their results in VGPRs)
{
    const u64 v = (u64)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// x in [0, 2m) -> [0, m) with the NEGATED modulus (2^64 - m), wave-uniform: t = x - m as one v_lshl_add_u64, then a select on
// the sign of t (v_ashrrev_i32 + two v_bfi_b32): 4 instructions.  hipcc's `x >= m ? x - m : x` is compare + subtract pair +
// two v_cndmask_b32 through VCC with its wait states, and `x + negm` in C++ becomes v_subrev_co / s_nop 1 / v_subb_co with the
// high word of the constant moved into a vector register first -- the whole step inside one statement avoids both.  (The
// temporaries are three of the butterfly blocks' fixed registers.)
__device__ __forceinline__ u64 csub_neg(u64 x, u64 negm)
{
    u32 lo, hi;
    asm("v_lshl_add_u64 v[126:127], %[x], 0, %[nm]\n\t"
        "v_ashrrev_i32 v125, 31, v127\n\t"
        "v_bfi_b32 %[lo], v125, %[xl], v126\n\t"
        "v_bfi_b32 %[hi], v125, %[xh], v127"
        : [lo] "=&v"(lo), [hi] "=&v"(hi)
        : [x] "v"(x), [xl] "v"((u32)x), [xh] "v"((u32)(x >> 32)), [nm] "s"(negm)
        : "v125", "v126", "v127");
    return ((u64)hi << 32) | lo;
}

// Global addresses are "uniform base + 32-bit lane offset in BYTES" (the global_load / global_store form with a scalar base
// register pair and one vector offset register).  Written with element offsets the scaling happens after the zero-extension, the
// compiler can no longer prove that the offset fits 32 bits, and every access pattern keeps a 64-bit per-lane offset pair alive
// across the item loop (eight registers instead of four, and 64-bit vector adds per access).
__device__ __forceinline__ const u64 *at_bytes(const u64 *ubase, u32 boff)
{
    return reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ubase) + boff);
}
__device__ __forceinline__ u64 *at_bytes(u64 *ubase, u32 boff) { return reinterpret_cast<u64 *>(reinterpret_cast<char *>(ubase) + boff); }
__device__ __forceinline__ const u64x2 *at_bytes(const u64x2 *ubase, u32 boff)
{
    return reinterpret_cast<const u64x2 *>(reinterpret_cast<const char *>(ubase) + boff);
}

// the CPT coefficients a thread holds of row r (pass 1): x[CPT r .. CPT r + CPT)
template <u32 CPT>
__device__ __forceinline__ void row_get(u64 *x, int r, const u64 *p)
{
    if (CPT == 2) {
        const u64x2 v = *reinterpret_cast<const u64x2 *>(p);
        x[2 * r] = v.x, x[2 * r + 1] = v.y;
    } else {
        x[r] = *p;
    }
}
template <u32 CPT>
__device__ __forceinline__ void row_put(const u64 *x, int r, u64 *p)
{
    if (CPT == 2) {
        u64x2 v;
        v.x = x[2 * r], v.y = x[2 * r + 1];
        *reinterpret_cast<u64x2 *>(p) = v;
    } else {
        *p = x[r];
    }
}
// Global-memory accesses of the slice itself: through pointers in the GLOBAL address space.  The item loop makes the slice base
// opaque (a scalar pair, see the kernel), and with that the compiler no longer knows which address space a derived pointer is in:
// it would emit FLAT loads and stores, which also count in lgkmcnt -- every wait for an LDS read or a scalar twiddle load behind
// one then waits for HBM.
// (Non-temporal loads / stores of the slices, per launch site, were measured in r04: the transforms get 3-6 % shorter and the
// kernels that consume their output -- which then find nothing in the memory-side cache -- longer by the same microseconds.)
typedef __attribute__((address_space(1))) u64 g_u64;
typedef __attribute__((address_space(1))) u64x2 g_u64x2;
__device__ __forceinline__ u64x2 pair_get_global(const u64 *p) { return *(const g_u64x2 *)p; }
__device__ __forceinline__ void pair_put_global(u64x2 v, u64 *p) { *(g_u64x2 *)p = v; }
template <u32 CPT>
__device__ __forceinline__ void row_put_global(const u64 *x, int r, u64 *p)
{
    if (CPT == 2) {
        u64x2 v;
        v.x = x[2 * r], v.y = x[2 * r + 1];
        pair_put_global(v, p);
    } else {
        *(g_u64 *)p = x[r];
    }
}
template <u32 CPT>
__device__ __forceinline__ void row_get_global(u64 *x, int r, const u64 *p)
{
    if (CPT == 2) {
        const u64x2 v = pair_get_global(p);
        x[2 * r] = v.x, x[2 * r + 1] = v.y;
    } else {
        x[r] = *(const g_u64 *)p;
    }
}

// DS operations of one wave execute in issue order: a hand-off inside the wave only needs the compiler kept from
// moving the reads above the writes
__device__ __forceinline__ void wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct Args {
    u64 *data;            // slices [nitems][NS] (item -> slice through the enumeration below)
    const u64x2 *twp;     // natural-order {w, w_shoup63} pairs: per modulus [fwd N][inv N]
    const u64x2 *twk;     // kernel-ordered pairs of passes 3, 4: per modulus [fwd][inv][slices per limb][TWK_PER_SLICE]
    const DevConsts *dc;
    u32 N;                // ring dimension (pairs per table)
    u32 s0;               // log2 slices per limb
    u32 nitems;           // slices to transform
    u32 mod_base, mod_count;  // limb i uses modulus mod_base + i % mod_count
    u32 flags;
    // forward: the limbs are a compact enumeration of [nb][4][skip_M] without limbs < skip_L of slots 0, 1 (skip_L = 0: off)
    u32 skip_L, skip_M;
    // inverse, standard order in: the lane-ordered EVALUATION input of operand-0 polynomials is also written to
    // copy_out[bin][4][copy_M][N], slots 0, 1, limbs < copy_L (input limbs are [nb][copy_K][2][copy_L])
    u64 *copy_out;
    u32 copy_K, copy_L, copy_M;
    // forward: items [lift_first, nitems) are BV key-switch digits: limb (bin, i, j) of data2[nb][lift_L][lift_L][N]
    // is the centred lift into q_j of residue limb i of the COEFFICIENT polynomial lift_src + bin * lift_stride, transformed;
    // the lift (and, with two folded slices per limb, the outermost stage) happens in the load phase.  lift_first >= nitems: off
    u64 *data2;
    const u64 *lift_src;
    size_t lift_stride;
    u32 lift_first, lift_L;
};
enum : u32 {
    F_STD_IN = 1,    // inverse: EVALUATION input in standard (bit-reversed) order instead of lane order
    F_LAZY_OUT = 2,  // forward: leave [0, 8q) residues (the consumer reduces anyway)
    F_FOLDED = 4,    // inverse: the consumer applies the outermost stage and N^-1: hand over [0, 4q) residues as they are
    F_X_LANE_IN = 8, // inverse, standard order in: operand-0 polynomials come lane-ordered from copy_out's place instead (no copy)
};

// Host side: the twiddle pairs of passes 3 and 4 of one (modulus, direction) in kernel order, for every slice of a limb.
// nat: the N natural-order pairs {w, w_shoup63} (index = 2^stage + group).  out: [1 << s0][TWK_PER_SLICE] pairs.
template <u32 LOGNS>
inline void build_twk_table_t(const u64 *nat, u32 s0, std::vector<u64> &out)
{
    typedef Geo<LOGNS> G;
    out.assign((size_t)(1u << s0) * G::TWK_PER_SLICE * 2, 0);
    for (u32 blk = 0; blk < (1u << s0); blk++) {
        u64 *o = &out[(size_t)blk * G::TWK_PER_SLICE * 2];
        auto put = [&](size_t dst, size_t src) {
            o[2 * dst] = nat[2 * src];
            o[2 * dst + 1] = nat[2 * src + 1];
        };
        for (u32 w = 0; w < G::W; w++) {
            for (u32 sc = 0; sc < 4; sc++)  // pass 3: the four stages below pass 2, 16 W << sc local groups
                for (u32 jj = 0; jj < (1u << sc); jj++)
                    for (u32 la = 0; la < 16; la++) {
                        const u32 ml = (16 * G::W) << sc;
                        put(((size_t)w * 15 + ((1u << sc) - 1 + jj)) * 16 + la, ((size_t)ml << s0) + (size_t)blk * ml + (((16 * w + la) << sc) + jj));
                    }
            for (u32 l = 0; l < 64; l++) {  // pass 4: the last two stages (slots 0..3 and 4..11)
                const u32 m11 = 256 * G::W, m12 = 512 * G::W;
                for (u32 jj = 0; jj < 4; jj++)
                    put(G::W * 15 * 16 + ((size_t)w * 12 + jj) * 64 + l, ((size_t)m11 << s0) + (size_t)blk * m11 + 256 * w + 4 * l + jj);
                for (u32 jj = 0; jj < 8; jj++)
                    put(G::W * 15 * 16 + ((size_t)w * 12 + 4 + jj) * 64 + l, ((size_t)m12 << s0) + (size_t)blk * m12 + 512 * w + 8 * l + jj);
            }
        }
    }
}
inline void build_twk_table(const u64 *nat, u32 s0, std::vector<u64> &out) { build_twk_table_t<13>(nat, s0, out); }
// lane-order position p of a slice of T * 16 coefficients -> standard (bit-reversed) position
inline u32 lane_to_std_t(u32 p, u32 threads)
{
    const u32 tau = (p >> 1) % threads, j = (p >> 1) / threads;
    return 16 * tau + 2 * j + (p & 1);
}
inline u32 lane_to_std(u32 p) { return lane_to_std_t(p, T); }

// in-kernel cycle stamps are a tooling build (tools/ntt_lab.hip defines NTT16_STAMP before including this file)
#ifndef NTT16_STAMP
#define NTT16_STAMP(i)
#endif

// ---- the four register passes (x[16] in the layout the pass names) ---------------------------------------------------------
// group index of a stage with ml local groups: global table index (ml << s0) + blk * ml + i
#define NTT16_TWL(ml, i) tw[(((u32)(ml)) << a.s0) + blk * (u32)(ml) + (i)]
// LDS addresses are written as (one per-thread base) + (compile-time constant): phi is additive over multiples of 32, and a
// base the compiler has to keep per row ends up in scratch memory.  Global addresses: uniform row base + one lane offset.
#define NTT16_FENCE() __builtin_amdgcn_sched_barrier(0)
// Wave priority falls as a wave advances from one workgroup barrier to the next.  The SIMD arbiter serves the highest
// priority, then the oldest wave: with equal priorities the older of the waves a workgroup has on each SIMD runs at full speed,
// the younger on what is left, and the workgroup's barriers then wait for the younger ones (measured: 9000 of 32000 cycles per
// slice).  With the priority tied to progress the wave that is behind is served first and all arrive together.  Forward: the
// barrier sits after pass 1, so pass 2 is the start of the cycle and pass 1 of the NEXT slice its end (lowest priority; its
// loads are issued at the highest, they are a handful of instructions): 2048 slices 88.2 -> 82.5 us against "pass 1 highest",
// the run's ragged launches unchanged (profiles/r03/ntt16_lab_wave_priorities.txt).  Inverse (r04): both barriers sit between
// pass 2' and pass 1' (around the row reads), so pass 1' opens the cycle and pass 2' closes it at the lowest priority
// (profiles/r04/ntt16_lab_wave_priorities_inverse.txt).
#define NTT16_PASS_PRIO(p) __builtin_amdgcn_s_setprio(p)
#ifndef NTT16_PF
#define NTT16_PF 3, 0, 3, 2, 1   // forward: loads, pass 1, pass 2, pass 3, pass 4
#endif
#ifndef NTT16_PI
#define NTT16_PI 3, 3, 2, 0, 1   // inverse: loads, pass 4', pass 3', pass 2', pass 1'
#endif
#define NTT16_PRIO_PICK_(i, a0, a1, a2, a3, a4) ((i) == 0 ? (a0) : (i) == 1 ? (a1) : (i) == 2 ? (a2) : (i) == 3 ? (a3) : (a4))
#define NTT16_PRIO_PICK(i, ...) NTT16_PRIO_PICK_(i, __VA_ARGS__)
#define NTT16_PRIO_F(i) NTT16_PASS_PRIO(NTT16_PRIO_PICK(i, NTT16_PF))
#define NTT16_PRIO_I(i) NTT16_PASS_PRIO(NTT16_PRIO_PICK(i, NTT16_PI))

#ifndef NTT16_LIFT_XCD
#define NTT16_LIFT_XCD 1
#endif
// LIFT (forward only): the launch may carry key-switch digit items (Args::lift_first); a separate instantiation, so that the
// plain forward transform does not pay for the lift's registers
template <u32 LOGNS, bool INV, bool LIFT = false>
__global__ void __launch_bounds__(Geo<LOGNS>::T, 4) ntt16_kernel_t(Args a)
{
    typedef Geo<LOGNS> G;
    constexpr u32 NS = G::NS, T = G::T, R = G::R, LOGR = G::LOGR, CPT = G::CPT, TWK_PER_SLICE = G::TWK_PER_SLICE;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 tau = threadIdx.x;
    const u32 w = __builtin_amdgcn_readfirstlane(tau >> 6), l = tau & 63;
    const u32 la = l >> 2, lc = l & 3;
    // element -> image position, per layout (phi(e) = e + 2 (e >> 5)):
    //   pass 1: e = 1024 r + 2 tau          -> p1 + 1088 r            p1 = phi(2 tau)
    //   pass 2: e = 1024 w + 64 k + l       -> p2 + 68 k              p2 = phi(1024 w + l)
    //   pass 3: e = 1024 w + 64 a + 4 k + c -> p3 + 4 k + 2 (k >> 3)  p3 = 1088 w + 68 a + c
    //   pass 4: e = 1024 w + 16 l + k       -> p4 + k                 p4 = phi(1024 w + 16 l)
    u64 *const p1 = lds + phi(CPT * tau);
    u64 *const p2 = lds + phi(1024 * w + l);
    u64 *const p3 = lds + 1088 * w + 68 * la + lc;
    u64 *const p4 = lds + phi(1024 * w + 16 * l);

    u64 x[16];
    // item -> (limb, slice of the limb); forward launches may enumerate [nb][4][skip_M] without the Q limbs of slots 0, 1
    auto limb_of = [&](u32 it) -> u32 {
        u32 lb = it >> a.s0;
        if (!INV && a.skip_L) {
            const u32 P = a.skip_M - a.skip_L, per = 2 * P + 2 * a.skip_M;
            const u32 cb = lb / per, r = lb % per;
            lb = cb * 4 * a.skip_M + (r < 2 * P ? (r / P) * a.skip_M + a.skip_L + r % P : 2 * a.skip_M + (r - 2 * P));
        }
        return lb;
    };
    for (u32 item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        // The lane offsets are made opaque once per item: as loop invariants the compiler widens them to 64-bit pairs, adds the
        // table bases it can hoist, keeps all of that alive across the loop -- and, at this kernel's register budget, spills it.
        // Only the thread index itself is carried from item to item; the four offsets are a handful of instructions per slice.
        u32 tau_ = tau;
        asm volatile("" : "+v"(tau_));
        const u32 voffb = 16 * tau_;       // lane offset (in bytes) of the lane-ordered pairs 2 (T j + tau)
        const u32 coffb = 8 * CPT * tau_;  // ... and of the row accesses 1024 r + CPT tau (pass 1)
        const u32 lab = 4 * (tau_ & 60), lb = 16 * (tau_ & 63);  // ... and into the kernel-ordered twiddle tables of passes 3 (16 la) and 4 (16 l)
        const bool lift = LIFT && !INV && item >= a.lift_first;
        // The G = lift_L * 2^s0 digit items of one (ciphertext, source limb) read the same source limb.  Consecutive workgroups sit
        // on consecutive XCDs, each with its own L2: dealt in order, every XCD fetches every source limb from HBM.  Transposing each
        // block of 8 G items (8 x G) hands the items that share a source to workgroups 8 apart -- one XCD, one fetch, G - 1 L2 hits.
        u32 item_l = item;
        if (lift) {
            item_l = item - a.lift_first;
#if NTT16_LIFT_XCD
            const u32 G = a.lift_L << a.s0, blk8 = 8 * G, within = item_l % blk8;
            if (item_l - within + blk8 <= a.nitems - a.lift_first) item_l = item_l - within + (within & 7) * G + (within >> 3);
#endif
        }
        const u32 blk = item_l & ((1u << a.s0) - 1);
        const u32 limb = lift ? item_l >> a.s0 : limb_of(item);
        // (uniform; laundered through a scalar register pair so that the slice's addresses stay "scalar base + lane offset": left to
        // itself the compiler hoists a.data + lane offset out of the item loop as a 64-bit vector base, which the inverse kernel
        // -- at its 128-register budget -- spilled to scratch, and a scratch reload is a vector-memory operation: the
        // s_waitcnt vmcnt(0) in front of its use drained the previous slice's stores and this slice's twiddle loads before the
        // first data load was even issued)
        u64 *g = (lift ? a.data2 : a.data) + (((size_t)limb << a.s0) + blk) * NS;
        asm("" : "+s"(g));
        const u32 mod = __builtin_amdgcn_readfirstlane(a.mod_base + limb % a.mod_count);
        const DcS dcs = (DcS)(u64)a.dc;
        const u64 q = dcs->mod[mod].q;
        const u64 q2 = 2 * q, q4 = 4 * q;
        ModC mc;
        mc.nql = (u32)(0 - q), mc.nqh = (u32)((0 - q) >> 32), mc.nq4 = 0 - q4, mc.q4 = q4;
        const TwS tw = (TwS)(u64)(a.twp + ((size_t)mod * 2 + (INV ? 1 : 0)) * a.N);
        const u64x2 *__restrict__ twk = a.twk + (((size_t)mod * 2 + (INV ? 1 : 0)) << a.s0) * TWK_PER_SLICE + (size_t)blk * TWK_PER_SLICE;
        const u64x2 *__restrict__ tw3 = twk + (size_t)w * 15 * 16;                 // [slot 0..14][16 a]
        const u64x2 *__restrict__ tw4 = twk + G::W * 15 * 16 + (size_t)w * 12 * 64;   // [slot 0..11][64 l]
        // per-lane twiddles of passes 3 and 4, loaded one stage ahead of their use (named by stage: 7, 8, 9, 10, 11, 12)
        u64x2 t7[1], t8[2], t9[4], t10[8], t11[4], t12[8];
#define NTT16_LOAD3(dst, sc)                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < (1 << (sc)); j_++) dst[j_] = *at_bytes(tw3 + 16 * ((1 << (sc)) - 1 + j_), lab)
#define NTT16_LOAD3H(dst, sc, from)                                                \
    _Pragma("unroll") for (int j_ = (from); j_ < (from) + 4; j_++) dst[j_] = *at_bytes(tw3 + 16 * ((1 << (sc)) - 1 + j_), lab)
#define NTT16_LOAD4(dst, first, n)                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < (n); j_++) dst[j_] = *at_bytes(tw4 + 64 * ((first) + j_), lb)

        if (!INV) {
            NTT16_STAMP(0);
            NTT16_PRIO_F(0);
            // ---- pass 1: rows r = 0..7 (bits 12..10) of the column pair -------------------------------------------------
            if (lift) {
                // Centred lift of a residue v of q_i into q_j for equal-width primes (q_i < 2 q_j for every pair: the host checks):
                // the centred representative v - [v > q_i / 2] q_i lies in (-q_j, q_j), so its residue mod q_j is v itself for
                // v <= floor(q_i / 2) and v + (q_j - q_i) above -- one masked 64-bit add, canonical result (five instructions; the
                // formulation it replaces reduced v mod q_j first and corrected by q_i mod q_j with a sign fix-up: fourteen).
                const u32 LL = a.lift_L, li = __builtin_amdgcn_readfirstlane((limb / LL) % LL), lj = __builtin_amdgcn_readfirstlane(limb % LL);
                const u64 *src = a.lift_src + (size_t)(limb / (LL * LL)) * a.lift_stride + (size_t)li * a.N;
                const u64 qi = dcs->mod[li].q;
                const u64 nqh1 = 0 - (qi / 2 + 1), dq = q - qi;
                auto lift1 = [&](u64 v) -> u64 {
                    u64 r;
                    asm("v_lshl_add_u64 v[126:127], %[v], 0, %[nqh1]\n\t"   // v - (floor(q_i / 2) + 1): negative iff v <= floor(q_i / 2)
                        "v_ashrrev_i32 v125, 31, v127\n\t"
                        "v_bfi_b32 v126, v125, 0, %[dql]\n\t"               // q_j - q_i where v is above the half, 0 otherwise
                        "v_bfi_b32 v127, v125, 0, %[dqh]\n\t"
                        "v_lshl_add_u64 %[r], %[v], 0, v[126:127]"
                        : [r] "=v"(r)
                        : [v] "v"(v), [nqh1] "s"(nqh1), [dql] "s"((u32)dq), [dqh] "s"((u32)(dq >> 32))
                        : "v125", "v126", "v127");
                    return r;
                };
                if (a.s0 == 1) {
                    // two folded slices per limb: the outermost stage (u, v) -> (u + v psi^{N/2}, u - v psi^{N/2}) is one more
                    // butterfly, and this slice keeps its half (kernels_pie.hip fold_store does the same for the other kernels)
                    u64x2 fw;
                    fw.x = dcs->fold_w[lj];
                    fw.y = dcs->fold_w_sh[lj] >> 1;
                    const Tw t = make_tw(fw);
                    u64 y[16];  // the other half of the limb
#pragma unroll
                    for (int r = 0; r < (int)R; r++) {
                        row_get_global<CPT>(x, r, at_bytes(src + 1024 * r, coffb));
                        row_get_global<CPT>(y, r, at_bytes(src + NS + 1024 * r, coffb));
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        u64 u0 = lift1(x[k]), v0 = lift1(y[k]);
                        bfly<false, true>(u0, v0, t, mc);
                        x[k] = blk ? v0 : u0;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < (int)R; r++) row_get_global<CPT>(x, r, at_bytes(src + 1024 * r, coffb));
#pragma unroll
                    for (int k = 0; k < 16; k++) x[k] = lift1(x[k]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < (int)R; r++) row_get_global<CPT>(x, r, at_bytes(g + 1024 * r, coffb));
            }
            NTT16_PRIO_F(1);
#pragma unroll
            for (int s = 0; s < (int)LOGR; s++) {
                const int d = (int)(R >> 1) >> s;
#pragma unroll
                for (int r = 0; r < (int)R; r++) {
                    if (r & d) continue;
                    const Tw t = make_tw(NTT16_TWL(1u << s, (u32)r >> (LOGR - s)));
#pragma unroll
                    for (int c = 0; c < (int)CPT; c++) bfly<false, true>(x[CPT * r + c], x[CPT * (r + d) + c], t, mc);
                }
            }
            NTT16_STAMP(1);
            __syncthreads();  // every wave has finished the previous slice's LDS reads
            NTT16_STAMP(2);
#pragma unroll
            for (int r = 0; r < (int)R; r++) row_put<CPT>(x, r, p1 + 1088 * r);
            NTT16_STAMP(3);
            __syncthreads();
            NTT16_STAMP(4);
            // ---- pass 2: e = 1024 w + 64 k + l, stages 3..6 -----------------------------------------------------------------
            NTT16_PRIO_F(2);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p2[68 * k];
#pragma unroll
            for (int sb = 0; sb < 4; sb++) {
                const int d = 8 >> sb;
#pragma unroll
                for (int mm = 0; mm < 8; mm += 2) {
                    const int k0 = bfly_lo(mm, d), k1 = bfly_lo(mm + 1, d);
                    const Tw t0 = make_tw(NTT16_TWL(G::W << sb, (w << sb) + ((u32)k0 >> (4 - sb))));
                    const Tw t1 = make_tw(NTT16_TWL(G::W << sb, (w << sb) + ((u32)k1 >> (4 - sb))));
                    bfly2<false, true>(x[k0], x[k0 + d], t0, x[k1], x[k1 + d], t1, mc);
                }
            }
            // per-lane twiddles from here on, each stage's loaded one stage ahead (the last stages' eight in two halves: with the
            // next slice in flight the register budget is x 32 + y 32 + twiddles <= 48)
            NTT16_LOAD3(t7, 0);
            NTT16_LOAD3(t8, 1);
#pragma unroll
            for (int k = 0; k < 16; k++) p2[68 * k] = x[k];
            NTT16_STAMP(5);
            wave_sync();
            // ---- pass 3: e = 1024 w + 64 a + 4 k + c, stages 7..10 -------------------------------------------------------
            NTT16_PRIO_F(3);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p3[4 * k + 2 * (k >> 3)];
            NTT16_LOAD3(t9, 2);
            NTT16_FENCE();
            {
                const Tw t = make_tw(t7[0]);
#pragma unroll
                for (int k = 0; k < 8; k += 2) bfly2<false, false, true>(x[k], x[k + 8], t, x[k + 1], x[k + 9], t, mc);
            }
            NTT16_LOAD3H(t10, 3, 0);
            NTT16_FENCE();
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const int k0 = bfly_lo(mm, 4), k1 = bfly_lo(mm + 1, 4);
                bfly2<false, false, true>(x[k0], x[k0 + 4], make_tw(t8[k0 >> 3]), x[k1], x[k1 + 4], make_tw(t8[k1 >> 3]), mc);
            }
            NTT16_LOAD3H(t10, 3, 4);
            NTT16_FENCE();
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const int k0 = bfly_lo(mm, 2), k1 = bfly_lo(mm + 1, 2);
                bfly2<false, false>(x[k0], x[k0 + 2], make_tw(t9[k0 >> 2]), x[k1], x[k1 + 2], make_tw(t9[k1 >> 2]), mc);
            }
            NTT16_LOAD4(t11, 0, 4);
            NTT16_FENCE();
#pragma unroll
            for (int k = 0; k < 16; k += 4)
                bfly2<false, false>(x[k], x[k + 1], make_tw(t10[k >> 1]), x[k + 2], x[k + 3], make_tw(t10[(k >> 1) + 1]), mc);
#pragma unroll
            for (int k = 0; k < 16; k++) p3[4 * k + 2 * (k >> 3)] = x[k];
            NTT16_STAMP(6);
            wave_sync();
            // ---- pass 4: e = 1024 w + 16 l + k, stages 11, 12 ---------------------------------------------------------------
            NTT16_PRIO_F(4);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u64x2 v = *reinterpret_cast<const u64x2 *>(p4 + 2 * j);
                x[2 * j] = v.x;
                x[2 * j + 1] = v.y;
            }
            NTT16_LOAD4(t12, 4, 4);
            NTT16_FENCE();
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {  // stage 11: both butterflies of a group of four share its twiddle
                const Tw t = make_tw(t11[g4]);
                bfly2<false, false>(x[4 * g4], x[4 * g4 + 2], t, x[4 * g4 + 1], x[4 * g4 + 3], t, mc);
                if (g4 == 1) {
                    NTT16_FENCE();
                    _Pragma("unroll") for (int j_ = 4; j_ < 8; j_++) t12[j_] = *at_bytes(tw4 + 64 * (4 + j_), lb);
                    NTT16_FENCE();
                }
            }
            NTT16_FENCE();
#pragma unroll
            for (int k = 0; k < 16; k += 4)  // stage 12
                bfly2<false, false>(x[k], x[k + 1], make_tw(t12[k >> 1]), x[k + 2], x[k + 3], make_tw(t12[(k >> 1) + 1]), mc);
            NTT16_STAMP(7);
            // ---- lane-ordered store -------------------------------------------------------------------------------------------
            const bool lazy = (a.flags & F_LAZY_OUT) != 0 && !lift;  // the key-switch MAC splits canonical digits into 30-bit halves
            if (!lazy) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = csub_neg(csub_neg(csub_neg(x[k], 0 - q4), 0 - q2), 0 - q);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                u64x2 v;
                v.x = x[2 * j], v.y = x[2 * j + 1];
                pair_put_global(v, at_bytes(g + 2 * T * j, voffb));
            }
            NTT16_STAMP(8);
        } else {
            // ---- input: 16 contiguous coefficients per thread -------------------------------------------------------------
            NTT16_STAMP(0);
            NTT16_PRIO_I(0);
            NTT16_LOAD4(t12, 4, 4);   // (the other four behind the data loads: the register budget is x 32 + twiddles)
            // operand-0 polynomials of a standard-order launch: their lane-ordered EVALUATION form lives in the Q limbs of the QP
            // operand array -- written there by this launch (copy) or already by stage A (F_X_LANE_IN: read from there)
            const bool is_x = (a.flags & F_STD_IN) && a.copy_out && (limb / (2 * a.copy_L)) % a.copy_K == 0;
            u64 *co = nullptr;
            if (is_x) {
                const u32 bin = limb / (2 * a.copy_L * a.copy_K), cc = (limb / a.copy_L) & 1, i = limb % a.copy_L;
                co = a.copy_out + (((((size_t)bin * 4 + cc) * a.copy_M + i) << a.s0) + blk) * NS;
                asm("" : "+s"(co));
            }
            const bool x_in = is_x && (a.flags & F_X_LANE_IN);
            if ((a.flags & F_STD_IN) && !x_in) {
                // Standard order in.  A wave's pass-4' elements are the 1024 contiguous coefficients of its own block: it loads
                // exactly those (coalesced: pair 64 j + l of the block per lane and instruction), drops them into its own region of
                // the image and picks up its 16 contiguous coefficients -- a hand-off inside the wave.  (Until r04 the slice
                // came in as the rows of pass 1 -- columns across all waves -- which took a second workgroup barrier.)
                u64 yv[16];
                const u64 *gw = g + 1024 * w;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const u64x2 v = pair_get_global(at_bytes(gw + 128 * j, lb));
                    yv[2 * j] = v.x, yv[2 * j + 1] = v.y;
                }
                // (the image is free: every wave passed the barrier behind the previous slice's pass-1' reads)
                {
                    // element 1024 w + 128 j + 2 l at phi(.): phi is additive over multiples of 32, 128 j -> 136 j
                    u64 *const pw = lds + phi(1024 * w + 2 * l);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        u64x2 v;
                        v.x = yv[2 * j], v.y = yv[2 * j + 1];
                        *reinterpret_cast<u64x2 *>(pw + 136 * j) = v;
                    }
                }
                wave_sync();
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const u64x2 v = *reinterpret_cast<const u64x2 *>(p4 + 2 * j);
                    x[2 * j] = v.x;
                    x[2 * j + 1] = v.y;
                }
                if (is_x) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        u64x2 v;
                        v.x = x[2 * j], v.y = x[2 * j + 1];
                        pair_put_global(v, at_bytes(co + 2 * T * j, voffb));
                    }
                }
            } else {
                const u64 *src = x_in ? co : g;
                asm("" : "+s"(src));
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const u64x2 v = pair_get_global(at_bytes(src + 2 * T * j, voffb));
                    x[2 * j] = v.x, x[2 * j + 1] = v.y;
                }
            }
            NTT16_FENCE();
            _Pragma("unroll") for (int j_ = 4; j_ < 8; j_++) t12[j_] = *at_bytes(tw4 + 64 * (4 + j_), lb);
            NTT16_FENCE();
            NTT16_PRIO_I(1);
            // ---- pass 4': stages 12, 11 ----------------------------------------------------------------------------------------
#pragma unroll
            for (int k = 0; k < 16; k += 4) {  // stage 12
                bfly2<true, false>(x[k], x[k + 1], make_tw(t12[k >> 1]), x[k + 2], x[k + 3], make_tw(t12[(k >> 1) + 1]), mc);
                if (k == 4) {   // the first half of this stage's twiddles is dead: the next stage's take their registers
                    NTT16_FENCE();
                    NTT16_LOAD4(t11, 0, 4);
                    NTT16_FENCE();
                }
            }
            NTT16_LOAD3(t10, 3);
            NTT16_FENCE();
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {  // stage 11
                const Tw t = make_tw(t11[g4]);
                bfly2<true, false>(x[4 * g4], x[4 * g4 + 2], t, x[4 * g4 + 1], x[4 * g4 + 3], t, mc);
            }
            NTT16_STAMP(1);
            // (the stores below overwrite the image the other waves read in pass 1' of the previous slice: they are behind the
            // barrier that follows those reads)
            NTT16_STAMP(2);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                u64x2 v;
                v.x = x[2 * j];
                v.y = x[2 * j + 1];
                *reinterpret_cast<u64x2 *>(p4 + 2 * j) = v;
            }
            wave_sync();
            NTT16_STAMP(3);
            // ---- pass 3': stages 10..7 -----------------------------------------------------------------------------------------
            NTT16_PRIO_I(2);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p3[4 * k + 2 * (k >> 3)];
            NTT16_LOAD3(t9, 2);
            NTT16_FENCE();
#pragma unroll
            for (int k = 0; k < 16; k += 4)
                bfly2<true, false>(x[k], x[k + 1], make_tw(t10[k >> 1]), x[k + 2], x[k + 3], make_tw(t10[(k >> 1) + 1]), mc);
            NTT16_LOAD3(t8, 1);
            NTT16_LOAD3(t7, 0);
            NTT16_FENCE();
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const int k0 = bfly_lo(mm, 2), k1 = bfly_lo(mm + 1, 2);
                bfly2<true, false>(x[k0], x[k0 + 2], make_tw(t9[k0 >> 2]), x[k1], x[k1 + 2], make_tw(t9[k1 >> 2]), mc);
            }
            NTT16_FENCE();
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const int k0 = bfly_lo(mm, 4), k1 = bfly_lo(mm + 1, 4);
                bfly2<true, false>(x[k0], x[k0 + 4], make_tw(t8[k0 >> 3]), x[k1], x[k1 + 4], make_tw(t8[k1 >> 3]), mc);
            }
            {
                const Tw t = make_tw(t7[0]);
#pragma unroll
                for (int k = 0; k < 8; k += 2) bfly2<true, false>(x[k], x[k + 8], t, x[k + 1], x[k + 9], t, mc);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) p3[4 * k + 2 * (k >> 3)] = x[k];
            wave_sync();
            NTT16_STAMP(4);
            // ---- pass 2': stages 6..3 ------------------------------------------------------------------------------------------
            NTT16_PRIO_I(3);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p2[68 * k];
#pragma unroll
            for (int sb = 3; sb >= 0; sb--) {
                const int d = 8 >> sb;
#pragma unroll
                for (int mm = 0; mm < 8; mm += 2) {
                    const int k0 = bfly_lo(mm, d), k1 = bfly_lo(mm + 1, d);
                    const Tw t0 = make_tw(NTT16_TWL(G::W << sb, (w << sb) + ((u32)k0 >> (4 - sb))));
                    const Tw t1 = make_tw(NTT16_TWL(G::W << sb, (w << sb) + ((u32)k1 >> (4 - sb))));
                    bfly2<true, true>(x[k0], x[k0 + d], t0, x[k1], x[k1 + d], t1, mc);
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) p2[68 * k] = x[k];
            NTT16_STAMP(5);
            __syncthreads();
            NTT16_STAMP(6);
            // ---- pass 1': the top LOGR stages, stored straight to HBM ------------------------------------------------------------------
            NTT16_PRIO_I(4);
#pragma unroll
            for (int r = 0; r < (int)R; r++) row_get<CPT>(x, r, p1 + 1088 * r);
            // The image is free from here on.  This is where the workgroup waits for "everybody has read it", not in front of the
            // next slice's first stores: the waves come out of the barrier above together and read eight rows each, so nobody
            // waits here -- behind pass 4' of the next slice the older half of the waves stood 9 000 of a slice's 42 000 cycles
            // waiting for the younger half (equal priorities: the arbiter serves the oldest wave first;
            // profiles/r04/ntt16_lab_cycle_stamps_inverse.txt).
            __syncthreads();
#pragma unroll
            for (int s = (int)LOGR - 1; s >= 0; s--) {
                const int d = (int)(R >> 1) >> s;
#pragma unroll
                for (int r = 0; r < (int)R; r++) {
                    if (r & d) continue;
                    const Tw t = make_tw(NTT16_TWL(1u << s, (u32)r >> (LOGR - s)));
#pragma unroll
                    for (int c = 0; c < (int)CPT; c++) bfly<true, true>(x[CPT * r + c], x[CPT * (r + d) + c], t, mc);
                }
            }
            NTT16_STAMP(7);
            const u64 n_inv = dcs->mod[mod].n_inv, n_inv_sh = dcs->mod[mod].n_inv_sh;
            if (!(a.flags & F_FOLDED)) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (a.s0 == 0)
                        x[k] = csub_neg(mul_shoup_lazy(x[k], n_inv, n_inv_sh, q), 0 - q);
                    else  // split transform: the global-memory stages expect canonical residues
                        x[k] = csub_neg(csub_neg(x[k], 0 - q2), 0 - q);
                }
            }
#pragma unroll
            for (int r = 0; r < (int)R; r++) row_put_global<CPT>(x, r, at_bytes(g + 1024 * r, coffb));
            NTT16_STAMP(8);
        }
    }
}
#undef NTT16_LOAD3
#undef NTT16_LOAD3H
#undef NTT16_LOAD4
#undef NTT16_FENCE
#undef NTT16_TWL

}  // namespace ntt16
}  // namespace piehip
