// madasm.h -- v_mad_u64_u32 based multiply-accumulate helpers (device only, gfx950).
#pragma once
#include "modarith.h"

namespace piehip {

// 32 x 32 + 64 -> 64.  Plain C++: hipcc selects v_mad_u64_u32 whenever all 64 result bits are live (they are in every
// accumulator below), schedules it with exact hazard knowledge and takes uniform operands from SGPRs.  (An earlier version
// issued the instruction through inline asm: every asm statement costs an s_nop and early-clobber copies on gfx950, and the
// premise -- v_mul_lo_u32 at half the rate of the mad -- does not hold: tools/microbench_ops.hip measures 4.1 vs 4.2 cycles.)
__device__ __forceinline__ u64 mad_u(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ u64 mul_u(u32 a, u32 b) { return (u64)a * b; }

// Carry-free accumulation of products of residues < 2^60: operands are split into 30-bit halves and the
// three product columns (2^0, 2^30, 2^60) are summed in separate 64-bit words.  Every partial product is
// < 2^60, so up to 16 terms fit without overflow (column 1 takes two products per term: up to 8 terms
// ... see COLACC_MAX_TERMS).  4 mads per term, no carries, no wait states.
struct Split30 {
    u32 lo, hi;
};
__device__ __forceinline__ Split30 split30(u64 x)
{
    Split30 s;
    s.lo = (u32)x & 0x3FFFFFFFu;
    s.hi = (u32)(x >> 30);
    return s;
}
struct ColAcc {
    u64 c0, c1, c2;
};
static const u32 COLACC_MAX_TERMS = 8;  // column 1 receives 2 products < 2^60 per term: 16 * 2^60 = 2^64
__device__ __forceinline__ void colacc_mac(ColAcc &a, Split30 x, Split30 y)
{
    a.c0 = mad_u(x.lo, y.lo, a.c0);
    a.c1 = mad_u(x.lo, y.hi, a.c1);
    a.c1 = mad_u(x.hi, y.lo, a.c1);
    a.c2 = mad_u(x.hi, y.hi, a.c2);
}
// Two accumulators (the two components of an index ciphertext) take the same database word d < 2^60: split d and issue the
// eight multiply-adds as ONE instruction block.  Left to the compiler, the sums are reassociated into v_mad_u64_u32 ..., 0
// plus a 64-bit add per column (98 instructions for 7 words instead of 56) and the halves of d live in extra registers.
__device__ __forceinline__ void colacc_mac2(ColAcc &a, ColAcc &b, Split30 x0, Split30 x1, u64 d)
{
    u32 dl, dh;
    asm("v_and_b32 %[dl], 0x3fffffff, %[d0]\n\t"
        "v_alignbit_b32 %[dh], %[d1], %[d0], 30\n\t"
        "v_mad_u64_u32 %[a0], vcc, %[x0l], %[dl], %[a0]\n\t"
        "v_mad_u64_u32 %[a1], vcc, %[x0l], %[dh], %[a1]\n\t"
        "v_mad_u64_u32 %[b0], vcc, %[x1l], %[dl], %[b0]\n\t"
        "v_mad_u64_u32 %[b1], vcc, %[x1l], %[dh], %[b1]\n\t"
        "v_mad_u64_u32 %[a2], vcc, %[x0h], %[dh], %[a2]\n\t"
        "v_mad_u64_u32 %[b2], vcc, %[x1h], %[dh], %[b2]\n\t"
        "v_mad_u64_u32 %[a1], vcc, %[x0h], %[dl], %[a1]\n\t"
        "v_mad_u64_u32 %[b1], vcc, %[x1h], %[dl], %[b1]\n\t"
        : [a0] "+v"(a.c0), [a1] "+v"(a.c1), [a2] "+v"(a.c2), [b0] "+v"(b.c0), [b1] "+v"(b.c1), [b2] "+v"(b.c2), [dl] "=&v"(dl),
          [dh] "=&v"(dh)
        : [d0] "v"((u32)d), [d1] "v"((u32)(d >> 32)), [x0l] "v"(x0.lo), [x0h] "v"(x0.hi), [x1l] "v"(x1.lo), [x1h] "v"(x1.hi)
        : "vcc");
}
// push the overflow of the two low columns upwards (value unchanged): after it c0, c1 < 2^30, so another
// COLACC_MAX_TERMS terms fit; the top column then holds up to 15 terms' worth before it overflows
static const u32 COLACC_MAX_TOTAL = 15;
__device__ __forceinline__ void colacc_carry(ColAcc &a)
{
    a.c1 += a.c0 >> 30;
    a.c0 &= 0x3FFFFFFFull;
    a.c2 += a.c1 >> 30;
    a.c1 &= 0x3FFFFFFFull;
}
// c0 + c1 2^30 + c2 2^60 as a 128-bit integer
__device__ __forceinline__ U128 colacc_value(const ColAcc &a)
{
    U128 r = {a.c0, 0};
    add128(r, U128{a.c1 << 30, a.c1 >> 34});
    add128(r, U128{a.c2 << 60, a.c2 >> 4});
    return r;
}

}  // namespace piehip
