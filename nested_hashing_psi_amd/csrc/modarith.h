// modarith.h -- 64-bit modular arithmetic for RNS primes < 2^61, shared by host (g++/hipcc
// host pass) and device (gfx950) code so both evaluate the same integer formulas.
//
// Replaces the arithmetic OpenFHE performs under the reference's calls at
// src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.cpp:108,112,113,116,123,126
// (NativeInteger ModMul / ModAdd; OpenFHE itself is not part of the reference tree).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PH_HD __host__ __device__ __forceinline__
#else
#define PH_HD inline
#endif

namespace piehip {

typedef uint64_t u64;
typedef uint32_t u32;

// Per-modulus constants.  r1:r0 = floor(2^128 / q) (two-word Barrett); fconst/fshift give the
// 60-bit fixed-point fraction used by the HPS rounding terms (see fixfrac).
struct Mod {
    u64 q;
    u64 r0, r1;
    u64 n_inv, n_inv_sh;  // N^-1 mod q and its Shoup companion (inverse NTT scaling)
    u64 fconst;           // floor(2^(127-fshift) / q)
    u32 fshift;           // clz(q)
    u32 pad;
};

PH_HD u64 mulhi(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

struct U128 {
    u64 lo, hi;
};

PH_HD U128 mul128(u64 a, u64 b)
{
    U128 r;
    r.lo = a * b;
    r.hi = mulhi(a, b);
    return r;
}
PH_HD void add128(U128 &acc, U128 x)
{
    u64 lo = acc.lo + x.lo;
    acc.hi += x.hi + (lo < acc.lo ? 1 : 0);
    acc.lo = lo;
}
PH_HD void mac128(U128 &acc, u64 a, u64 b) { add128(acc, mul128(a, b)); }

// z = z1:z0 < 2^128  ->  z mod q.  The quotient estimate is exact to within 2, computed mod 2^64
// (wrap-around is harmless: only z - qhat*q mod 2^64 is used and the true remainder is < 3q < 2^64).
PH_HD u64 barrett128(u64 z1, u64 z0, const Mod &m)
{
    u64 c = mulhi(z0, m.r0);
    U128 t2 = mul128(z0, m.r1);
    U128 t3 = mul128(z1, m.r0);
    u64 mid = t2.lo + t3.lo;
    u64 carry = (mid < t2.lo ? 1 : 0);
    u64 mid2 = mid + c;
    carry += (mid2 < mid ? 1 : 0);
    u64 qhat = z1 * m.r1 + t2.hi + t3.hi + carry;
    u64 r = z0 - qhat * m.q;
    r = r >= 2 * m.q ? r - 2 * m.q : r;
    r = r >= m.q ? r - m.q : r;
    return r;
}
PH_HD u64 reduce128(U128 z, const Mod &m) { return barrett128(z.hi, z.lo, m); }
// z < 2^123 and 2^59 < q < 2^60: one-word Barrett.  mu = floor(2^123 / q) = (r1:r0) >> 5 fits 64 bits; the quotient
// estimate floor(floor(z / 2^59) mu / 2^64) is at most 3 below floor(z / q), so the remainder z - qhat q lies in
// [0, 4q) < 2^62 and two conditional subtractions finish.  A third of the multiplications of barrett128.
PH_HD u64 reduce123(U128 z, const Mod &m)
{
    const u64 mu = (m.r1 << 59) | (m.r0 >> 5);
    const u64 zh = (z.hi << 5) | (z.lo >> 59);
    const u64 qhat = mulhi(zh, mu);
    u64 r = z.lo - qhat * m.q;
    r = r >= 2 * m.q ? r - 2 * m.q : r;
    r = r >= m.q ? r - m.q : r;
    return r;
}
// z < 2^124 (15 products of residues < 2^60, plus a residue): the same with one bit less of z.  t = floor(floor(z / 2^60)
// mu / 2^64) lies in (z / 2q - 3, z / 2q], so z - 2 t q is in [0, 7q) < 2^63: three conditional subtractions.
PH_HD u64 reduce124(U128 z, const Mod &m)
{
    const u64 mu = (m.r1 << 59) | (m.r0 >> 5);
    const u64 zh = (z.hi << 4) | (z.lo >> 60);
    const u64 qhat = mulhi(zh, mu) << 1;
    u64 r = z.lo - qhat * m.q;
    r = r >= 4 * m.q ? r - 4 * m.q : r;
    r = r >= 2 * m.q ? r - 2 * m.q : r;
    r = r >= m.q ? r - m.q : r;
    return r;
}
PH_HD u64 mulmod(u64 a, u64 b, const Mod &m)
{
    U128 z = mul128(a, b);
    return barrett128(z.hi, z.lo, m);
}
PH_HD u64 addmod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return s >= q ? s - q : s;
}
PH_HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Shoup multiplication by a constant w < q with wsh = floor(w 2^64 / q): any a < 2^64,
// result in [0, 2q).
PH_HD u64 mul_shoup_lazy(u64 a, u64 w, u64 wsh, u64 q) { return a * w - mulhi(a, wsh) * q; }
PH_HD u64 mul_shoup(u64 a, u64 w, u64 wsh, u64 q)
{
    u64 r = mul_shoup_lazy(a, w, wsh, q);
    return r >= q ? r - q : r;
}
// exact floor(a w / q) and a w mod q for a < q (the "integer part / fractional part" split of the
// HPS scaling terms)
PH_HD void divmod_shoup(u64 a, u64 w, u64 wsh, u64 q, u64 &quot, u64 &rem)
{
    u64 qe = mulhi(a, wsh);
    u64 r = a * w - qe * q;
    if (r >= q) {
        r -= q;
        qe += 1;
    }
    quot = qe;
    rem = r;
}

// y / q (y < q) as a fixed-point fraction with 60 fractional bits, error < 2^-59.  Replaces the
// floating-point rounding terms of OpenFHE's HPS base conversions with an integer rule that is
// identical on CPU and GPU (SURVEY.md section 7 "HPS floating-point corrections").
PH_HD u64 fixfrac(u64 y, const Mod &m) { return mulhi(y << m.fshift, m.fconst) >> 3; }
static const u64 FIX_HALF = 1ULL << 59;

}  // namespace piehip
