// kernels_ntt.hip -- negacyclic NTT / inverse NTT over 60-bit RNS primes for gfx950.
//
// Replaces DCRTPoly::SetFormat (OpenFHE) under EvalMult(ct,ct), reference call site
// src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.cpp:123.
// Forward: Cooley-Tukey, natural order in, bit-reversed order out, twiddles psi^bitrev(k).
// Inverse: Gentleman-Sande, bit-reversed in, natural out, scaled by N^-1.
// Harvey lazy butterflies with Shoup twiddles; canonical residues in and out.
//
// One workgroup owns one limb (or one 2^-s0 slice of it) and keeps it in LDS for all of its
// stages: one coalesced HBM read and one write per limb-NTT (16 N bytes, the algorithmic minimum).
// A limb of N = 16384 is 128 KiB of the CU's 160 KiB LDS; N = 32768 runs its outermost stage as a
// streaming pass over global memory and the rest as two 16384-point LDS transforms.
#include "kernels.hpp"

namespace piehip {

static const u32 NTT_LDS_MAX_LOG = 14;  // 2^14 coefficients * 8 B = 128 KiB

struct NttArgs {
    u64 *data;
    const u64 *tables;
    const DevConsts *dc;
    u32 N, logN;
    u32 s0;  // log2 of the number of LDS slices per limb (stages m < 2^s0 run in global memory)
    u32 mod_base, mod_count;
};

// ---------------------------------------------------------------------------------------------
// generic LDS kernel: radix-2 stages with a barrier per stage.  Works for every N >= 64; the
// specialised radix-16 register kernels below take over for the production sizes.
// ---------------------------------------------------------------------------------------------
template <bool INV>
__global__ void __launch_bounds__(1024) ntt_lds_generic(NttArgs a)
{
    extern __shared__ u64 s[];
    const u32 tid = threadIdx.x, nthr = blockDim.x;
    const u32 nl = a.N >> a.s0;                       // coefficients in this slice
    const u32 limb = blockIdx.x >> a.s0, blk = blockIdx.x & ((1u << a.s0) - 1);
    const u32 mod = a.mod_base + limb % a.mod_count;
    const Mod m = a.dc->mod[mod];
    const u64 q = m.q, q2 = 2 * m.q;
    const u64 *tw = a.tables + (size_t)mod * 4 * a.N + (INV ? 2 * (size_t)a.N : 0);
    const u64 *tws = tw + a.N;
    u64 *g = a.data + (size_t)limb * a.N + (size_t)blk * nl;

    for (u32 i = tid; i < nl; i += nthr) s[i] = g[i];
    __syncthreads();

    if (!INV) {
        u32 logt = a.logN - a.s0 - 1;
        for (u32 ml = 1; ml < nl; ml <<= 1, logt--) {
            const u32 t = 1u << logt;
            for (u32 bi = tid; bi < nl / 2; bi += nthr) {
                const u32 i = bi >> logt, j = bi & (t - 1);
                const u32 x = (i << (logt + 1)) + j;
                const u32 k = (ml << a.s0) + blk * ml + i;
                const u64 w = tw[k], ws = tws[k];
                u64 u = s[x];
                u = u >= q2 ? u - q2 : u;
                const u64 v = mul_shoup_lazy(s[x + t], w, ws, q);
                s[x] = u + v;
                s[x + t] = u - v + q2;
            }
            __syncthreads();
        }
        for (u32 i = tid; i < nl; i += nthr) {
            u64 v = s[i];
            v = v >= q2 ? v - q2 : v;
            g[i] = v >= q ? v - q : v;
        }
    } else {
        u32 logt = 0;
        for (u32 hl = nl >> 1; hl >= 1; hl >>= 1, logt++) {
            const u32 t = 1u << logt;
            for (u32 bi = tid; bi < nl / 2; bi += nthr) {
                const u32 i = bi >> logt, j = bi & (t - 1);
                const u32 x = (i << (logt + 1)) + j;
                const u32 k = (hl << a.s0) + blk * hl + i;
                const u64 w = tw[k], ws = tws[k];
                const u64 u = s[x], v = s[x + t];
                const u64 sm = u + v;
                s[x] = sm >= q2 ? sm - q2 : sm;
                s[x + t] = mul_shoup_lazy(u - v + q2, w, ws, q);
            }
            __syncthreads();
        }
        if (a.s0 == 0) {
            for (u32 i = tid; i < nl; i += nthr) g[i] = mul_shoup(s[i], m.n_inv, m.n_inv_sh, q);
        } else {
            for (u32 i = tid; i < nl; i += nthr) {
                u64 v = s[i];
                g[i] = v >= q ? v - q : v;
            }
        }
    }
}

// one outer stage in global memory (only N > 2^14 needs it): forward stage with m groups, or the
// matching inverse stage; canonical in and out.  The last inverse stage also applies N^-1.
template <bool INV>
__global__ void __launch_bounds__(256) ntt_global_stage(NttArgs a, u32 mstage, u32 last_inverse)
{
    const u32 half = a.N >> 1;
    const u32 bi = blockIdx.x * 256 + threadIdx.x;  // butterfly within the limb
    const u32 limb = blockIdx.y;
    if (bi >= half) return;
    const u32 mod = a.mod_base + limb % a.mod_count;
    const Mod m = a.dc->mod[mod];
    const u64 q = m.q;
    const u64 *tw = a.tables + (size_t)mod * 4 * a.N + (INV ? 2 * (size_t)a.N : 0);
    const u64 *tws = tw + a.N;
    u32 logm = 0;
    while ((1u << logm) < mstage) logm++;
    const u32 logt = a.logN - 1 - logm, t = 1u << logt;
    const u32 i = bi >> logt, j = bi & (t - 1);
    const size_t x = (size_t)limb * a.N + ((size_t)i << (logt + 1)) + j;
    const u64 w = tw[mstage + i], ws = tws[mstage + i];
    const u64 u = a.data[x], v = a.data[x + t];
    if (!INV) {
        const u64 p = mul_shoup(v, w, ws, q);
        a.data[x] = addmod(u, p, q);
        a.data[x + t] = submod(u, p, q);
    } else {
        u64 sm = addmod(u, v, q);
        u64 df = mul_shoup(submod(u, v, q), w, ws, q);
        if (last_inverse) {
            sm = mul_shoup(sm, m.n_inv, m.n_inv_sh, q);
            df = mul_shoup(df, m.n_inv, m.n_inv_sh, q);
        }
        a.data[x] = sm;
        a.data[x + t] = df;
    }
}

static PerDeviceOnce g_attr_set[2];

u32 ntt_fast_s0(u32 logN)
{
    const u32 s0 = logN > NTT_LDS_MAX_LOG ? logN - NTT_LDS_MAX_LOG : 0;
    return (logN - s0 >= 12 && logN - s0 <= 14) ? s0 : ~0u;
}

bool launch_ntt_digits(const NttPlan &pl, const u64 *d2, u64 *dig, u32 nb, u32 L, bool sigma, bool folded_layout, hipStream_t st)
{
    if (pl.force_generic || !pl.twp || !pl.twc || ntt_fast_s0(pl.logN) != 0) return false;
    if (sigma && ntt16_applies(pl, folded_layout)) return false;  // the lane order is the 16-coefficient kernel's: digits kernel + its transform
    return launch_ntt_fast(pl.twp, pl.twc, pl.dc, pl.N, pl.logN, 0, dig, nb * L * L, 0, L, false, sigma, pl.transform_cus(), st, d2, L,
                           (sigma && folded_layout) ? 1u : 0u);
}

bool ntt_supports_extra(const NttPlan &pl, bool folded)
{
    if (pl.force_generic || !pl.twp) return false;
    if (ntt16_applies(pl, folded)) return true;
    if (folded) return pl.twc_fold != nullptr;
    return pl.twc != nullptr && ntt_fast_s0(pl.logN) == 0;
}

void launch_ntt(const NttPlan &pl, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inverse, hipStream_t st, bool sigma,
                bool folded, const NttExtra *ex)
{
    if (!nlimbs) return;
    if (ntt16_applies(pl, folded) && launch_ntt16(pl, folded, data, nlimbs, mod_base, mod_count, inverse, sigma, st, ex)) return;
    if (folded) {  // two half-size slices per limb, outer stage done by the neighbouring kernels
        NttExtra exf;
        if (ex) exf = *ex;
        exf.folded = true;
        (void)launch_ntt_fast(pl.twp, pl.twc_fold, pl.dc, pl.N, pl.logN, 1, data, nlimbs, mod_base, mod_count, inverse, sigma,
                              pl.transform_cus(), st, nullptr, 0, 0, &exf);
        return;
    }
    NttArgs a;
    a.data = data;
    a.tables = pl.tables;
    a.dc = pl.dc;
    a.N = pl.N;
    a.logN = pl.logN;
    a.s0 = pl.logN > NTT_LDS_MAX_LOG ? pl.logN - NTT_LDS_MAX_LOG : 0;
    a.mod_base = mod_base;
    a.mod_count = mod_count;
    const u32 nl = pl.N >> a.s0;
    const size_t lds = (size_t)nl * sizeof(u64);
    u32 threads = nl / 2;
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    if (g_attr_set[inverse ? 1 : 0].first_on_current_device()) {
        // a workgroup may use up to 160 KiB of LDS on gfx950; raise the dynamic-LDS cap once
        if (inverse)
            (void)hipFuncSetAttribute((const void *)ntt_lds_generic<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (1 << NTT_LDS_MAX_LOG) * 8);
        else
            (void)hipFuncSetAttribute((const void *)ntt_lds_generic<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (1 << NTT_LDS_MAX_LOG) * 8);
    }
    const dim3 ggrid((pl.N / 2 + 255) / 256, nlimbs);
    const bool fast_ok = !pl.force_generic && pl.twp && pl.twc && pl.logN - a.s0 >= 12;
    if (!inverse) {
        for (u32 ms = 1; ms < (1u << a.s0); ms <<= 1)
            hipLaunchKernelGGL(ntt_global_stage<false>, ggrid, dim3(256), 0, st, a, ms, 0u);
        if (!(fast_ok && launch_ntt_fast(pl.twp, pl.twc, pl.dc, pl.N, pl.logN, a.s0, data, nlimbs, mod_base, mod_count, false, sigma, pl.transform_cus(), st, nullptr, 0, 0, ex)))
            hipLaunchKernelGGL(ntt_lds_generic<false>, dim3(nlimbs << a.s0), dim3(threads), lds, st, a);
    } else {
        if (!(fast_ok && launch_ntt_fast(pl.twp, pl.twc, pl.dc, pl.N, pl.logN, a.s0, data, nlimbs, mod_base, mod_count, true, sigma, pl.transform_cus(), st, nullptr, 0, 0, ex)))
            hipLaunchKernelGGL(ntt_lds_generic<true>, dim3(nlimbs << a.s0), dim3(threads), lds, st, a);
        for (u32 ms = (1u << a.s0) >> 1; ms >= 1; ms >>= 1)
            hipLaunchKernelGGL(ntt_global_stage<true>, ggrid, dim3(256), 0, st, a, ms, ms == 1 ? 1u : 0u);
    }
}

}  // namespace piehip
