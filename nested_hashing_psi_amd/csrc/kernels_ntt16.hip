// kernels_ntt16.hip -- launch side of the 16-coefficients-per-thread NTT (ntt16_kernel.h) for slices of 2^13 coefficients:
// ring 2^13 as one slice per limb, ring 2^14 as two slices with the outermost stage folded into the neighbouring kernels.
// Replaces DCRTPoly::SetFormat under EvalMult(ct, ct) (reference BatchedFHEHIPPIE.cpp:123).
#include "kernels.hpp"
#include "ntt16_kernel.h"

namespace piehip {

void build_twk16_table(const u64 *nat_pairs, u32 s0, u32 slice_log, std::vector<u64> &out)
{
    if (slice_log == 14)
        ntt16::build_twk_table_t<14>(nat_pairs, s0, out);
    else
        ntt16::build_twk_table_t<13>(nat_pairs, s0, out);
}

void ntt16_sigma_inverse_map(u32 logN, u32 s0, std::vector<u32> &map)
{
    const u32 N = 1u << logN, n = N >> s0;
    map.resize(N);
    for (u32 p = 0; p < N; p++) map[p] = (p / n) * n + ntt16::lane_to_std_t(p % n, n / 16);
}

template <u32 LOGNS>
static void launch_ntt16_t(const ntt16::Args &a, bool inverse, bool lift, u32 num_cus, hipStream_t st)
{
    typedef ntt16::Geo<LOGNS> G;
    constexpr size_t lds = (size_t)G::LDS_WORDS * sizeof(u64);
    static PerDeviceOnce attr[3];
    if (attr[inverse ? 1 : (lift ? 2 : 0)].first_on_current_device()) {
        if (inverse)
            (void)hipFuncSetAttribute((const void *)ntt16::ntt16_kernel_t<LOGNS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else if (lift)
            (void)hipFuncSetAttribute((const void *)ntt16::ntt16_kernel_t<LOGNS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            (void)hipFuncSetAttribute((const void *)ntt16::ntt16_kernel_t<LOGNS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // resident workgroups per CU: two of 512 threads (68 KiB of LDS, 128 VGPRs each) or one of 1024 (136 KiB); persistent beyond
    const u32 slots = (LOGNS == 13 ? 2u : 1u) * num_cus;
    const u32 grid = a.nitems < slots ? a.nitems : slots;
    if (inverse)
        hipLaunchKernelGGL((ntt16::ntt16_kernel_t<LOGNS, true>), dim3(grid), dim3(G::T), lds, st, a);
    else if (lift)
        hipLaunchKernelGGL((ntt16::ntt16_kernel_t<LOGNS, false, true>), dim3(grid), dim3(G::T), lds, st, a);
    else
        hipLaunchKernelGGL((ntt16::ntt16_kernel_t<LOGNS, false>), dim3(grid), dim3(G::T), lds, st, a);
}

bool launch_ntt16(const NttPlan &pl, bool folded, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inverse, bool sigma,
                  hipStream_t st, const NttExtra *ex, const Ntt16Digits *dg)
{
    const u64 *twk = folded ? pl.twk16_fold : pl.twk16;
    const u32 s0 = folded ? 1u : 0u;
    const u32 slice_log = pl.logN - s0;
    if (pl.force_generic || !twk || !pl.twp || (slice_log != 13 && slice_log != 14)) return false;
    if (!inverse && !sigma) return false;  // standard-order output is the 32-coefficient kernel's
    ntt16::Args a;
    a.data = data;
    a.twp = reinterpret_cast<const ntt16::u64x2 *>(pl.twp);
    a.twk = reinterpret_cast<const ntt16::u64x2 *>(twk);
    a.dc = pl.dc;
    a.N = pl.N;
    a.s0 = s0;
    a.nitems = nlimbs << s0;
    a.mod_base = mod_base;
    a.mod_count = mod_count;
    a.flags = 0;
    if (inverse && !sigma) a.flags |= ntt16::F_STD_IN;
    if (inverse && folded) a.flags |= ntt16::F_FOLDED;
    if (inverse && !sigma && ex && ex->copy_out && ex->x_lane_in) a.flags |= ntt16::F_X_LANE_IN;
    if (!inverse && ex && ex->lazy_out) a.flags |= ntt16::F_LAZY_OUT;
    a.skip_L = (ex && !inverse) ? ex->skip_L : 0;
    a.skip_M = ex ? ex->skip_M : 0;
    a.copy_out = (ex && inverse && !sigma) ? ex->copy_out : nullptr;
    a.copy_K = ex ? ex->copy_K : 1;
    a.copy_L = ex ? ex->copy_L : 1;
    a.copy_M = ex ? ex->copy_M : 1;
    a.lift_first = ~0u;
    a.data2 = nullptr;
    a.lift_src = nullptr;
    a.lift_stride = 0;
    a.lift_L = 1;
    if (dg && !inverse) {  // the key-switch digits ride in the same launch: nb * L * L more limbs, lifted in the load phase
        a.lift_first = a.nitems;
        a.nitems += (dg->nb * dg->L * dg->L) << s0;
        a.data2 = dg->dig;
        a.lift_src = dg->d2;
        a.lift_stride = dg->stride2;
        a.lift_L = dg->L;
    }
    const bool lift = a.lift_first != ~0u;
    if (slice_log == 14)
        launch_ntt16_t<14>(a, inverse, lift, pl.transform_cus(), st);
    else
        launch_ntt16_t<13>(a, inverse, lift, pl.transform_cus(), st);
    return true;
}

}  // namespace piehip
