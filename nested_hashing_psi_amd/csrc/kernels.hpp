// kernels.hpp -- launch wrappers of the gfx950 kernels behind the C ABI (include/piehip.h).
// Every wrapper only enqueues on `st`; none allocates or synchronises (graph-capturable).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <vector>

#include "params.hpp"

namespace piehip {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one bit per device ordinal and kernel instantiation,
// so that handles on several devices of one process each raise the limit where they launch.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    bool first_on_current_device()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
        const unsigned long long bit = 1ULL << dev;
        return (done.fetch_or(bit) & bit) == 0;
    }
};

// Device-resident NTT tables: for modulus a, tables + a*4*N holds tw | tw_sh | itw | itw_sh (N each).
struct NttPlan {
    const u64 *tables;
    const u64 *twp;       // interleaved {w, w_shoup} pairs: per modulus [fwd N pairs][inv N pairs]
    const u64 *twc;       // pass-C twiddles in kernel order: per modulus [fwd][inv], see build_twc_table
    const u64 *twc_fold;  // the same for the folded configuration (two half-size slices per limb), or null
    const u64 *twk16 = nullptr;       // ntt16_kernel.h: kernel-ordered pairs of passes 3, 4, one 2^13 slice per limb (ring 2^13), or null
    const u64 *twk16_fold = nullptr;  // ... two folded slices per limb: of 2^13 (ring 2^14) or 2^14 coefficients (ring 2^15), or null
    const DevConsts *dc;  // device pointer
    u32 N, logN;
    u32 num_cus;
    u32 max_slots = 0;    // piehip_set_transform_slots: cap on the persistent transform grids (workgroups), 0 = every slot
    bool force_generic;   // tests: route every size through the radix-2 LDS kernel
    // CUs the persistent transform grids may fill (two workgroup slots of 512 threads, or one of 1024, per CU)
    u32 transform_cus() const
    {
        if (!max_slots) return num_cus;
        const u32 c = (max_slots + 1) / 2;
        return c < num_cus ? (c ? c : 1u) : num_cus;
    }
};

// In-place negacyclic NTT over `nlimbs` limbs [nlimbs][N]; limb i uses modulus mod_base + i % mod_count.
// (replaces DCRTPoly::SetFormat under BatchedFHEHIPPIE.cpp:123)
void build_twc_table(const u64 *nat_pairs, u32 logN, u32 s0, std::vector<u64> &out);
u32 ntt_fast_s0(u32 logN);  // log2 slices per limb the fast kernel would use, or ~0u if it does not apply
// Optional extras of the register-blocked kernel (ciphertext multiplication, kernels_ntt_fast.hip):
//   inverse, standard order in:  limbs [nb][copy_K][2][copy_L]; the lane-ordered EVALUATION input of operand 0 of every
//                                bin is also written to the Q limbs of copy_out[nb][4][copy_M][N] (slots 0, 1)
//   forward, lane order:         lazy_out -- no final normalisation
//   forward:                     nlimbs counts a compact enumeration of [nb][4][skip_M] that omits limbs < skip_L of
//                                slots 0 and 1 (they already hold EVALUATION data)
struct NttExtra {
    u64 *copy_out = nullptr;
    u32 copy_K = 1, copy_L = 1, copy_M = 1;
    u32 skip_L = 0, skip_M = 0;
    bool lazy_out = false;  // forward, lane order: leave the residues in [0, 8q) (the consumer reduces anyway)
    bool folded = false;    // set by launch_ntt(.., folded): inverse transforms then hand over unnormalised [0, 4q) residues
    bool x_lane_in = false; // 16-coefficient kernel, inverse, standard order in: the operand-0 polynomials are READ, lane-ordered, from
                            // where copy_out would have put them (stage A wrote them there: StageAXOut) and nothing is copied
};
// The 16-coefficients-per-thread kernel (ntt16_kernel.h, kernels_ntt16.hip) for slices of 2^13 and 2^14 coefficients: every transform whose
// EVALUATION side is in lane order, and every inverse transform.  Its lane order differs from the 32-coefficient kernel's:
// a context uses one of the two for all lane-ordered arrays (ntt16_applies).  Returns false when it does not apply.
void build_twk16_table(const u64 *nat_pairs, u32 s0, u32 slice_log, std::vector<u64> &out);  // slices of 2^13 or 2^14
void ntt16_sigma_inverse_map(u32 logN, u32 s0, std::vector<u32> &map);
// dg (forward, lane order, mod_base 0, mod_count L): the BV key-switch digits of nb polynomials join the launch -- limb (bin, i, j)
// of dig[nb][L][L][N] = transform of the centred lift into q_j of residue limb i of the COEFFICIENT polynomial d2 + bin * stride2
// (replaces launch_digits + a second transform launch)
struct Ntt16Digits {
    const u64 *d2;
    size_t stride2;
    u64 *dig;
    u32 nb, L;
};
bool launch_ntt16(const NttPlan &pl, bool folded, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inverse, bool sigma,
                  hipStream_t st, const NttExtra *ex, const Ntt16Digits *dg = nullptr);
inline bool ntt16_applies(const NttPlan &pl, bool folded)
{
    return !pl.force_generic && pl.twp && (folded ? (pl.twk16_fold && (pl.logN == 14 || pl.logN == 15)) : (pl.twk16 && pl.logN == 13));
}
bool launch_ntt_fast(const u64 *twp, const u64 *twc, const DevConsts *dc, u32 N, u32 logN, u32 s0, u64 *data, u32 nlimbs, u32 mod_base,
                     u32 mod_count, bool inverse, bool sigma, u32 num_cus, hipStream_t st, const u64 *lift_src = nullptr,
                     u32 lift_L = 0, u32 sigma_split = 0, const NttExtra *ex = nullptr);
// true when launch_ntt(.., folded) / launch_ntt(..) for this plan goes to the register-blocked kernel and honours NttExtra
bool ntt_supports_extra(const NttPlan &pl, bool folded);
// forward transform of the BV digits with the digit lift fused into the load (no digits kernel):
// d2[nb][L][N] COEFFICIENT -> dig[nb][L(i)][L(j)][N] EVALUATION.  Returns false if the register-blocked
// kernel does not cover this ring dimension as one slice (callers then use launch_digits + launch_ntt).
// folded_layout: write the lane order of the folded configuration (two slices per limb), matching arrays
// produced by launch_ntt(.., folded = true)
bool launch_ntt_digits(const NttPlan &pl, const u64 *d2, u64 *dig, u32 nb, u32 L, bool sigma, bool folded_layout,
                       hipStream_t st);
void ntt_sigma_inverse_map(u32 logN, u32 s0, std::vector<u32> &map);  // s0 = ~0u: identity
// sigma: keep the EVALUATION side in the register-blocked kernel's lane order (internal arrays only; ignored,
// i.e. standard order, when that kernel does not apply -- ntt_sigma_inverse_map is then the identity)
// folded: the outermost stage is NOT done here (the caller's neighbouring kernels apply it, kernels_pie.hip
// "Outer-stage folding"); the limb is transformed as two independent half-size slices.  Requires pl.twc_fold.
void launch_ntt(const NttPlan &pl, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inverse, hipStream_t st,
                bool sigma = false, bool folded = false, const NttExtra *ex = nullptr);

// Stage A: acc[b][K][2][L][N] = sum_j idx[h][j] (.) db[h][beta][j] + minus    (BatchedFHEHIPPIE.cpp:101-116)
// small_moduli: every RNS modulus is < 2^60 (enables the v_mad_u64_u32 column-accumulator kernel)
// bstride: bin-layer count of the database array db[K][bstride][E][L][N] when only b <= bstride layers (starting at the
// layer db points to) are evaluated; 0 = b.  h0, hn: only the inner hash functions [h0, h0 + hn) (hn = 0: all from h0)
// nq, q: the accumulators are row q of a batch of nq queries, acc[b][nq][K][2][L][N] (nq = 1: the layout above)
// xo (column-accumulator kernels only): the accumulators of inner hash function 0 -- operand X of the first ciphertext product --
// are not written to acc but, lane-ordered (ntt16_kernel.h), into the Q limbs of the QP operand array xo->out[row][4][M][N],
// slots 0, 1: where the tensor product reads them.  The inverse transform then takes them from there (NttExtra::x_lane_in)
// instead of writing that copy itself.
struct StageAXOut {
    u64 *out = nullptr;
    u32 M = 0, logns = 0;  // limbs per polynomial of the QP array; log2 of the transform's slice length
};
void launch_stage_a(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus,
                    const u64 *db, u64 *acc, hipStream_t st, bool small_moduli, u32 bstride = 0, u32 h0 = 0, u32 hn = 0, u32 nq = 1,
                    u32 q = 0, const StageAXOut *xo = nullptr);
// Stage A of a batch of nq <= STAGE_A_MAX_QUERIES queries on one database: acc[b][nq][K][2][L][N].  Where the column-accumulator
// kernel applies (small_moduli) the queries go through it in groups of two to four, each group reading the database once
static const u32 STAGE_A_MAX_QUERIES = 8;
struct StageAQueries {
    const u64 *idx[STAGE_A_MAX_QUERIES];
    const u64 *minus[STAGE_A_MAX_QUERIES];
};
void launch_stage_a_batch(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const StageAQueries &qs, u32 nq, const u64 *db,
                          u64 *acc, hipStream_t st, bool small_moduli, u32 bstride = 0, u32 h0 = 0, u32 hn = 0,
                          const StageAXOut *xo = nullptr);

// (per-host-thread switch, set from the context before its launches: all Q and P moduli lie in (2^59, 2^60), which lets
// the base-conversion and key-switch kernels use the carry-free v_mad_u64_u32 column accumulators and one-word Barrett)
void set_small_moduli(bool v);

// Base conversions, COEFFICIENT format.  Polynomial (o, c), o < n_outer, c < 2,
// is read at in + o*in_stride_outer + c*in_stride_inner ([L][N] limbs) and written to
// out + ((o*out_polys + out_slot + c)*M)*N ([M][N] limbs).
// skip_q: leave the Q limbs of the output alone (they already hold the operand's EVALUATION form, see NttExtra)
void launch_expand_q_to_qp(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t in_stride_outer,
                           size_t in_stride_inner, u32 n_outer, u64 *out, u32 out_polys, u32 out_slot, hipStream_t st,
                           bool fold = false, bool skip_q = false);
void launch_scale_pq_expand(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t in_stride_outer,
                            size_t in_stride_inner, u32 n_outer, u64 *out, u32 out_polys, u32 out_slot, hipStream_t st,
                            bool fold = false);
// both conversions of a ciphertext multiplication in one launch: X (polynomials at x + o*sx + c*si) centred-lifted into slots 0, 1
// of out[n_outer][4][M][N], Y (at y + o*sy + c*si) scaled by P/Q into slots 2, 3
void launch_expand_both(const DevConsts *dc, u32 N, u32 L, const u64 *x, size_t sx, const u64 *y, size_t sy, size_t si, u32 n_outer,
                        u64 *out, hipStream_t st, bool fold = false, bool skip_q = false);
// e[nb][4][M][N] (a0 a1 b0 b1, EVALUATION) -> d[nb][3][M][N]
void launch_tensor(const DevConsts *dc, u32 N, u32 M, const u64 *e, u64 *d, u32 nb, hipStream_t st);
// d[nb][3][M][N] (COEFFICIENT) -> components 0,1 to out01 + bin*stride01 + c*L*N, component 2 to out2 + bin*stride2
// fold: the outermost NTT stage of the neighbouring transforms is applied here (see kernels_pie.hip, "Outer-stage
// folding"); fold_comp2: component 2 also feeds a forward transform directly (3-component output)
void launch_scale_round(const DevConsts *dc, u32 N, u32 L, const u64 *d, u32 nb, u64 *out01, size_t stride01, u64 *out2,
                        size_t stride2, hipStream_t st, bool fold = false, bool fold_comp2 = false);
// BV digits: d2c at d2 + bin*stride2 ([L][N], COEFFICIENT) -> dig[nb][L(i)][L(j)][N] (centred lift of residue i into q_j)
void launch_digits(const DevConsts *dc, u32 N, u32 L, const u64 *d2, size_t stride2, u32 nb, u64 *dig, hipStream_t st,
                   bool fold = false);
// out[bin][c][j] = (d01[bin][c][j] + sum_i dig[bin][i][j] (.) key[i][c][j]) (.) mask[bin][j]   (mask may be null)
// out_map (may be null): coefficient n of the result is written to position out_map[n] (lane order -> standard)
// key_group > 1: ciphertext `bin` uses key + (bin % key_group) * key_stride (EvalMerge: one rotation key per position)
// mask_div > 1: ciphertext `bin` takes mask[bin / mask_div] (a query batch: mask_div queries per bin layer)
// sigma_T != 0: out_map is the lane order of a transform with sigma_T threads per slice and sigma_kp coefficient pairs per thread
// (16: kernels_ntt_fast.hip, 8: ntt16_kernel.h); stores then go through an LDS tile
void launch_relin_mac(const DevConsts *dc, u32 N, u32 L, const u64 *d01, size_t stride01, const u64 *dig, const u64 *key,
                      const u64 *mask, u64 *out, u32 nb, hipStream_t st, const u32 *out_map = nullptr, size_t key_stride = 0,
                      u32 key_group = 1, u32 sigma_T = 0, u32 sigma_kp = 16, u32 mask_div = 1);
// rotation-based PIE (FHEHIPPIE.cpp:61-77), see kernels_pie.hip
void launch_bcast_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, size_t xs, u32 group, const u64 *pt, size_t ps_outer,
                            size_t ps_inner, u64 *out, u32 nct, hipStream_t st);
void launch_rot_prepare(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u32 *maps, u32 map_group, bool acc, bool identity_first,
                        u64 *d01, u64 *d2, u32 nct, hipStream_t st);
void launch_sum_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, u32 group, const u64 *pt, size_t pt_stride, u64 *out,
                          size_t out_stride, u32 ngroups, hipStream_t st);
// writes `value` to a word of page-locked host memory (its device address) behind everything queued on `st` so far
void launch_host_flag(u64 *flag_dev, u64 value, hipStream_t st);
// element-wise helpers on nct ciphertexts [nct][2][L][N]
void launch_ct_add(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u64 *y, u64 *out, u32 nct, hipStream_t st);
// (pt_div > 1: ciphertext i takes plaintext i / pt_div)
void launch_ct_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u64 *pt, size_t pt_stride, u64 *out,
                         u32 nct, hipStream_t st, u32 pt_div = 1);
// out[r][p] = in[r][map[p]] for nrows limbs
void launch_permute(u32 N, const u64 *in, const u32 *map, u64 *out, u32 nrows, hipStream_t st);
// packed encoding: slots[npt][B] -> u[npt][N] residues mod t at their EVALUATION positions
void launch_encode_scatter(const DevConsts *dc, u32 N, u32 M, const int64_t *slots, u32 B, const u32 *inv_pos, u64 *u,
                           u32 npt, hipStream_t st);
// coefficients mod t [npt][N] -> centred lift into every q_i: out[npt][L][N]
void launch_encode_lift(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *u, u64 *out, u32 npt, hipStream_t st);

// ---- offline phase: nested hashing and database gather on the device (kernels_hash.hip) -----------------------
size_t hash_sort_temp_bytes(u32 n, u32 e);
hipError_t launch_hash_build(const u64 *d_tab, const u64 *d_items, u32 n, u32 k, u32 e, u32 K, u32 b, u32 E, u64 evict_seed,
                             u64 shuffle_seed, u64 *d_tbl, u32 *d_keys, u32 *d_vals, u32 *d_start, void *d_temp, size_t temp_bytes,
                             u32 *d_fail, hipStream_t st);
void launch_shuffle_rows(u64 *d_tbl, u32 rows, u32 b, u32 E, u64 seed, hipStream_t st);
void launch_gather_slots(const u64 *d_tbl, u32 B, u32 K, u32 b, u32 E, u64 t, int64_t *d_slots, u32 *d_fail, hipStream_t st);
void launch_mask_slots(u64 t, u32 b, u32 B, u64 seed, int64_t *d_out, hipStream_t st);

// ---- client harness (kernels_client.hip) ----------------------------------------------------------------------------
void launch_enc_message(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *coeff_t, const int32_t *e, u64 *em, u32 nct, hipStream_t st);
void launch_enc_finish(const DevConsts *dc, u32 N, u32 L, const u64 *em, const u64 *sk, const u64 *a_in, u64 *out, u32 nct,
                       hipStream_t st);
void launch_ks_finish(const DevConsts *dc, u32 N, u32 L, const u64 *e, const u64 *sk, const u64 *s_from, u64 *ks, hipStream_t st);
void launch_square(const DevConsts *dc, u32 N, u32 L, const u64 *s, u64 *s2, hipStream_t st);
void launch_dec_dot(const DevConsts *dc, u32 N, u32 L, const u64 *ct, const u64 *sk, u64 *xs, u32 nct, hipStream_t st);
void launch_dec_round(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *xs, u64 *coeff_t, u32 nct, hipStream_t st);
void launch_decode_gather(const DevConsts *dc, u32 N, u32 M, const u64 *u, const u32 *slot_pos, u32 B, int64_t *slots, u32 nct,
                          hipStream_t st);

}  // namespace piehip
