// kernels_pie.hip -- the coefficient-wise kernels of BatchedFHEHIPPIE::run() for gfx950:
// fused ct x pt multiply-accumulate (stage A), HPS base conversions, tensor product, BV digit
// decomposition and key-switch accumulation, mask multiply, automorphism permutation, packed
// encoding.  All are streaming u64 modular arithmetic: one thread per coefficient, consecutive
// lanes on consecutive coefficients (coalesced 8-byte or 16-byte lanes), constants scalar-loaded
// from one DevConsts block.  Reference call sites: BatchedFHEHIPPIE.cpp:101-127.
#include "kernels.hpp"
#include "madasm.h"

namespace piehip {

static const u32 TPB = 256;

// ---------------------------------------------------------------------------------------------
// Stage A (rows A3+A4): acc[beta][h][c][l][n] = sum_j idx[h][j][c][l][n] * db[h][beta][j][l][n] + minus[c][l][n]
//
// HBM-bound: every database plaintext limb is read exactly once per run() (b K E L W bytes, 196 MiB at C3).
// A thread owns two adjacent coefficients (16-byte lanes) of one (h, limb) and BPT bin layers: the two index
// ciphertext components are loaded once per j and reused for all BPT layers, so index traffic is
// (b / BPT) K E 2L W instead of b K E 2L W; the BPT database loads per j are independent streams in flight.
// 128-bit lazy accumulation (products < 2^120, E < 128 terms), one Barrett reduction at the end.
// ---------------------------------------------------------------------------------------------
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <int BPT>
__global__ void __launch_bounds__(TPB) stage_a_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                      const u64 *__restrict__ idx, const u64 *__restrict__ minus,
                                                      const u64 *__restrict__ db, u64 *__restrict__ acc)
{
    const u32 n = 2 * (blockIdx.x * TPB + threadIdx.x);
    const u32 l = blockIdx.y;
    const u32 groups = b / BPT;
    const u32 h = blockIdx.z / groups, beta0 = (blockIdx.z % groups) * BPT;
    if (n >= N) return;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + n;
    const u64 *pd = db + (((size_t)h * b + beta0) * E) * LN + (size_t)l * N + n;
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
        for (int t = 0; t < BPT; t++) d[t] = *reinterpret_cast<const u64x2 *>(pd + (size_t)t * bin_stride + (size_t)j * LN);
#pragma unroll
        for (int t = 0; t < BPT; t++) {
            mac128(a[t][0][0], i0.x, d[t].x);
            mac128(a[t][0][1], i0.y, d[t].y);
            mac128(a[t][1][0], i1.x, d[t].x);
            mac128(a[t][1][1], i1.y, d[t].y);
        }
        if ((j & 127) == 127) {
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
        u64 *po = acc + (((size_t)(beta0 + t) * K + h) * 2) * LN + (size_t)l * N + n;
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
// < 2^60 and E <= COLACC_MAX_TOTAL (one carry sweep after 8 terms keeps the columns from overflowing).
template <int BPT, int CPT>
__global__ void __launch_bounds__(TPB) stage_a_mad_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                          const u64 *__restrict__ idx, const u64 *__restrict__ minus,
                                                          const u64 *__restrict__ db, u64 *__restrict__ acc)
{
    // CPT coefficients per thread: 2 -> 16-byte lanes; 1 -> 8-byte lanes, half the accumulator registers
    const u32 n = CPT * (blockIdx.x * TPB + threadIdx.x);
    const u32 l = blockIdx.y;
    const u32 groups = b / BPT;
    const u32 h = blockIdx.z / groups, beta0 = (blockIdx.z % groups) * BPT;
    if (n >= N) return;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + n;
    const u64 *pd = db + (((size_t)h * b + beta0) * E) * LN + (size_t)l * N + n;
    const size_t bin_stride = (size_t)E * LN;
    ColAcc a[BPT][2][CPT];
#pragma unroll
    for (int t = 0; t < BPT; t++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < CPT; e++) a[t][c][e] = ColAcc{0, 0, 0};
    for (u32 j0 = 0; j0 < E; j0 += COLACC_MAX_TERMS) {
        const u32 j1 = j0 + COLACC_MAX_TERMS < E ? j0 + COLACC_MAX_TERMS : E;
        for (u32 j = j0; j < j1; j++) {
            u64 iv[2][CPT], dv[BPT][CPT];
            if (CPT == 2) {
                const u64x2 i0 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN);
                const u64x2 i1 = *reinterpret_cast<const u64x2 *>(pi + (size_t)j * 2 * LN + LN);
                iv[0][0] = i0.x, iv[0][CPT - 1] = i0.y, iv[1][0] = i1.x, iv[1][CPT - 1] = i1.y;
#pragma unroll
                for (int t = 0; t < BPT; t++) {
                    const u64x2 d = *reinterpret_cast<const u64x2 *>(pd + (size_t)t * bin_stride + (size_t)j * LN);
                    dv[t][0] = d.x, dv[t][CPT - 1] = d.y;
                }
            } else {
                iv[0][0] = pi[(size_t)j * 2 * LN];
                iv[1][0] = pi[(size_t)j * 2 * LN + LN];
#pragma unroll
                for (int t = 0; t < BPT; t++) dv[t][0] = pd[(size_t)t * bin_stride + (size_t)j * LN];
            }
            Split30 is[2][CPT];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < CPT; e++) is[c][e] = split30(iv[c][e]);
#pragma unroll
            for (int t = 0; t < BPT; t++)
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    const Split30 ds = split30(dv[t][e]);
                    colacc_mac(a[t][0][e], is[0][e], ds);
                    colacc_mac(a[t][1][e], is[1][e], ds);
                }
        }
        if (j1 < E) {  // another chunk follows: make room in the low columns
#pragma unroll
            for (int t = 0; t < BPT; t++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int e = 0; e < CPT; e++) colacc_carry(a[t][c][e]);
        }
    }
#pragma unroll
    for (int t = 0; t < BPT; t++) {
        u64 *po = acc + (((size_t)(beta0 + t) * K + h) * 2) * LN + (size_t)l * N + n;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int e = 0; e < CPT; e++)
                po[(size_t)c * LN + e] = addmod(reduce128(colacc_value(a[t][c][e]), m), minus[(size_t)c * LN + (size_t)l * N + n + e], m.q);
    }
}

void launch_stage_a(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus,
                    const u64 *db, u64 *acc, hipStream_t st, bool small_moduli)
{
    // bin layers per thread: the largest divisor of b that keeps the accumulators in registers
    const bool mad = small_moduli && E <= COLACC_MAX_TOTAL;
    const int cap = mad ? 7 : 8;
    int bpt = 1;
    for (int c = cap; c >= 1; c--)
        if (b % c == 0) {
            bpt = c;
            break;
        }
    // the mad kernel keeps 6 accumulator registers per (layer, component, coefficient): above 4 layers per
    // thread it handles one coefficient per thread (8-byte lanes) to stay at >= 2 waves per SIMD
    const int cpt = (mad && bpt > 4) ? 1 : 2;
    dim3 grid((N / cpt + TPB - 1) / TPB, L, K * (b / bpt));
#define SA(B_)                                                                                                       \
    do {                                                                                                             \
        if (mad && cpt == 1) hipLaunchKernelGGL((stage_a_mad_kernel<B_, 1>), grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc); \
        else if (mad) hipLaunchKernelGGL((stage_a_mad_kernel<B_, 2>), grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc); \
        else hipLaunchKernelGGL(stage_a_kernel<B_>, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc);  \
    } while (0)
    switch (bpt) {
        case 8: hipLaunchKernelGGL(stage_a_kernel<8>, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc); break;
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

// ---------------------------------------------------------------------------------------------
// Base conversions (row A6).  One thread per coefficient reads its L (or 2L+1) residues at limb
// stride N and writes every output limb.  Rounding terms use the 60-bit fixed-point rule of
// modarith.h (identical to oracle/pie_oracle.c: po_expand_q_to_qp, po_scale_pq_expand, po_scale_round_tp).
// ---------------------------------------------------------------------------------------------
// centred CRT lift of y_i-weighted residues from a source basis into target modulus `tm`:
//   sum_i y_i * hat[i] - v * prodmod
template <u32 MAXS>
__device__ __forceinline__ u64 crt_out(const u64 *y, u32 ns, const u64 *hat, u32 hat_stride, u64 v, u64 prodmod,
                                       const Mod &tm)
{
    U128 acc = {0, 0};
    for (u32 i = 0; i < ns; i++) mac128(acc, y[i], hat[(size_t)i * hat_stride]);
    mac128(acc, v, tm.q - prodmod);  // - v * prodmod (mod tm); v <= ns: one Barrett reduction for the whole sum
    return reduce128(acc, tm);
}

__global__ void __launch_bounds__(TPB) expand_q_to_qp_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ in,
                                                             size_t so, size_t si, u64 *__restrict__ out, u32 out_polys,
                                                             u32 out_slot)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 o = blockIdx.y >> 1, c = blockIdx.y & 1;
    const u32 M = 2 * L + 1, Lp = L + 1;
    const u64 *pin = in + (size_t)o * so + (size_t)c * si + n;
    u64 *pout = out + ((size_t)(o * out_polys + out_slot + c) * M) * N + n;
    u64 y[MAX_L];
    u64 fsum = 0;
    for (u32 i = 0; i < L; i++) {
        const u64 x = pin[(size_t)i * N];
        pout[(size_t)i * N] = x;
        y[i] = mul_shoup(x, dc->qhat_inv[i], dc->qhat_inv_sh[i], dc->mod[i].q);
        fsum += fixfrac(y[i], dc->mod[i]);
    }
    const u64 v = (fsum + FIX_HALF) >> 60;
    for (u32 j = 0; j < Lp; j++)
        pout[(size_t)(L + j) * N] = crt_out<MAX_L>(y, L, &dc->qhat_modp[0][j], 8, v, dc->Q_modp[j], dc->mod[L + j]);
}

__global__ void __launch_bounds__(TPB) scale_pq_expand_kernel(const DevConsts *dc, u32 N, u32 L,
                                                              const u64 *__restrict__ in, size_t so, size_t si,
                                                              u64 *__restrict__ out, u32 out_polys, u32 out_slot)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 o = blockIdx.y >> 1, c = blockIdx.y & 1;
    const u32 M = 2 * L + 1, Lp = L + 1;
    const u64 *pin = in + (size_t)o * so + (size_t)c * si + n;
    u64 *pout = out + ((size_t)(o * out_polys + out_slot + c) * M) * N + n;
    u64 y[MAX_L];
    u64 fsum = 0;
    U128 itot = {0, 0};
    for (u32 i = 0; i < L; i++) {
        const Mod &mi = dc->mod[i];
        y[i] = mul_shoup(pin[(size_t)i * N], dc->qhat_inv[i], dc->qhat_inv_sh[i], mi.q);
        // y_i P / q_i = y_i floor(P/q_i) + floor(y_i w_i / q_i) + (y_i w_i mod q_i) / q_i
        u64 fl, z;
        divmod_shoup(y[i], dc->P_modq[i], dc->P_modq_sh[i], mi.q, fl, z);
        add128(itot, U128{fl, 0});
        fsum += fixfrac(z, mi);
    }
    add128(itot, U128{(fsum + FIX_HALF) >> 60, 0});
    u64 yp[MAX_L + 1];
    u64 fs2 = 0;
    for (u32 j = 0; j < Lp; j++) {
        const Mod &pj = dc->mod[L + j];
        U128 acc = {0, 0};
        for (u32 i = 0; i < L; i++) mac128(acc, y[i], dc->PI_modp[i][j]);
        add128(acc, itot);
        const u64 r = reduce128(acc, pj);
        pout[(size_t)(L + j) * N] = r;
        yp[j] = mul_shoup(r, dc->phat_inv[j], dc->phat_inv_sh[j], pj.q);
        fs2 += fixfrac(yp[j], pj);
    }
    const u64 v = (fs2 + FIX_HALF) >> 60;
    for (u32 i = 0; i < L; i++)
        pout[(size_t)i * N] = crt_out<MAX_L + 1>(yp, Lp, &dc->phat_modq[0][i], 8, v, dc->P_modq[i], dc->mod[i]);
}

static void launch_expand_common(bool scale, const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si,
                                 u32 n_outer, u64 *out, u32 out_polys, u32 out_slot, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, n_outer * 2);
    if (scale)
        hipLaunchKernelGGL(scale_pq_expand_kernel, grid, dim3(TPB), 0, st, dc, N, L, in, so, si, out, out_polys, out_slot);
    else
        hipLaunchKernelGGL(expand_q_to_qp_kernel, grid, dim3(TPB), 0, st, dc, N, L, in, so, si, out, out_polys, out_slot);
}
void launch_expand_q_to_qp(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si, u32 n_outer, u64 *out,
                           u32 out_polys, u32 out_slot, hipStream_t st)
{
    launch_expand_common(false, dc, N, L, in, so, si, n_outer, out, out_polys, out_slot, st);
}
void launch_scale_pq_expand(const DevConsts *dc, u32 N, u32 L, const u64 *in, size_t so, size_t si, u32 n_outer, u64 *out,
                            u32 out_polys, u32 out_slot, hipStream_t st)
{
    launch_expand_common(true, dc, N, L, in, so, si, n_outer, out, out_polys, out_slot, st);
}

// ---------------------------------------------------------------------------------------------
// Tensor product over QP (row A5 step 4): d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) tensor_kernel(const DevConsts *dc, u32 N, u32 M, const u64 *__restrict__ e,
                                                     u64 *__restrict__ d)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 a = blockIdx.y, bin = blockIdx.z;
    const Mod m = dc->mod[a];
    const size_t MN = (size_t)M * N;
    const u64 *pe = e + (size_t)bin * 4 * MN + (size_t)a * N + n;
    u64 *pd = d + (size_t)bin * 3 * MN + (size_t)a * N + n;
    const u64 a0 = pe[0], a1 = pe[MN], b0 = pe[2 * MN], b1 = pe[3 * MN];
    pd[0] = mulmod(a0, b0, m);
    U128 x = mul128(a0, b1);
    mac128(x, a1, b0);
    pd[MN] = reduce128(x, m);
    pd[2 * MN] = mulmod(a1, b1, m);
}
void launch_tensor(const DevConsts *dc, u32 N, u32 M, const u64 *e, u64 *d, u32 nb, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, M, nb);
    hipLaunchKernelGGL(tensor_kernel, grid, dim3(TPB), 0, st, dc, N, M, e, d);
}

// ---------------------------------------------------------------------------------------------
// Scale-and-round by t/P from QP into Q (row A6)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) scale_round_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ d,
                                                          u64 *__restrict__ out01, size_t stride01,
                                                          u64 *__restrict__ out2, size_t stride2)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 comp = blockIdx.y, bin = blockIdx.z;
    const u32 M = 2 * L + 1, Lp = L + 1;
    const u64 *pin = d + ((size_t)(bin * 3 + comp) * M) * N + n;
    u64 *pout = comp < 2 ? out01 + (size_t)bin * stride01 + (size_t)comp * L * N + n : out2 + (size_t)bin * stride2 + n;
    u64 yp[MAX_L + 1];
    u64 fsum = 0;
    U128 itot = {0, 0};
    for (u32 j = 0; j < Lp; j++) {
        const Mod &pj = dc->mod[L + j];
        yp[j] = mul_shoup(pin[(size_t)(L + j) * N], dc->qp_hat_inv[L + j], dc->qp_hat_inv_sh[L + j], pj.q);
        u64 fl, z;
        divmod_shoup(yp[j], dc->tQ_modp[j], dc->tQ_modp_sh[j], pj.q, fl, z);
        add128(itot, U128{fl, 0});
        fsum += fixfrac(z, pj);
    }
    add128(itot, U128{(fsum + FIX_HALF) >> 60, 0});
    for (u32 k = 0; k < L; k++) {
        const Mod &qk = dc->mod[k];
        U128 acc = mul128(pin[(size_t)k * N], dc->tPinv_modq[k]);
        for (u32 j = 0; j < Lp; j++) mac128(acc, yp[j], dc->tQF_modq[j][k]);
        add128(acc, itot);
        pout[(size_t)k * N] = reduce128(acc, qk);
    }
}
void launch_scale_round(const DevConsts *dc, u32 N, u32 L, const u64 *d, u32 nb, u64 *out01, size_t stride01, u64 *out2,
                        size_t stride2, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, 3, nb);
    hipLaunchKernelGGL(scale_round_kernel, grid, dim3(TPB), 0, st, dc, N, L, d, out01, stride01, out2, stride2);
}

// ---------------------------------------------------------------------------------------------
// BV relinearisation (row A7): digit decomposition and key-switch accumulation
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) digits_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ d2,
                                                     size_t stride2, u64 *__restrict__ dig)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 i = blockIdx.y / L, j = blockIdx.y % L, bin = blockIdx.z;
    const u64 v = d2[(size_t)bin * stride2 + (size_t)i * N + n];
    const Mod &mj = dc->mod[j];
    const u64 qi = dc->mod[i].q;
    u64 r = barrett128(0, v, mj);
    if (v > qi / 2) r = submod(r, dc->qi_modqj[i][j], mj.q);  // centred lift of the residue mod q_i
    dig[(((size_t)bin * L + i) * L + j) * N + n] = r;
}
void launch_digits(const DevConsts *dc, u32 N, u32 L, const u64 *d2, size_t stride2, u32 nb, u64 *dig, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, L * L, nb);
    hipLaunchKernelGGL(digits_kernel, grid, dim3(TPB), 0, st, dc, N, L, d2, stride2, dig);
}

__global__ void __launch_bounds__(TPB) relin_mac_kernel(const DevConsts *dc, u32 N, u32 L, const u64 *__restrict__ d01,
                                                        size_t stride01, const u64 *__restrict__ dig,
                                                        const u64 *__restrict__ key, const u64 *__restrict__ mask,
                                                        u64 *__restrict__ out, const u32 *__restrict__ out_map)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 c = blockIdx.y / L, j = blockIdx.y % L, bin = blockIdx.z;
    const Mod m = dc->mod[j];
    const size_t LN = (size_t)L * N;
    U128 acc = {0, 0};
    for (u32 i = 0; i < L; i++)
        mac128(acc, dig[(((size_t)bin * L + i) * L + j) * N + n], key[(((size_t)i * 2 + c) * L + j) * N + n]);
    u64 r = addmod(reduce128(acc, m), d01[(size_t)bin * stride01 + (size_t)c * LN + (size_t)j * N + n], m.q);
    if (mask) r = mulmod(r, mask[(size_t)bin * LN + (size_t)j * N + n], m);
    out[((size_t)bin * 2 + c) * LN + (size_t)j * N + (out_map ? out_map[n] : n)] = r;
}
void launch_relin_mac(const DevConsts *dc, u32 N, u32 L, const u64 *d01, size_t stride01, const u64 *dig, const u64 *key,
                      const u64 *mask, u64 *out, u32 nb, hipStream_t st, const u32 *out_map)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, nb);
    hipLaunchKernelGGL(relin_mac_kernel, grid, dim3(TPB), 0, st, dc, N, L, d01, stride01, dig, key, mask, out, out_map);
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
                                                           u64 *__restrict__ out)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y % L;
    const size_t o = ((size_t)blockIdx.z * 2 * L + blockIdx.y) * N + n;
    out[o] = mulmod(x[o], pt[(size_t)blockIdx.z * pt_stride + (size_t)l * N + n], dc->mod[l]);
}
void launch_ct_mul_plain(const DevConsts *dc, u32 N, u32 L, const u64 *x, const u64 *pt, size_t pt_stride, u64 *out,
                         u32 nct, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, 2 * L, nct);
    hipLaunchKernelGGL(ct_mul_plain_kernel, grid, dim3(TPB), 0, st, dc, N, L, x, pt, pt_stride, out);
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
