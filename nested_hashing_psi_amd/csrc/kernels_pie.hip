// kernels_pie.hip -- the coefficient-wise kernels of BatchedFHEHIPPIE::run() for gfx950:
// fused ct x pt multiply-accumulate (stage A), HPS base conversions, tensor product, BV digit
// decomposition and key-switch accumulation, mask multiply, automorphism permutation, packed
// encoding.  All are streaming u64 modular arithmetic: one thread per coefficient, consecutive
// lanes on consecutive coefficients (coalesced 8-byte or 16-byte lanes), constants scalar-loaded
// from one DevConsts block.  Reference call sites: BatchedFHEHIPPIE.cpp:101-127.
#include "kernels.hpp"

namespace piehip {

static const u32 TPB = 256;

// ---------------------------------------------------------------------------------------------
// Stage A (rows A3+A4): acc[beta][h][c][l][n] = sum_j idx[h][j][c][l][n] * db[h][beta][j][l][n] + minus[c][l][n]
// 128-bit lazy accumulation, one Barrett reduction per 32 terms.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) stage_a_kernel(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E,
                                                      const u64 *__restrict__ idx, const u64 *__restrict__ minus,
                                                      const u64 *__restrict__ db, u64 *__restrict__ acc)
{
    const u32 n = blockIdx.x * TPB + threadIdx.x;
    const u32 l = blockIdx.y;
    const u32 beta = blockIdx.z / K, h = blockIdx.z % K;
    if (n >= N) return;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N;
    const u64 *pi = idx + ((size_t)h * E) * 2 * LN + (size_t)l * N + n;
    const u64 *pd = db + (((size_t)h * b + beta) * E) * LN + (size_t)l * N + n;
    U128 a0 = {0, 0}, a1 = {0, 0};
    for (u32 j = 0; j < E; j++) {
        const u64 d = pd[(size_t)j * LN];
        const u64 i0 = pi[(size_t)j * 2 * LN];
        const u64 i1 = pi[(size_t)j * 2 * LN + LN];
        mac128(a0, i0, d);
        mac128(a1, i1, d);
        if ((j & 31) == 31) {
            a0.lo = reduce128(a0, m);
            a0.hi = 0;
            a1.lo = reduce128(a1, m);
            a1.hi = 0;
        }
    }
    u64 *po = acc + (((size_t)beta * K + h) * 2) * LN + (size_t)l * N + n;
    po[0] = addmod(reduce128(a0, m), minus[(size_t)l * N + n], m.q);
    po[LN] = addmod(reduce128(a1, m), minus[LN + (size_t)l * N + n], m.q);
}

void launch_stage_a(const DevConsts *dc, u32 N, u32 L, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus,
                    const u64 *db, u64 *acc, hipStream_t st)
{
    dim3 grid((N + TPB - 1) / TPB, L, b * K);
    hipLaunchKernelGGL(stage_a_kernel, grid, dim3(TPB), 0, st, dc, N, L, K, b, E, idx, minus, db, acc);
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
    const u64 s = reduce128(acc, tm);
    return submod(s, mulmod(v, prodmod, tm), tm.q);
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
        const u64 r = addmod(reduce128(acc, pj), reduce128(itot, pj), pj.q);
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
        pout[(size_t)k * N] = addmod(reduce128(acc, qk), reduce128(itot, qk), qk.q);
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
