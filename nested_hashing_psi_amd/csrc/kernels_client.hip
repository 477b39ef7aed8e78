// kernels_client.hip -- client-side BFV operations of the harness: secret-key encryption and decryption.
//
// Harness, not the hot path: the reference client (src/Client/FHE/BatchedFHEPSIClient.cpp) encrypts the index
// matrix and the minus vector with the secret key (:155-156,161-168) and decrypts the b result ciphertexts
// (:249-265).  Sampling happens on the host (client.cpp, deterministic streams); the polynomial arithmetic
// runs here so that the end-to-end measurement does not time a CPU NTT.
#include "kernels.hpp"

namespace piehip {

static const u32 CTPB = 256;

// em[ct][l][n] = e[ct][n] (small signed) + round(Q m / t) term, COEFFICIENT format, to be transformed
//   round(Q m / t) = (Q m - [Q m]_t) / t  with [.]_t centred;  mod q_i: -[Q m]_t t^-1
__global__ void __launch_bounds__(CTPB) enc_message_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 M,
                                                           const u64 *__restrict__ coeff_t, const int32_t *__restrict__ e,
                                                           u64 *__restrict__ em)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y, ct = blockIdx.z;
    const Mod mt = dc->mod[M];
    const Mod m = dc->mod[l];
    const u64 t = mt.q;
    const u64 rr = mulmod(coeff_t[(size_t)ct * N + n], dc->Q_modt, mt);
    u64 term = rr > t / 2 ? mulmod(t - rr, dc->t_inv_modq[l], m) : mulmod(rr, dc->t_inv_modq[l], m);
    if (rr <= t / 2) term = term ? m.q - term : 0;
    const int32_t ev = e[(size_t)ct * N + n];
    const u64 el = ev >= 0 ? (u64)ev : m.q - (u64)(-ev);
    em[((size_t)ct * L + l) * N + n] = addmod(el, term, m.q);
}

// c0 = em - a s   (all EVALUATION); ct layout [ct][2][L][N] with c1 = a already in place
__global__ void __launch_bounds__(CTPB) enc_finish_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, const u64 *__restrict__ em,
                                                          const u64 *__restrict__ sk, const u64 *__restrict__ a_in,
                                                          u64 *__restrict__ out)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y, ct = blockIdx.z;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N, x = (size_t)l * N + n;
    // a_in: the uniform polynomials [nct][L][N] as uploaded; null: they already sit in the c1 halves of out
    const u64 a = a_in ? a_in[(size_t)ct * LN + x] : out[(size_t)ct * 2 * LN + LN + x];
    if (a_in) out[(size_t)ct * 2 * LN + LN + x] = a;
    out[(size_t)ct * 2 * LN + x] = submod(em[(size_t)ct * LN + x], mulmod(a, sk[x], m), m.q);
}

// key-switch key rows: b = e - a s (+ s_from on limb == digit); ks layout [digit][2][L][N], a already in place
__global__ void __launch_bounds__(CTPB) ks_finish_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, const u64 *__restrict__ e,
                                                         const u64 *__restrict__ sk, const u64 *__restrict__ s_from,
                                                         u64 *__restrict__ ks)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y, dgt = blockIdx.z;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N, x = (size_t)l * N + n;
    const u64 a = ks[((size_t)dgt * 2 + 1) * LN + x];
    u64 v = submod(e[(size_t)dgt * LN + x], mulmod(a, sk[x], m), m.q);
    if (l == dgt) v = addmod(v, s_from[x], m.q);
    ks[((size_t)dgt * 2 + 0) * LN + x] = v;
}

__global__ void __launch_bounds__(CTPB) square_kernel(const DevConsts *__restrict__ dc, u32 N, const u64 *__restrict__ s,
                                                      u64 *__restrict__ s2)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const size_t x = (size_t)blockIdx.y * N + n;
    s2[x] = mulmod(s[x], s[x], dc->mod[blockIdx.y]);
}

// x = c0 + c1 s (EVALUATION) for nct ciphertexts -> xs[ct][L][N]
__global__ void __launch_bounds__(CTPB) dec_dot_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, const u64 *__restrict__ ct,
                                                       const u64 *__restrict__ sk, u64 *__restrict__ xs)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const u32 l = blockIdx.y, c = blockIdx.z;
    const Mod m = dc->mod[l];
    const size_t LN = (size_t)L * N, x = (size_t)l * N + n;
    xs[(size_t)c * LN + x] = addmod(ct[(size_t)c * 2 * LN + x], mulmod(ct[(size_t)c * 2 * LN + LN + x], sk[x], m), m.q);
}

// COEFFICIENT x (mod Q) -> m = round(t x / Q) mod t   (HPS scale-and-round, 60-bit fixed-point rounding term)
__global__ void __launch_bounds__(CTPB) dec_round_kernel(const DevConsts *__restrict__ dc, u32 N, u32 L, u32 M,
                                                         const u64 *__restrict__ xs, u64 *__restrict__ coeff_t)
{
    const u32 n = blockIdx.x * CTPB + threadIdx.x;
    if (n >= N) return;
    const u32 c = blockIdx.y;
    const Mod mt = dc->mod[M];
    const u64 t = mt.q;
    u64 acc = 0, fsum = 0;
    for (u32 i = 0; i < L; i++) {
        const Mod &mi = dc->mod[i];
        const u64 y = mul_shoup(xs[((size_t)c * L + i) * N + n], dc->qhat_inv[i], dc->qhat_inv_sh[i], mi.q);
        u64 fl, z;
        divmod_shoup(y, t, dc->t_modq_sh[i], mi.q, fl, z);  // floor(t y / q_i) < t, remainder
        acc = addmod(acc, fl, t);
        fsum += fixfrac(z, mi);
    }
    const u64 rnd = (fsum + FIX_HALF) >> 60;
    u64 r = acc + rnd;  // rnd <= L
    while (r >= t) r -= t;
    coeff_t[(size_t)c * N + n] = r;
}

// slots[c][i] = centred value at EVALUATION position slot_pos[i]
__global__ void __launch_bounds__(CTPB) decode_gather_kernel(const DevConsts *__restrict__ dc, u32 N, u32 M, const u64 *__restrict__ u,
                                                             const u32 *__restrict__ slot_pos, u32 B, int64_t *__restrict__ slots)
{
    const u32 i = blockIdx.x * CTPB + threadIdx.x;
    if (i >= B) return;
    const u64 t = dc->mod[M].q;
    const u64 v = u[(size_t)blockIdx.y * N + slot_pos[i]];
    slots[(size_t)blockIdx.y * B + i] = v > t / 2 ? -(int64_t)(t - v) : (int64_t)v;
}

void launch_enc_message(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *coeff_t, const int32_t *e, u64 *em, u32 nct, hipStream_t st)
{
    hipLaunchKernelGGL(enc_message_kernel, dim3((N + CTPB - 1) / CTPB, L, nct), dim3(CTPB), 0, st, dc, N, L, M, coeff_t, e, em);
}
void launch_enc_finish(const DevConsts *dc, u32 N, u32 L, const u64 *em, const u64 *sk, const u64 *a_in, u64 *out, u32 nct,
                       hipStream_t st)
{
    hipLaunchKernelGGL(enc_finish_kernel, dim3((N + CTPB - 1) / CTPB, L, nct), dim3(CTPB), 0, st, dc, N, L, em, sk, a_in, out);
}
void launch_ks_finish(const DevConsts *dc, u32 N, u32 L, const u64 *e, const u64 *sk, const u64 *s_from, u64 *ks, hipStream_t st)
{
    hipLaunchKernelGGL(ks_finish_kernel, dim3((N + CTPB - 1) / CTPB, L, L), dim3(CTPB), 0, st, dc, N, L, e, sk, s_from, ks);
}
void launch_square(const DevConsts *dc, u32 N, u32 L, const u64 *s, u64 *s2, hipStream_t st)
{
    hipLaunchKernelGGL(square_kernel, dim3((N + CTPB - 1) / CTPB, L), dim3(CTPB), 0, st, dc, N, s, s2);
}
void launch_dec_dot(const DevConsts *dc, u32 N, u32 L, const u64 *ct, const u64 *sk, u64 *xs, u32 nct, hipStream_t st)
{
    hipLaunchKernelGGL(dec_dot_kernel, dim3((N + CTPB - 1) / CTPB, L, nct), dim3(CTPB), 0, st, dc, N, L, ct, sk, xs);
}
void launch_dec_round(const DevConsts *dc, u32 N, u32 L, u32 M, const u64 *xs, u64 *coeff_t, u32 nct, hipStream_t st)
{
    hipLaunchKernelGGL(dec_round_kernel, dim3((N + CTPB - 1) / CTPB, nct), dim3(CTPB), 0, st, dc, N, L, M, xs, coeff_t);
}
void launch_decode_gather(const DevConsts *dc, u32 N, u32 M, const u64 *u, const u32 *slot_pos, u32 B, int64_t *slots, u32 nct,
                          hipStream_t st)
{
    hipLaunchKernelGGL(decode_gather_kernel, dim3((B + CTPB - 1) / CTPB, nct), dim3(CTPB), 0, st, dc, N, M, u, slot_pos, B, slots);
}

}  // namespace piehip
