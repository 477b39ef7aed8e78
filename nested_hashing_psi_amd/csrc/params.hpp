// params.hpp -- host-side BFV-RNS parameter tables for the PIE hot path.
//
// Everything OpenFHE's CryptoContext would hold for the reference's calls at
// BatchedFHEHIPPIE.cpp:108-126 (reference src/Client/FHE/BatchedFHEPSIClient.cpp:72-78 creates it):
// the RNS prime chain Q, the auxiliary basis P of the HPS multiplication, NTT twiddle tables,
// the packed-encoding slot map and the CRT constants of the base conversions.
// Conventions follow SURVEY.md appendix A (all [OFHE-UNVERIFIED] there).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "modarith.h"

namespace piehip {

static const u32 MAX_L = 7;            // RNS limbs in Q
static const u32 MAX_M = 2 * MAX_L + 1;  // limbs in QP

// CRT constants, laid out for direct upload (scalar-loaded by the kernels).
// Index conventions: i over Q (0..L-1), j over P (0..L); global limb ids Q: 0..L-1, P: L..2L;
// mod[M] is the plaintext modulus t.
struct DevConsts {
    Mod mod[MAX_M + 1];
    u64 qhat_inv[8];          // [(Q/q_i)^-1]_{q_i}
    u64 qhat_inv_sh[8];
    u64 qhat_modp[8][8];      // [i][j]  [Q/q_i]_{p_j}
    u64 Q_modp[8];            // [Q]_{p_j}
    u64 P_modq[8];            // w_i = [P]_{q_i}
    u64 P_modq_sh[8];
    u64 PI_modp[8][8];        // [i][j]  [floor(P/q_i)]_{p_j}
    u64 phat_inv[8];          // [(P/p_j)^-1]_{p_j}
    u64 phat_inv_sh[8];
    u64 phat_modq[8][8];      // [j][i]  [P/p_j]_{q_i}
    u64 qp_hat_inv[16];       // [(QP/m)^-1]_m   (only the P entries are used)
    u64 qp_hat_inv_sh[16];
    u64 tPinv_modq[8];        // [t P^-1]_{q_k}
    u64 tQ_modp[8];           // [tQ]_{p_j}
    u64 tQ_modp_sh[8];
    u64 tQF_modq[8][8];       // [j][k]  [floor(tQ/p_j)]_{q_k}
    u64 qi_modqj[8][8];       // [i][j]  q_i mod q_j  (centred digit lift of the BV key switch)
    // outermost NTT stage folded into the neighbouring coefficient-wise kernels (kernels_pie.hip):
    u64 fold_w[MAX_M + 1], fold_w_sh[MAX_M + 1];    // forward: psi^{N/2} (twiddle of the stage with one group)
    u64 fold_ia[MAX_M + 1], fold_ia_sh[MAX_M + 1];  // inverse: N^-1            (sum branch)
    u64 fold_ib[MAX_M + 1], fold_ib_sh[MAX_M + 1];  // inverse: psi^{-N/2} N^-1 (difference branch)
    // the same two constants times (Q/q_i)^-1: the CRT digit y_i of a base conversion straight from the folded load
    u64 fold_iaq[8], fold_iaq_sh[8], fold_ibq[8], fold_ibq_sh[8];
    // ... times (QP/p_j)^-1 for the P limbs and times t P^-1 for the Q limbs: what scale-and-round multiplies its folded inputs by
    // first anyway (scale_round_core: yp and the own-limb term)
    u64 fold_iap[8], fold_iap_sh[8], fold_ibp[8], fold_ibp_sh[8];
    u64 fold_iat[8], fold_iat_sh[8], fold_ibt[8], fold_ibt_sh[8];
    // client harness (encryption / decryption): t^-1 mod q_i, [Q]_t, Shoup companion of t mod q_i
    u64 t_inv_modq[8];
    u64 t_modq_sh[8];
    u64 Q_modt;
    u32 N, logN, L, M;
};

struct HostParams {
    u32 N = 0, logN = 0, L = 0, M = 0;
    u64 t = 0;
    std::vector<u64> moduli;          // q_0..q_{L-1}, p_0..p_L, t
    std::vector<u64> psi;             // smallest primitive 2N-th roots, one per modulus
    // per modulus, N entries each: tw[k] = psi^{bitrev(k)}, itw[k] = psi^{-bitrev(k)}, + Shoup companions
    std::vector<std::vector<u64>> tw, tw_sh, itw, itw_sh;
    std::vector<u32> slot_pos;        // packed-encoding slot -> EVALUATION position
    DevConsts dc;

    // returns empty string on success, else the error text
    std::string init(u32 N, u32 L, u64 t, const u64 *q, const u64 *p);
    // EVALUATION-domain index map of the automorphism X -> X^g: out[p] = in[map[p]]
    std::vector<u32> automorph_map(u32 g) const;
};

bool is_prime_u64(u64 n);
// `count` primes = 1 (mod 2N), descending, strictly below `below`
bool prime_chain(u32 N, u64 below, u32 count, u64 *out);
u64 min_primitive_root(u64 q, u32 N);
u64 powmod(u64 a, u64 e, u64 q);
u64 invmod(u64 a, u64 q);
u32 bitrev32(u32 x, u32 bits);

}  // namespace piehip
