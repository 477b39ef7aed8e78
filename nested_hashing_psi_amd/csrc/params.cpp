// params.cpp -- host-side parameter generation (see params.hpp).
#include "params.hpp"

#include <string.h>

namespace piehip {

typedef unsigned __int128 u128;

static inline u64 mm(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

u64 powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mm(r, a, q);
        a = mm(a, a, q);
    }
    return r;
}
u64 invmod(u64 a, u64 q) { return powmod(a % q, q - 2, q); }

u32 bitrev32(u32 x, u32 bits)
{
    u32 r = 0;
    for (u32 i = 0; i < bits; i++, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}

bool is_prime_u64(u64 n)
{
    // deterministic Miller-Rabin for 64-bit integers
    static const u64 B[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (u64 b : B) {
        if (n == b) return true;
        if (n % b == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    for (; !(d & 1); d >>= 1) s++;
    for (u64 b : B) {
        u64 x = powmod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; r++) {
            x = mm(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

bool prime_chain(u32 N, u64 below, u32 count, u64 *out)
{
    const u64 step = 2ULL * N;
    if (below <= step + 1) return false;
    u64 c = ((below - 2) / step) * step + 1;
    u32 got = 0;
    for (; got < count && c > step; c -= step)
        if (is_prime_u64(c)) out[got++] = c;
    return got == count;
}

u64 min_primitive_root(u64 q, u32 N)
{
    const u64 m = 2ULL * N;
    if ((q - 1) % m) return 0;
    u64 root = 0;
    for (u64 g = 2; g < 1000 && !root; g++) {
        u64 x = powmod(g, (q - 1) / m, q);
        if (powmod(x, N, q) == q - 1) root = x;
    }
    if (!root) return 0;
    // every primitive 2N-th root is an odd power of `root`; keep the smallest
    const u64 sq = mm(root, root, q);
    u64 cur = root, best = root;
    for (u32 k = 1; k < N; k++) {
        cur = mm(cur, sq, q);
        if (cur < best) best = cur;
    }
    return best;
}

static inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

static Mod make_mod(u64 q, u32 N)
{
    Mod m;
    memset(&m, 0, sizeof(m));
    m.q = q;
    u128 ratio = (~(u128)0) / q;
    m.r0 = (u64)ratio;
    m.r1 = (u64)(ratio >> 64);
    m.fshift = (u32)__builtin_clzll(q);
    m.fconst = (u64)((((u128)1) << (127 - m.fshift)) / q);
    m.n_inv = invmod(N, q);
    m.n_inv_sh = shoup(m.n_inv, q);
    return m;
}

// product of a list of moduli, skipping index `skip` (or none if skip < 0), reduced mod q
static u64 prod_mod(const u64 *ms, u32 n, int skip, u64 q)
{
    u64 r = 1 % q;
    for (u32 k = 0; k < n; k++)
        if ((int)k != skip) r = mm(r, ms[k] % q, q);
    return r;
}

std::string HostParams::init(u32 N_, u32 L_, u64 t_, const u64 *q, const u64 *p)
{
    if (N_ < 8 || (N_ & (N_ - 1)) || N_ > (1u << 16)) return "N must be a power of two in [8, 65536]";
    if (L_ < 1 || L_ > MAX_L) return "L must be in [1, 7]";
    if ((t_ - 1) % (2ULL * N_) || !is_prime_u64(t_)) return "plaintext modulus t must be prime and = 1 (mod 2N)";
    N = N_;
    L = L_;
    M = 2 * L + 1;
    t = t_;
    logN = 0;
    while ((1u << logN) < N) logN++;
    moduli.assign(M + 1, 0);
    if (q && p) {
        for (u32 i = 0; i < L; i++) moduli[i] = q[i];
        for (u32 j = 0; j <= L; j++) moduli[L + j] = p[j];
    } else if (!q && !p) {
        if (!prime_chain(N, 1ULL << 60, M, moduli.data())) return "prime chain exhausted";
    } else {
        return "pass both q and p, or neither";
    }
    moduli[M] = t;
    for (u32 a = 0; a < M; a++) {
        const u64 m = moduli[a];
        if (m >> 61) return "RNS primes must be < 2^61";
        if (m <= t) return "RNS primes must exceed the plaintext modulus";
        if ((m - 1) % (2ULL * N) || !is_prime_u64(m)) return "RNS moduli must be prime and = 1 (mod 2N)";
        for (u32 b = 0; b < a; b++)
            if (moduli[b] == m) return "RNS moduli must be distinct";
    }

    psi.assign(M + 1, 0);
    tw.assign(M + 1, {});
    tw_sh.assign(M + 1, {});
    itw.assign(M + 1, {});
    itw_sh.assign(M + 1, {});
    memset(&dc, 0, sizeof(dc));
    dc.N = N;
    dc.logN = logN;
    dc.L = L;
    dc.M = M;
    for (u32 a = 0; a <= M; a++) {
        const u64 m = moduli[a];
        dc.mod[a] = make_mod(m, N);
        psi[a] = min_primitive_root(m, N);
        if (!psi[a]) return "no primitive 2N-th root";
        const u64 ipsi = invmod(psi[a], m);
        tw[a].assign(N, 0);
        itw[a].assign(N, 0);
        tw_sh[a].assign(N, 0);
        itw_sh[a].assign(N, 0);
        u64 pw = 1, ipw = 1;
        for (u32 e = 0; e < N; e++) {
            const u32 k = bitrev32(e, logN);
            tw[a][k] = pw;
            itw[a][k] = ipw;
            tw_sh[a][k] = shoup(pw, m);
            itw_sh[a][k] = shoup(ipw, m);
            pw = mm(pw, psi[a], m);
            ipw = mm(ipw, ipsi, m);
        }
        dc.fold_w[a] = tw[a][1];
        dc.fold_w_sh[a] = tw_sh[a][1];
        dc.fold_ia[a] = dc.mod[a].n_inv;
        dc.fold_ia_sh[a] = dc.mod[a].n_inv_sh;
        dc.fold_ib[a] = mm(itw[a][1], dc.mod[a].n_inv, m);
        dc.fold_ib_sh[a] = shoup(dc.fold_ib[a], m);
    }

    // slot i <-> evaluation point psi^(5^i), slot N/2+i <-> psi^(-5^i); EVALUATION position p
    // holds a(psi^(2 bitrev(p) + 1))
    slot_pos.assign(N, 0);
    {
        const u64 m2 = 2ULL * N;
        u64 e = 1;
        for (u32 i = 0; i < N / 2; i++) {
            slot_pos[i] = bitrev32((u32)((e - 1) / 2), logN);
            slot_pos[N / 2 + i] = bitrev32((u32)((m2 - e - 1) / 2), logN);
            e = (e * 5) % m2;
        }
    }

    const u64 *Q = moduli.data();
    const u64 *P = moduli.data() + L;
    const u32 Lp = L + 1;
    for (u32 i = 0; i < L; i++) {
        const u64 qi = Q[i];
        dc.qhat_inv[i] = invmod(prod_mod(Q, L, (int)i, qi), qi);
        dc.qhat_inv_sh[i] = shoup(dc.qhat_inv[i], qi);
        dc.P_modq[i] = prod_mod(P, Lp, -1, qi);
        dc.P_modq_sh[i] = shoup(dc.P_modq[i], qi);
        dc.tPinv_modq[i] = mm(t % qi, invmod(dc.P_modq[i], qi), qi);
        for (u32 j = 0; j < Lp; j++) {
            const u64 pj = P[j];
            dc.qhat_modp[i][j] = prod_mod(Q, L, (int)i, pj);
            // floor(P/q_i) = (P - (P mod q_i)) / q_i, and P = 0 (mod p_j)
            const u64 v = mm(dc.P_modq[i] % pj, invmod(qi, pj), pj);
            dc.PI_modp[i][j] = v ? pj - v : 0;
        }
        for (u32 j = 0; j < L; j++) dc.qi_modqj[i][j] = qi % Q[j];
        dc.t_inv_modq[i] = invmod(t % qi, qi);
        dc.fold_iaq[i] = mm(dc.fold_ia[i], dc.qhat_inv[i], qi);
        dc.fold_iaq_sh[i] = shoup(dc.fold_iaq[i], qi);
        dc.fold_ibq[i] = mm(dc.fold_ib[i], dc.qhat_inv[i], qi);
        dc.fold_ibq_sh[i] = shoup(dc.fold_ibq[i], qi);
        dc.t_modq_sh[i] = shoup(t % qi, qi);
    }
    for (u32 j = 0; j < Lp; j++) {
        const u64 pj = P[j];
        dc.Q_modp[j] = prod_mod(Q, L, -1, pj);
        dc.phat_inv[j] = invmod(prod_mod(P, Lp, (int)j, pj), pj);
        dc.phat_inv_sh[j] = shoup(dc.phat_inv[j], pj);
        dc.tQ_modp[j] = mm(t % pj, dc.Q_modp[j], pj);
        dc.tQ_modp_sh[j] = shoup(dc.tQ_modp[j], pj);
        for (u32 i = 0; i < L; i++) {
            const u64 qi = Q[i];
            dc.phat_modq[j][i] = prod_mod(P, Lp, (int)j, qi);
            // floor(tQ/p_j) = (tQ - (tQ mod p_j)) / p_j, and tQ = 0 (mod q_i)
            const u64 v = mm(dc.tQ_modp[j] % qi, invmod(pj, qi), qi);
            dc.tQF_modq[j][i] = v ? qi - v : 0;
        }
    }
    dc.Q_modt = prod_mod(Q, L, -1, t);
    for (u32 a = 0; a < M; a++) {
        dc.qp_hat_inv[a] = invmod(prod_mod(moduli.data(), M, (int)a, moduli[a]), moduli[a]);
        dc.qp_hat_inv_sh[a] = shoup(dc.qp_hat_inv[a], moduli[a]);
    }
    for (u32 j = 0; j < Lp; j++) {
        const u64 pj = P[j];
        dc.fold_iap[j] = mm(dc.fold_ia[L + j], dc.qp_hat_inv[L + j], pj);
        dc.fold_iap_sh[j] = shoup(dc.fold_iap[j], pj);
        dc.fold_ibp[j] = mm(dc.fold_ib[L + j], dc.qp_hat_inv[L + j], pj);
        dc.fold_ibp_sh[j] = shoup(dc.fold_ibp[j], pj);
    }
    for (u32 k = 0; k < L; k++) {
        const u64 qk = Q[k];
        dc.fold_iat[k] = mm(dc.fold_ia[k], dc.tPinv_modq[k], qk);
        dc.fold_iat_sh[k] = shoup(dc.fold_iat[k], qk);
        dc.fold_ibt[k] = mm(dc.fold_ib[k], dc.tPinv_modq[k], qk);
        dc.fold_ibt_sh[k] = shoup(dc.fold_ibt[k], qk);
    }
    return "";
}

std::vector<u32> HostParams::automorph_map(u32 g) const
{
    std::vector<u32> map(N);
    const u64 m2 = 2ULL * N;
    for (u32 p = 0; p < N; p++) {
        const u64 e = ((2ULL * bitrev32(p, logN) + 1) * g) % m2;
        map[p] = bitrev32((u32)((e - 1) / 2), logN);
    }
    return map;
}

}  // namespace piehip
