/*
 * pie_oracle.c -- CPU restatement of the batched-FHE PIE hot path.  TEST INFRASTRUCTURE ONLY
 * (see pie_oracle.h for the scope statement, the reference citations and the parity status:
 * "parity unpinned" at ciphertext-bit level, pinned at decrypted-slot level).
 *
 * Everything is unsigned 64-bit modular arithmetic over primes < 2^62 with 128-bit
 * intermediates.  All arrays crossing the API are canonical residues in [0, modulus).
 */
#include "pie_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

/* ------------------------------------------------------------------------------------------
 * modular arithmetic
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    u64 q;
    u64 r0, r1;         /* floor(2^128 / q), low / high word (two-word Barrett) */
    u64 psi;            /* smallest primitive 2N-th root of unity */
    u64 n_inv, n_inv_sh;
    u64 *tw, *tw_sh;    /* tw[k] = psi^{bitrev(k)}, k in [1,N); Shoup companions floor(w 2^64 / q) */
    u64 *itw, *itw_sh;  /* itw[k] = psi^{-bitrev(k)} */
    int fshift;         /* clz(q) */
    u64 fconst;         /* floor(2^(127-fshift) / q): fixed-point reciprocal for the rounding terms */
} po_mod;

static inline u64 mulhi64(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

/* z < 2^128 -> z mod q (q < 2^62) */
static inline u64 barrett128(u128 z, const po_mod *m)
{
    u64 z0 = (u64)z, z1 = (u64)(z >> 64);
    u64 c = mulhi64(z0, m->r0);
    u128 t2 = (u128)z0 * m->r1;
    u128 t3 = (u128)z1 * m->r0;
    u128 mid = (u128)(u64)t2 + (u64)t3 + c;
    u64 qhat = z1 * m->r1 + (u64)(t2 >> 64) + (u64)(t3 >> 64) + (u64)(mid >> 64);
    u64 r = z0 - qhat * m->q;
    while (r >= m->q) r -= m->q;
    return r;
}
static inline u64 mulmod(u64 a, u64 b, const po_mod *m) { return barrett128((u128)a * b, m); }
static inline u64 addmod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return s >= q ? s - q : s;
}
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }
static inline u64 redmod(u64 a, const po_mod *m) { return barrett128((u128)a, m); }

static u64 powmod_raw(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = (u64)(((u128)r * a) % q);
        a = (u64)(((u128)a * a) % q);
        e >>= 1;
    }
    return r;
}
static u64 invmod_raw(u64 a, u64 q) { return powmod_raw(a, q - 2, q); } /* q prime */

static inline u64 shoup_of(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
/* a < 2^64 arbitrary, w < q: result in [0, 2q) */
static inline u64 mul_shoup_lazy(u64 a, u64 w, u64 wsh, u64 q) { return a * w - mulhi64(a, wsh) * q; }

/* fixed-point fraction y/q (y < q) with 60 fractional bits; error < 2^-59 */
static inline u64 fixfrac(u64 y, const po_mod *m) { return mulhi64(y << m->fshift, m->fconst) >> 3; }
#define FIX_ONE (1ULL << 60)
#define FIX_HALF (1ULL << 59)

int po_is_prime(u64 n)
{
    static const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (int i = 0; i < 12; i++) {
        if (n == bases[i]) return 1;
        if (n % bases[i] == 0) return 0;
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        s++;
    }
    for (int i = 0; i < 12; i++) {
        u64 x = powmod_raw(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            x = (u64)(((u128)x * x) % n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}

int po_gen_primes(u32 N, u64 below, u32 count, u64 *out)
{
    u64 step = 2ULL * N;
    if (below <= step + 1) return -1;
    /* largest candidate = 1 (mod 2N) strictly below `below` */
    u64 c = ((below - 2) / step) * step + 1;
    u32 got = 0;
    while (got < count && c > step) {
        if (po_is_prime(c)) out[got++] = c;
        c -= step;
    }
    return got == count ? 0 : -1;
}

u64 po_min_root(u64 q, u32 N)
{
    u64 m = 2ULL * N;
    if ((q - 1) % m) return 0;
    u64 e = (q - 1) / m;
    u64 root = 0;
    for (u64 g = 2; g < 1000; g++) {
        u64 x = powmod_raw(g, e, q);
        if (powmod_raw(x, N, q) == q - 1) {
            root = x;
            break;
        }
    }
    if (!root) return 0;
    /* all primitive 2N-th roots are root^k, k odd: take the smallest */
    u64 r2 = (u64)(((u128)root * root) % q);
    u64 cur = root, best = root;
    for (u32 k = 1; k < N; k++) {
        cur = (u64)(((u128)cur * r2) % q);
        if (cur < best) best = cur;
    }
    return best;
}

/* ------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------- */
struct po_ctx {
    u32 N, logN, L, M; /* M = 2L+1 moduli in QP; mod[M] is the plaintext modulus t */
    u64 t;
    po_mod *mod;
    u32 *slot_pos; /* slot i -> position in the EVALUATION array */
    /* CRT constants; index conventions: i over Q (0..L-1), j over P (0..L), global limb ids
     * Q: 0..L-1, P: L..2L */
    u64 *qhat_inv;    /* [L]      [(Q/q_i)^-1]_{q_i}                       */
    u64 *qhat_modp;   /* [L][L+1] [Q/q_i]_{p_j}                            */
    u64 *Q_modp;      /* [L+1]    [Q]_{p_j}                                */
    u64 *P_modq;      /* [L]      [P]_{q_i}  (= w_i of the P/Q scaling)    */
    u64 *PI_modp;     /* [L][L+1] [floor(P/q_i)]_{p_j}                     */
    u64 *phat_inv;    /* [L+1]    [(P/p_j)^-1]_{p_j}                       */
    u64 *phat_modq;   /* [L+1][L] [P/p_j]_{q_i}                            */
    u64 *Pfull_modq;  /* [L]      [P]_{q_i} (same as P_modq; kept for clarity of the P->Q expansion) */
    u64 *qp_hat_inv;  /* [M]      [(QP/m)^-1]_m                            */
    u64 *tPinv_modq;  /* [L]      [t P^-1]_{q_k}                           */
    u64 *tQ_modp;     /* [L+1]    [tQ]_{p_j}                               */
    u64 *tQF_modq;    /* [L+1][L] [floor(tQ/p_j)]_{q_k}                    */
    u64 *t_inv_modq;  /* [L]      [t^-1]_{q_i}                             */
    u64 Q_modt;       /*          [Q]_t                                    */
};

static u32 bitrev(u32 x, u32 bits)
{
    u32 r = 0;
    for (u32 i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

static int mod_init(po_mod *m, u64 q, u32 N, u32 logN)
{
    memset(m, 0, sizeof(*m));
    m->q = q;
    u128 ratio = (~(u128)0) / q; /* floor((2^128-1)/q) == floor(2^128/q) for q not a power of two */
    m->r0 = (u64)ratio;
    m->r1 = (u64)(ratio >> 64);
    m->fshift = __builtin_clzll(q);
    m->fconst = (u64)((((u128)1) << (127 - m->fshift)) / q);
    m->psi = po_min_root(q, N);
    if (!m->psi) return -1;
    u64 psi_inv = invmod_raw(m->psi, q);
    m->n_inv = invmod_raw(N, q);
    m->n_inv_sh = shoup_of(m->n_inv, q);
    m->tw = (u64 *)malloc(sizeof(u64) * N * 4);
    if (!m->tw) return -1;
    m->tw_sh = m->tw + N;
    m->itw = m->tw + 2 * N;
    m->itw_sh = m->tw + 3 * N;
    u64 pw = 1, ipw = 1;
    for (u32 e = 0; e < N; e++) { /* psi^e goes to index bitrev(e) */
        u32 k = bitrev(e, logN);
        m->tw[k] = pw;
        m->itw[k] = ipw;
        pw = (u64)(((u128)pw * m->psi) % q);
        ipw = (u64)(((u128)ipw * psi_inv) % q);
    }
    for (u32 k = 0; k < N; k++) {
        m->tw_sh[k] = shoup_of(m->tw[k], q);
        m->itw_sh[k] = shoup_of(m->itw[k], q);
    }
    return 0;
}

po_ctx *po_create(u32 N, u32 L, u64 t, const u64 *q, const u64 *p)
{
    if (N < 8 || (N & (N - 1)) || L < 1 || L > 7) return NULL;
    if ((t - 1) % (2ULL * N) || !po_is_prime(t)) return NULL;
    po_ctx *c = (po_ctx *)calloc(1, sizeof(po_ctx));
    c->N = N;
    c->L = L;
    c->M = 2 * L + 1;
    c->t = t;
    while ((1u << c->logN) < N) c->logN++;
    u32 M = c->M;
    u64 *chain = (u64 *)malloc(sizeof(u64) * (M + 1));
    if (q && p) {
        memcpy(chain, q, sizeof(u64) * L);
        memcpy(chain + L, p, sizeof(u64) * (L + 1));
    } else {
        if (po_gen_primes(N, 1ULL << 60, M, chain)) goto fail;
    }
    chain[M] = t;
    for (u32 a = 0; a < M; a++) {
        if (chain[a] >> 62 || !po_is_prime(chain[a]) || (chain[a] - 1) % (2ULL * N) || chain[a] == t) goto fail;
        for (u32 b2 = 0; b2 < a; b2++)
            if (chain[a] == chain[b2]) goto fail;
    }
    c->mod = (po_mod *)calloc(M + 1, sizeof(po_mod));
    for (u32 a = 0; a <= M; a++)
        if (mod_init(&c->mod[a], chain[a], N, c->logN)) goto fail;
    free(chain);
    chain = NULL;

    /* slot permutation: slot i <-> evaluation point psi^{5^i}, slot N/2+i <-> psi^{-5^i};
     * EVALUATION position p holds a(psi^{2 bitrev(p) + 1}) */
    c->slot_pos = (u32 *)malloc(sizeof(u32) * N);
    {
        u64 m2 = 2ULL * N, e = 1;
        for (u32 i = 0; i < N / 2; i++) {
            c->slot_pos[i] = bitrev((u32)((e - 1) / 2), c->logN);
            c->slot_pos[N / 2 + i] = bitrev((u32)((m2 - e - 1) / 2), c->logN);
            e = (e * 5) % m2;
        }
    }

    u32 Lp = L + 1;
    const po_mod *md = c->mod;
#define ALLOC(n) ((u64 *)calloc((n), sizeof(u64)))
    c->qhat_inv = ALLOC(L);
    c->qhat_modp = ALLOC(L * Lp);
    c->Q_modp = ALLOC(Lp);
    c->P_modq = ALLOC(L);
    c->PI_modp = ALLOC(L * Lp);
    c->phat_inv = ALLOC(Lp);
    c->phat_modq = ALLOC(Lp * L);
    c->Pfull_modq = ALLOC(L);
    c->qp_hat_inv = ALLOC(M);
    c->tPinv_modq = ALLOC(L);
    c->tQ_modp = ALLOC(Lp);
    c->tQF_modq = ALLOC(Lp * L);
    c->t_inv_modq = ALLOC(L);
#undef ALLOC
    for (u32 i = 0; i < L; i++) {
        u64 qi = md[i].q;
        u64 h = 1;
        for (u32 k = 0; k < L; k++)
            if (k != i) h = mulmod(h, md[k].q % qi, &md[i]);
        c->qhat_inv[i] = invmod_raw(h, qi);
        u64 pm = 1;
        for (u32 j = 0; j < Lp; j++) pm = mulmod(pm, md[L + j].q % qi, &md[i]);
        c->P_modq[i] = pm;
        c->Pfull_modq[i] = pm;
        c->tPinv_modq[i] = mulmod(t % qi, invmod_raw(pm, qi), &md[i]);
        c->t_inv_modq[i] = invmod_raw(t % qi, qi);
        for (u32 j = 0; j < Lp; j++) {
            const po_mod *pj = &md[L + j];
            u64 hh = 1;
            for (u32 k = 0; k < L; k++)
                if (k != i) hh = mulmod(hh, md[k].q % pj->q, pj);
            c->qhat_modp[i * Lp + j] = hh;
            /* floor(P/q_i) = (P - (P mod q_i)) / q_i  and  P = 0 (mod p_j) */
            u64 v = mulmod(pm % pj->q, invmod_raw(qi % pj->q, pj->q), pj);
            c->PI_modp[i * Lp + j] = negmod(v, pj->q);
        }
    }
    for (u32 j = 0; j < Lp; j++) {
        const po_mod *pj = &md[L + j];
        u64 Qm = 1;
        for (u32 k = 0; k < L; k++) Qm = mulmod(Qm, md[k].q % pj->q, pj);
        c->Q_modp[j] = Qm;
        u64 h = 1;
        for (u32 k = 0; k < Lp; k++)
            if (k != j) h = mulmod(h, md[L + k].q % pj->q, pj);
        c->phat_inv[j] = invmod_raw(h, pj->q);
        c->tQ_modp[j] = mulmod(t % pj->q, Qm, pj);
        for (u32 i = 0; i < L; i++) {
            const po_mod *qi = &md[i];
            u64 hh = 1;
            for (u32 k = 0; k < Lp; k++)
                if (k != j) hh = mulmod(hh, md[L + k].q % qi->q, qi);
            c->phat_modq[j * L + i] = hh;
            /* floor(tQ/p_j) = (tQ - (tQ mod p_j)) / p_j  and  tQ = 0 (mod q_i) */
            u64 v = mulmod(c->tQ_modp[j] % qi->q, invmod_raw(pj->q % qi->q, qi->q), qi);
            c->tQF_modq[j * L + i] = negmod(v, qi->q);
        }
    }
    for (u32 a = 0; a < M; a++) {
        u64 h = 1;
        for (u32 k = 0; k < M; k++)
            if (k != a) h = mulmod(h, md[k].q % md[a].q, &md[a]);
        c->qp_hat_inv[a] = invmod_raw(h, md[a].q);
    }
    {
        const po_mod *mt = &md[M];
        u64 Qt = 1;
        for (u32 k = 0; k < L; k++) Qt = mulmod(Qt, md[k].q % t, mt);
        c->Q_modt = Qt;
    }
    return c;
fail:
    free(chain);
    po_destroy(c);
    return NULL;
}

void po_destroy(po_ctx *c)
{
    if (!c) return;
    if (c->mod) {
        for (u32 a = 0; a <= c->M; a++) free(c->mod[a].tw);
        free(c->mod);
    }
    free(c->slot_pos);
    free(c->qhat_inv);
    free(c->qhat_modp);
    free(c->Q_modp);
    free(c->P_modq);
    free(c->PI_modp);
    free(c->phat_inv);
    free(c->phat_modq);
    free(c->Pfull_modq);
    free(c->qp_hat_inv);
    free(c->tPinv_modq);
    free(c->tQ_modp);
    free(c->tQF_modq);
    free(c->t_inv_modq);
    free(c);
}

u32 po_N(const po_ctx *c) { return c->N; }
u32 po_L(const po_ctx *c) { return c->L; }
u64 po_t(const po_ctx *c) { return c->t; }
void po_moduli(const po_ctx *c, u64 *out)
{
    for (u32 a = 0; a <= c->M; a++) out[a] = c->mod[a].q;
}
u64 po_psi(const po_ctx *c, u32 mi) { return c->mod[mi].psi; }
void po_twiddles(const po_ctx *c, u32 mi, u64 *fwd, u64 *inv)
{
    if (fwd) memcpy(fwd, c->mod[mi].tw, sizeof(u64) * c->N);
    if (inv) memcpy(inv, c->mod[mi].itw, sizeof(u64) * c->N);
}
void po_slot_positions(const po_ctx *c, u32 *pos) { memcpy(pos, c->slot_pos, sizeof(u32) * c->N); }

/* ------------------------------------------------------------------------------------------
 * NTT: forward = Cooley-Tukey, natural in -> bit-reversed out;
 * inverse = Gentleman-Sande, bit-reversed in -> natural out, scaled by N^-1.
 * Harvey lazy butterflies with Shoup twiddles; canonical in, canonical out.
 * ---------------------------------------------------------------------------------------- */
void po_ntt_fwd(const po_ctx *c, u32 mi, u64 *a)
{
    const po_mod *m = &c->mod[mi];
    const u64 q = m->q, q2 = 2 * q;
    const u32 N = c->N;
    u32 t = N;
    for (u32 mm = 1; mm < N; mm <<= 1) {
        t >>= 1;
        for (u32 i = 0; i < mm; i++) {
            const u64 w = m->tw[mm + i], ws = m->tw_sh[mm + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (u32 j = 0; j < t; j++) {
                u64 u = x[j];
                u = u >= q2 ? u - q2 : u;
                u64 v = mul_shoup_lazy(y[j], w, ws, q);
                x[j] = u + v;
                y[j] = u - v + q2;
            }
        }
    }
    for (u32 j = 0; j < N; j++) {
        u64 v = a[j];
        v = v >= q2 ? v - q2 : v;
        a[j] = v >= q ? v - q : v;
    }
}

void po_ntt_inv(const po_ctx *c, u32 mi, u64 *a)
{
    const po_mod *m = &c->mod[mi];
    const u64 q = m->q, q2 = 2 * q;
    const u32 N = c->N;
    u32 t = 1;
    for (u32 h = N >> 1; h >= 1; h >>= 1) {
        for (u32 i = 0; i < h; i++) {
            const u64 w = m->itw[h + i], ws = m->itw_sh[h + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (u32 j = 0; j < t; j++) {
                u64 u = x[j], v = y[j];
                u64 s = u + v;
                x[j] = s >= q2 ? s - q2 : s;
                y[j] = mul_shoup_lazy(u - v + q2, w, ws, q);
            }
        }
        t <<= 1;
    }
    for (u32 j = 0; j < N; j++) {
        u64 v = mul_shoup_lazy(a[j], m->n_inv, m->n_inv_sh, q);
        a[j] = v >= q ? v - q : v;
    }
}

/* ------------------------------------------------------------------------------------------
 * PRNG + samplers
 * ---------------------------------------------------------------------------------------- */
static inline u64 rotl64(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
void po_rng_seed(po_rng *r, u64 seed)
{
    for (int i = 0; i < 4; i++) {
        seed += 0x9E3779B97F4A7C15ULL;
        u64 z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        r->s[i] = z ^ (z >> 31);
    }
}
u64 po_rng_next(po_rng *r)
{
    u64 *s = r->s;
    u64 result = rotl64(s[1] * 5, 7) * 9;
    u64 t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}
u64 po_rng_below(po_rng *r, u64 bound)
{
    /* rejection on the smallest covering mask: exact uniformity, deterministic stream */
    u64 mask = bound - 1;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    for (;;) {
        u64 v = po_rng_next(r) & mask;
        if (v < bound) return v;
    }
}

static void sample_uniform(const po_ctx *c, po_rng *r, u64 *a /*[L][N]*/)
{
    for (u32 i = 0; i < c->L; i++)
        for (u32 j = 0; j < c->N; j++) a[(size_t)i * c->N + j] = po_rng_below(r, c->mod[i].q);
}
/* small signed integer polynomial -> [L][N] residues */
static void lift_small(const po_ctx *c, const int32_t *v, u64 *a)
{
    for (u32 i = 0; i < c->L; i++) {
        u64 q = c->mod[i].q;
        for (u32 j = 0; j < c->N; j++) a[(size_t)i * c->N + j] = v[j] >= 0 ? (u64)v[j] : q - (u64)(-v[j]);
    }
}
static void sample_ternary(const po_ctx *c, po_rng *r, int32_t *v)
{
    for (u32 j = 0; j < c->N; j++) v[j] = (int32_t)po_rng_below(r, 3) - 1;
}
/* centred binomial with variance 10 (sigma = 3.16; OpenFHE's default is a discrete Gaussian
 * with sigma = 3.19 -- an integer-only sampler keeps the fixtures libm-independent) */
static void sample_error(const po_ctx *c, po_rng *r, int32_t *v)
{
    for (u32 j = 0; j < c->N; j++) {
        u64 x = po_rng_next(r);
        v[j] = __builtin_popcountll(x & 0xFFFFF) - __builtin_popcountll((x >> 20) & 0xFFFFF);
    }
}
static void ntt_all_q(const po_ctx *c, u64 *a)
{
    for (u32 i = 0; i < c->L; i++) po_ntt_fwd(c, i, a + (size_t)i * c->N);
}
static void intt_all_q(const po_ctx *c, u64 *a)
{
    for (u32 i = 0; i < c->L; i++) po_ntt_inv(c, i, a + (size_t)i * c->N);
}

/* ------------------------------------------------------------------------------------------
 * packed encoding
 * ---------------------------------------------------------------------------------------- */
int po_encode(const po_ctx *c, const int64_t *slots, u32 nslots, u64 *coeff_t, u64 *eval_q)
{
    const u32 N = c->N;
    const u64 t = c->t;
    if (nslots > N) return -1;
    u64 *u = (u64 *)calloc(N, sizeof(u64));
    for (u32 i = 0; i < nslots; i++) {
        int64_t v = slots[i];
        u64 mag = v < 0 ? (u64)(-v) : (u64)v;
        if (mag >= t) {
            free(u);
            return -1;
        }
        u[c->slot_pos[i]] = v < 0 ? (mag ? t - mag : 0) : mag;
    }
    po_ntt_inv(c, c->M, u);
    if (coeff_t) memcpy(coeff_t, u, sizeof(u64) * N);
    if (eval_q) {
        for (u32 i = 0; i < c->L; i++) {
            u64 q = c->mod[i].q;
            u64 *dst = eval_q + (size_t)i * N;
            for (u32 j = 0; j < N; j++) dst[j] = u[j] > t / 2 ? q - (t - u[j]) : u[j]; /* centred lift */
            po_ntt_fwd(c, i, dst);
        }
    }
    free(u);
    return 0;
}

void po_decode(const po_ctx *c, const u64 *coeff_t, int64_t *slots, u32 nslots)
{
    const u32 N = c->N;
    const u64 t = c->t;
    u64 *u = (u64 *)malloc(sizeof(u64) * N);
    memcpy(u, coeff_t, sizeof(u64) * N);
    po_ntt_fwd(c, c->M, u);
    for (u32 i = 0; i < nslots && i < N; i++) {
        u64 v = u[c->slot_pos[i]];
        slots[i] = v > t / 2 ? -(int64_t)(t - v) : (int64_t)v;
    }
    free(u);
}

/* ------------------------------------------------------------------------------------------
 * keys / encryption / decryption
 * ---------------------------------------------------------------------------------------- */
void po_keygen(const po_ctx *c, u64 seed, u64 *sk)
{
    po_rng r;
    po_rng_seed(&r, seed);
    int32_t *v = (int32_t *)malloc(sizeof(int32_t) * c->N);
    sample_ternary(c, &r, v);
    lift_small(c, v, sk);
    ntt_all_q(c, sk);
    free(v);
}

/* key-switch key from s_from (EVALUATION, [L][N]) to sk: for digit i
 *   b_i = -a_i sk + e_i + s_from * [Q/q_i * (Q/q_i)^-1]   (the CRT one-hot: limb i only)
 * layout ks[i][0]=b_i, ks[i][1]=a_i, each [L][N] */
static void ks_keygen(const po_ctx *c, const u64 *sk, const u64 *s_from, u64 seed, u64 *ks)
{
    const u32 N = c->N, L = c->L;
    const size_t LN = (size_t)L * N;
    po_rng r;
    po_rng_seed(&r, seed);
    int32_t *ev = (int32_t *)malloc(sizeof(int32_t) * N);
    u64 *e = (u64 *)malloc(sizeof(u64) * LN);
    for (u32 i = 0; i < L; i++) {
        u64 *b = ks + ((size_t)i * 2 + 0) * LN;
        u64 *a = ks + ((size_t)i * 2 + 1) * LN;
        sample_uniform(c, &r, a);
        sample_error(c, &r, ev);
        lift_small(c, ev, e);
        ntt_all_q(c, e);
        for (u32 j = 0; j < L; j++) {
            const po_mod *m = &c->mod[j];
            for (u32 n = 0; n < N; n++) {
                size_t x = (size_t)j * N + n;
                u64 v = submod(e[x], mulmod(a[x], sk[x], m), m->q);
                if (j == i) v = addmod(v, s_from[x], m->q);
                b[x] = v;
            }
        }
    }
    free(ev);
    free(e);
}

void po_relin_keygen(const po_ctx *c, const u64 *sk, u64 seed, u64 *evk)
{
    const size_t LN = (size_t)c->L * c->N;
    u64 *s2 = (u64 *)malloc(sizeof(u64) * LN);
    for (u32 j = 0; j < c->L; j++)
        for (u32 n = 0; n < c->N; n++) {
            size_t x = (size_t)j * c->N + n;
            s2[x] = mulmod(sk[x], sk[x], &c->mod[j]);
        }
    ks_keygen(c, sk, s2, seed, evk);
    free(s2);
}

/* EVALUATION-domain index map of the automorphism X -> X^g: out[p] = in[map[p]] */
static void automorph_map(const po_ctx *c, u32 g, u32 *map)
{
    const u32 N = c->N, logN = c->logN;
    const u64 m2 = 2ULL * N;
    for (u32 p = 0; p < N; p++) {
        u64 e = 2ULL * bitrev(p, logN) + 1;
        u64 e2 = (e * g) % m2;
        map[p] = bitrev((u32)((e2 - 1) / 2), logN);
    }
}

void po_rot_keygen(const po_ctx *c, const u64 *sk, u32 g, u64 seed, u64 *rk)
{
    const u32 N = c->N, L = c->L;
    u32 *map = (u32 *)malloc(sizeof(u32) * N);
    automorph_map(c, g, map);
    u64 *sg = (u64 *)malloc(sizeof(u64) * (size_t)L * N);
    for (u32 j = 0; j < L; j++)
        for (u32 p = 0; p < N; p++) sg[(size_t)j * N + p] = sk[(size_t)j * N + map[p]];
    ks_keygen(c, sk, sg, seed, rk);
    free(map);
    free(sg);
}

u32 po_rot_index(const po_ctx *c, int32_t r)
{
    const u64 m2 = 2ULL * c->N;
    u64 base = 5;
    u32 k = (u32)(r < 0 ? -r : r);
    if (r < 0) { /* 5^-1 mod 2N: order of 5 is N/2 */
        u64 inv = 1, b5 = 5;
        u32 e = c->N / 2 - 1;
        while (e) {
            if (e & 1) inv = (inv * b5) % m2;
            b5 = (b5 * b5) % m2;
            e >>= 1;
        }
        base = inv;
    }
    u64 g = 1;
    for (u32 i = 0; i < k; i++) g = (g * base) % m2;
    return (u32)g;
}

void po_encrypt_sk(const po_ctx *c, const u64 *sk, const u64 *coeff_t, u64 seed, u64 *ct)
{
    const u32 N = c->N, L = c->L;
    const size_t LN = (size_t)L * N;
    const u64 t = c->t;
    po_rng r;
    po_rng_seed(&r, seed);
    u64 *c0 = ct, *c1 = ct + LN;
    sample_uniform(c, &r, c1);
    int32_t *ev = (int32_t *)malloc(sizeof(int32_t) * N);
    sample_error(c, &r, ev);
    u64 *e = (u64 *)malloc(sizeof(u64) * LN);
    lift_small(c, ev, e);
    /* add round(Q m / t) = (Q m - [Q m]_t) / t  with [.]_t centred;  mod q_i: -[Q m]_t t^-1 */
    for (u32 n = 0; n < N; n++) {
        u64 rr = mulmod(coeff_t[n] % t, c->Q_modt, &c->mod[c->M]);
        for (u32 i = 0; i < L; i++) {
            const po_mod *m = &c->mod[i];
            u64 term = rr > t / 2 ? mulmod(t - rr, c->t_inv_modq[i], m) : negmod(mulmod(rr, c->t_inv_modq[i], m), m->q);
            size_t x = (size_t)i * N + n;
            e[x] = addmod(e[x], term, m->q);
        }
    }
    ntt_all_q(c, e);
    for (u32 i = 0; i < L; i++) {
        const po_mod *m = &c->mod[i];
        for (u32 n = 0; n < N; n++) {
            size_t x = (size_t)i * N + n;
            c0[x] = submod(e[x], mulmod(c1[x], sk[x], m), m->q);
        }
    }
    free(ev);
    free(e);
}

int po_decrypt(const po_ctx *c, const u64 *sk, const u64 *ct, u32 ncomp, u64 *coeff_t_out)
{
    const u32 N = c->N, L = c->L;
    const size_t LN = (size_t)L * N;
    const u64 t = c->t;
    u64 *x = (u64 *)malloc(sizeof(u64) * LN);
    for (u32 i = 0; i < L; i++) {
        const po_mod *m = &c->mod[i];
        for (u32 n = 0; n < N; n++) {
            size_t k = (size_t)i * N + n;
            u64 s = sk[k];
            u64 acc = addmod(ct[k], mulmod(ct[LN + k], s, m), m->q);
            if (ncomp == 3) acc = addmod(acc, mulmod(ct[2 * LN + k], mulmod(s, s, m), m), m->q);
            x[k] = acc;
        }
    }
    intt_all_q(c, x);
    u64 max_dist = 0;
    for (u32 n = 0; n < N; n++) {
        u64 acc = 0, fsum = 0;
        for (u32 i = 0; i < L; i++) {
            const po_mod *m = &c->mod[i];
            u64 y = mulmod(x[(size_t)i * N + n], c->qhat_inv[i], m);
            u128 ty = (u128)t * y;
            u64 fl = (u64)(ty / m->q), z = (u64)(ty % m->q);
            acc = (acc + fl) % t;
            fsum += fixfrac(z, m);
        }
        u64 rnd = (fsum + FIX_HALF) >> 60;
        coeff_t_out[n] = (acc + rnd) % t;
        u64 f = fsum & (FIX_ONE - 1);
        u64 d = f < FIX_ONE - f ? f : FIX_ONE - f;
        if (d > max_dist) max_dist = d;
    }
    free(x);
    /* invariant noise budget = -log2(2 * max_dist / 2^60), floor */
    if (max_dist == 0) return 58;
    int bl = 64 - __builtin_clzll(max_dist); /* bit length */
    int budget = 59 - bl;
    return budget < 0 ? 0 : (budget > 58 ? 58 : budget);
}

/* ------------------------------------------------------------------------------------------
 * EvalAdd / EvalMult(ct,pt)   
 * ---------------------------------------------------------------------------------------- */
void po_add(const po_ctx *c, const u64 *x, const u64 *y, u64 *out)
{
    const u32 N = c->N, L = c->L;
    for (u32 comp = 0; comp < 2; comp++)
        for (u32 i = 0; i < L; i++) {
            u64 q = c->mod[i].q;
            size_t o = ((size_t)comp * L + i) * N;
            for (u32 n = 0; n < N; n++) out[o + n] = addmod(x[o + n], y[o + n], q);
        }
}

void po_mul_plain(const po_ctx *c, const u64 *x, const u64 *pt, u64 *out)
{
    const u32 N = c->N, L = c->L;
    for (u32 comp = 0; comp < 2; comp++)
        for (u32 i = 0; i < L; i++) {
            const po_mod *m = &c->mod[i];
            size_t o = ((size_t)comp * L + i) * N, po = (size_t)i * N;
            for (u32 n = 0; n < N; n++) out[o + n] = mulmod(x[o + n], pt[po + n], m);
        }
}

/* ------------------------------------------------------------------------------------------
 * RNS base conversions, COEFFICIENT format, one coefficient at a time
 * ---------------------------------------------------------------------------------------- */
/* centred CRT lift of x (given mod Q) into the P limbs; Q limbs copied */
void po_expand_q_to_qp(const po_ctx *c, const u64 *xq, u64 *xqp)
{
    const u32 N = c->N, L = c->L, Lp = L + 1;
    memcpy(xqp, xq, sizeof(u64) * (size_t)L * N);
    u64 y[8];
    for (u32 n = 0; n < N; n++) {
        u64 fsum = 0;
        for (u32 i = 0; i < L; i++) {
            y[i] = mulmod(xq[(size_t)i * N + n], c->qhat_inv[i], &c->mod[i]);
            fsum += fixfrac(y[i], &c->mod[i]);
        }
        u64 v = (fsum + FIX_HALF) >> 60;
        for (u32 j = 0; j < Lp; j++) {
            const po_mod *pj = &c->mod[L + j];
            u128 acc = 0;
            for (u32 i = 0; i < L; i++) acc += (u128)redmod(y[i], pj) * c->qhat_modp[i * Lp + j];
            u64 s = barrett128(acc, pj);
            xqp[(size_t)(L + j) * N + n] = submod(s, mulmod(v, c->Q_modp[j], pj), pj->q);
        }
    }
}

/* x (mod Q) -> round(P x / Q) (mod P), then centred CRT lift of that into the Q limbs */
void po_scale_pq_expand(const po_ctx *c, const u64 *xq, u64 *xqp)
{
    const u32 N = c->N, L = c->L, Lp = L + 1;
    u64 y[8];
    for (u32 n = 0; n < N; n++) {
        u64 fsum = 0;
        u128 itot = 0;
        for (u32 i = 0; i < L; i++) {
            const po_mod *m = &c->mod[i];
            y[i] = mulmod(xq[(size_t)i * N + n], c->qhat_inv[i], m);
            /* y_i P / q_i = y_i floor(P/q_i) + floor(y_i w_i / q_i) + (y_i w_i mod q_i) / q_i,  w_i = P mod q_i */
            u128 prod = (u128)y[i] * c->P_modq[i];
            itot += (u64)(prod / m->q);
            fsum += fixfrac((u64)(prod % m->q), m);
        }
        itot += (fsum + FIX_HALF) >> 60;
        for (u32 j = 0; j < Lp; j++) {
            const po_mod *pj = &c->mod[L + j];
            u128 acc = 0;
            for (u32 i = 0; i < L; i++) acc += (u128)redmod(y[i], pj) * c->PI_modp[i * Lp + j];
            u64 s = barrett128(acc, pj);
            xqp[(size_t)(L + j) * N + n] = addmod(s, barrett128(itot, pj), pj->q);
        }
        /* P -> Q: centred lift */
        u64 yp[8];
        u64 fs2 = 0;
        for (u32 j = 0; j < Lp; j++) {
            const po_mod *pj = &c->mod[L + j];
            yp[j] = mulmod(xqp[(size_t)(L + j) * N + n], c->phat_inv[j], pj);
            fs2 += fixfrac(yp[j], pj);
        }
        u64 v = (fs2 + FIX_HALF) >> 60;
        for (u32 i = 0; i < L; i++) {
            const po_mod *qi = &c->mod[i];
            u128 acc = 0;
            for (u32 j = 0; j < Lp; j++) acc += (u128)redmod(yp[j], qi) * c->phat_modq[j * L + i];
            u64 s = barrett128(acc, qi);
            xqp[(size_t)i * N + n] = submod(s, mulmod(v, c->Pfull_modq[i], qi), qi->q);
        }
    }
}

/* d (mod QP) -> round(t d / P) (mod Q) */
void po_scale_round_tp(const po_ctx *c, const u64 *xqp, u64 *xq)
{
    const u32 N = c->N, L = c->L, Lp = L + 1;
    u64 yp[8];
    for (u32 n = 0; n < N; n++) {
        u64 fsum = 0;
        u128 itot = 0;
        for (u32 j = 0; j < Lp; j++) {
            const po_mod *pj = &c->mod[L + j];
            yp[j] = mulmod(xqp[(size_t)(L + j) * N + n], c->qp_hat_inv[L + j], pj);
            u128 prod = (u128)yp[j] * c->tQ_modp[j];
            itot += (u64)(prod / pj->q);
            fsum += fixfrac((u64)(prod % pj->q), pj);
        }
        itot += (fsum + FIX_HALF) >> 60;
        for (u32 k = 0; k < L; k++) {
            const po_mod *qk = &c->mod[k];
            u128 acc = (u128)xqp[(size_t)k * N + n] * c->tPinv_modq[k];
            for (u32 j = 0; j < Lp; j++) acc += (u128)redmod(yp[j], qk) * c->tQF_modq[j * L + k];
            u64 s = barrett128(acc, qk);
            xq[(size_t)k * N + n] = addmod(s, barrett128(itot, qk), qk->q);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * EvalMult(ct,ct): HPS P-over-Q tensor + BV relinearisation (row A7)
 * ---------------------------------------------------------------------------------------- */
void po_mul_tensor(const po_ctx *c, const u64 *x, const u64 *y, u64 *out3)
{
    const u32 N = c->N, L = c->L, M = c->M;
    const size_t LN = (size_t)L * N, MN = (size_t)M * N;
    u64 *xc = (u64 *)malloc(sizeof(u64) * 2 * LN);
    u64 *yc = (u64 *)malloc(sizeof(u64) * 2 * LN);
    u64 *e = (u64 *)malloc(sizeof(u64) * 4 * MN); /* a0 a1 b0 b1 over QP */
    u64 *d = (u64 *)malloc(sizeof(u64) * 3 * MN);
    memcpy(xc, x, sizeof(u64) * 2 * LN);
    memcpy(yc, y, sizeof(u64) * 2 * LN);
    for (u32 k = 0; k < 2; k++) { /* (1) INTT over Q */
        intt_all_q(c, xc + k * LN);
        intt_all_q(c, yc + k * LN);
    }
    for (u32 k = 0; k < 2; k++) { /* (2) operand 1: Q -> QP; operand 2: scale by P/Q, P -> QP */
        po_expand_q_to_qp(c, xc + k * LN, e + k * MN);
        po_scale_pq_expand(c, yc + k * LN, e + (2 + k) * MN);
    }
    for (u32 k = 0; k < 4; k++) /* (3) NTT over QP */
        for (u32 a = 0; a < M; a++) po_ntt_fwd(c, a, e + k * MN + (size_t)a * N);
    for (u32 a = 0; a < M; a++) { /* (4) tensor */
        const po_mod *m = &c->mod[a];
        const u64 *a0 = e + (size_t)a * N, *a1 = e + MN + (size_t)a * N;
        const u64 *b0 = e + 2 * MN + (size_t)a * N, *b1 = e + 3 * MN + (size_t)a * N;
        u64 *d0 = d + (size_t)a * N, *d1 = d + MN + (size_t)a * N, *d2 = d + 2 * MN + (size_t)a * N;
        for (u32 n = 0; n < N; n++) {
            d0[n] = mulmod(a0[n], b0[n], m);
            d1[n] = barrett128((u128)a0[n] * b1[n] + (u128)a1[n] * b0[n], m);
            d2[n] = mulmod(a1[n], b1[n], m);
        }
    }
    for (u32 k = 0; k < 3; k++) /* (5) INTT over QP */
        for (u32 a = 0; a < M; a++) po_ntt_inv(c, a, d + k * MN + (size_t)a * N);
    for (u32 k = 0; k < 3; k++) { /* (6) scale by t/P into Q, (7) NTT over Q */
        po_scale_round_tp(c, d + k * MN, out3 + k * LN);
        ntt_all_q(c, out3 + k * LN);
    }
    free(xc);
    free(yc);
    free(e);
    free(d);
}

/* add the key-switched image of poly `src` (EVALUATION, [L][N]) under key ks to (out0,out1) */
static void keyswitch_acc(const po_ctx *c, const u64 *src, const u64 *ks, u64 *out0, u64 *out1)
{
    const u32 N = c->N, L = c->L;
    const size_t LN = (size_t)L * N;
    u64 *sc = (u64 *)malloc(sizeof(u64) * LN);
    u64 *dig = (u64 *)malloc(sizeof(u64) * N);
    memcpy(sc, src, sizeof(u64) * LN);
    intt_all_q(c, sc);
    for (u32 i = 0; i < L; i++) {
        const u64 qi = c->mod[i].q;
        const u64 *kb = ks + ((size_t)i * 2 + 0) * LN, *ka = ks + ((size_t)i * 2 + 1) * LN;
        for (u32 j = 0; j < L; j++) {
            const po_mod *m = &c->mod[j];
            const u64 *dj;
            if (j == i) {
                dj = src + (size_t)i * N; /* digit i in its own limb is the EVALUATION limb itself */
            } else {
                /* centred lift of the residue mod q_i into q_j, then NTT */
                u64 qi_mod = redmod(qi, m);
                for (u32 n = 0; n < N; n++) {
                    u64 v = sc[(size_t)i * N + n];
                    u64 r = redmod(v, m);
                    dig[n] = v > qi / 2 ? submod(r, qi_mod, m->q) : r;
                }
                po_ntt_fwd(c, j, dig);
                dj = dig;
            }
            u64 *o0 = out0 + (size_t)j * N, *o1 = out1 + (size_t)j * N;
            const u64 *kbj = kb + (size_t)j * N, *kaj = ka + (size_t)j * N;
            for (u32 n = 0; n < N; n++) {
                o0[n] = addmod(o0[n], mulmod(dj[n], kbj[n], m), m->q);
                o1[n] = addmod(o1[n], mulmod(dj[n], kaj[n], m), m->q);
            }
        }
    }
    free(sc);
    free(dig);
}

void po_relin(const po_ctx *c, const u64 *ct3, const u64 *evk, u64 *out)
{
    const size_t LN = (size_t)c->L * c->N;
    memmove(out, ct3, sizeof(u64) * 2 * LN);
    keyswitch_acc(c, ct3 + 2 * LN, evk, out, out + LN);
}

void po_mul(const po_ctx *c, const u64 *x, const u64 *y, const u64 *evk, u64 *out)
{
    const size_t LN = (size_t)c->L * c->N;
    u64 *d = (u64 *)malloc(sizeof(u64) * 3 * LN);
    po_mul_tensor(c, x, y, d);
    po_relin(c, d, evk, out);
    free(d);
}

void po_automorph(const po_ctx *c, const u64 *x, u32 g, const u64 *rk, u64 *out)
{
    const u32 N = c->N, L = c->L;
    const size_t LN = (size_t)L * N;
    u32 *map = (u32 *)malloc(sizeof(u32) * N);
    automorph_map(c, g, map);
    u64 *p1 = (u64 *)malloc(sizeof(u64) * LN);
    for (u32 j = 0; j < L; j++)
        for (u32 p = 0; p < N; p++) {
            out[(size_t)j * N + p] = x[(size_t)j * N + map[p]];
            p1[(size_t)j * N + p] = x[LN + (size_t)j * N + map[p]];
        }
    memset(out + LN, 0, sizeof(u64) * LN);
    keyswitch_acc(c, p1, rk, out, out + LN);
    free(map);
    free(p1);
}

/* ------------------------------------------------------------------------------------------
 * BatchedFHEHIPPIE::run()  (reference BatchedFHEHIPPIE.cpp:88-129), same loop nest, one
 * unfused pass per Eval* call, allocation-free inside the loops.
 * ---------------------------------------------------------------------------------------- */
void po_pie_run(const po_ctx *c, u32 K, u32 b, u32 E, const u64 *idx, const u64 *minus, const u64 *db,
                const u64 *masks, const u64 *evk, u64 *out, u32 bin_begin, u32 bin_end)
{
    const size_t LN = (size_t)c->L * c->N, CT = 2 * LN;
    u64 *inner = (u64 *)malloc(sizeof(u64) * CT);
    u64 *tmp = (u64 *)malloc(sizeof(u64) * CT);
    u64 *prod = (u64 *)malloc(sizeof(u64) * CT);
    if (bin_end > b) bin_end = b;
    for (u32 bin = bin_begin; bin < bin_end; bin++) {            /* .cpp:91 */
        for (u32 h = 0; h < K; h++) {                            /* .cpp:96 */
            for (u32 j = 0; j < E; j++) {                        /* .cpp:101 */
                const u64 *ct = idx + ((size_t)h * E + j) * CT;
                const u64 *pt = db + (((size_t)h * b + bin) * E + j) * LN;
                if (j == 0) {
                    po_mul_plain(c, ct, pt, inner);              /* .cpp:108 */
                } else {
                    po_mul_plain(c, ct, pt, tmp);                /* .cpp:113 */
                    po_add(c, inner, tmp, inner);                /* .cpp:112 */
                }
            }
            po_add(c, inner, minus, inner);                      /* .cpp:116 */
            if (h == 0)
                memcpy(prod, inner, sizeof(u64) * CT);           /* .cpp:119 */
            else
                po_mul(c, prod, inner, evk, prod);               /* .cpp:123 */
        }
        po_mul_plain(c, prod, masks + (size_t)bin * LN, out + (size_t)bin * CT); /* .cpp:126-127 */
    }
    free(inner);
    free(tmp);
    free(prod);
}
