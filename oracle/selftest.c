/*
 * selftest.c -- drives every entry point of the CPU oracle once on a small ring, for the sanitizer build
 * (make -C oracle selftest_san; tests/test_oracle_sanitize.py runs it under ASan + UBSan on the CPU).
 * TEST INFRASTRUCTURE ONLY (see pie_oracle.h).
 *
 * The scenario is the reference's own test shape at reduced size (tests/TestBatchedFHEPIE.cpp:89-139:
 * hash a server set into the nested table, pack it, one-hot index matrix + minus vector from the client's
 * Cuckoo table, run(), decrypt, count zero slots) plus the arithmetic of tests/TestOpenFHE.cpp:36-65
 * (add, multiply, rotate by +-1 of short vectors).  Exit code 0 = every check passed.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pie_hashing.h"
#include "pie_oracle.h"

static int fails = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            fprintf(stderr, "selftest: %s failed (line %d)\n", #cond, __LINE__); \
            fails++;                                                     \
        }                                                                \
    } while (0)

static void *xmalloc(size_t bytes)
{
    void *p = calloc(bytes ? bytes : 1, 1);
    if (!p) {
        fprintf(stderr, "selftest: out of memory\n");
        exit(2);
    }
    return p;
}

int main(void)
{
    const uint32_t N = 1024, L = 2, M = 2 * L + 1;
    const uint64_t t = 65537;
    po_ctx *c = po_create(N, L, t, NULL, NULL);
    CHECK(c != NULL);
    if (!c) return 1;
    CHECK(po_N(c) == N && po_L(c) == L && po_t(c) == t);
    const size_t LN = (size_t)L * N;

    /* parameters and tables */
    uint64_t mod[2 * 2 + 2];
    po_moduli(c, mod);
    for (uint32_t i = 0; i < M + 1; i++) {
        CHECK(po_is_prime(mod[i]) && mod[i] % (2 * N) == 1);
        CHECK(po_psi(c, i) == po_min_root(mod[i], N));
    }
    uint64_t chain[3];
    CHECK(po_gen_primes(N, 1ULL << 60, 3, chain) == 0 || chain[0] == mod[0]);
    uint64_t *fw = xmalloc(8 * N), *iv = xmalloc(8 * N);
    uint32_t *pos = xmalloc(4 * N);
    po_twiddles(c, 0, fw, iv);
    po_slot_positions(c, pos);
    CHECK(fw[0] == 1 || fw[1] != 0);

    /* transforms: inverse(forward(a)) == a for every modulus */
    po_rng rng;
    po_rng_seed(&rng, 5);
    uint64_t *a = xmalloc(8 * N), *a0 = xmalloc(8 * N);
    for (uint32_t mi = 0; mi <= M; mi++) {
        for (uint32_t j = 0; j < N; j++) a0[j] = a[j] = po_rng_below(&rng, mod[mi]);
        po_ntt_fwd(c, mi, a);
        po_ntt_inv(c, mi, a);
        CHECK(memcmp(a, a0, 8 * N) == 0);
    }

    /* encode / decode round trip with negative slots */
    int64_t *sl = xmalloc(8 * N), *sl2 = xmalloc(8 * N);
    for (uint32_t j = 0; j < N; j++) sl[j] = (int64_t)po_rng_below(&rng, t) - (int64_t)(t / 2);
    uint64_t *cf = xmalloc(8 * N), *ev = xmalloc(8 * LN);
    CHECK(po_encode(c, sl, N, cf, ev) == 0);
    po_decode(c, cf, sl2, N);
    CHECK(memcmp(sl, sl2, 8 * N) == 0);

    /* keys; add, ct x pt, ct x ct (+ relin), rotation on 12-vectors (tests/TestOpenFHE.cpp:36-65) */
    uint64_t *sk = xmalloc(8 * LN), *evk = xmalloc(8 * L * 2 * LN), *rk = xmalloc(8 * L * 2 * LN);
    po_keygen(c, 11, sk);
    po_relin_keygen(c, sk, 12, evk);
    const uint32_t g1 = po_rot_index(c, 1);
    po_rot_keygen(c, sk, g1, 13, rk);
    int64_t v1[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, v2[12] = {3, 2, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    uint64_t *c1 = xmalloc(8 * 2 * LN), *c2 = xmalloc(8 * 2 * LN), *c3 = xmalloc(8 * 3 * LN), *cr = xmalloc(8 * 2 * LN);
    uint64_t *pt2 = xmalloc(8 * LN), *dec = xmalloc(8 * N);
    int64_t out12[12];
    CHECK(po_encode(c, v1, 12, cf, NULL) == 0);
    po_encrypt_sk(c, sk, cf, 21, c1);
    CHECK(po_encode(c, v2, 12, cf, pt2) == 0);
    po_encrypt_sk(c, sk, cf, 22, c2);
    po_add(c, c1, c2, cr);
    CHECK(po_decrypt(c, sk, cr, 2, dec) > 0);
    po_decode(c, dec, out12, 12);
    for (int i = 0; i < 12; i++) CHECK(out12[i] == v1[i] + v2[i]);
    po_mul_plain(c, c1, pt2, cr);
    CHECK(po_decrypt(c, sk, cr, 2, dec) > 0);
    po_decode(c, dec, out12, 12);
    for (int i = 0; i < 12; i++) CHECK(out12[i] == v1[i] * v2[i]);
    po_mul_tensor(c, c1, c2, c3);
    CHECK(po_decrypt(c, sk, c3, 3, dec) > 0);
    po_decode(c, dec, out12, 12);
    for (int i = 0; i < 12; i++) CHECK(out12[i] == v1[i] * v2[i]);
    po_relin(c, c3, evk, cr);
    CHECK(po_decrypt(c, sk, cr, 2, dec) > 0);
    po_mul(c, c1, c2, evk, c3);
    CHECK(memcmp(c3, cr, 8 * 2 * LN) == 0);
    po_decode(c, dec, out12, 12);
    for (int i = 0; i < 12; i++) CHECK(out12[i] == v1[i] * v2[i]);
    po_automorph(c, c1, g1, rk, cr);
    CHECK(po_decrypt(c, sk, cr, 2, dec) > 0);
    po_decode(c, dec, out12, 12);
    for (int i = 0; i < 11; i++) CHECK(out12[i] == v1[i + 1]);

    /* base conversions: Q -> QP keeps the Q limbs; scale-round maps back into Q */
    uint64_t *xq = xmalloc(8 * LN), *xqp = xmalloc(8 * M * N), *yq = xmalloc(8 * LN);
    for (uint32_t i = 0; i < L; i++)
        for (uint32_t j = 0; j < N; j++) xq[(size_t)i * N + j] = po_rng_below(&rng, mod[i]);
    po_expand_q_to_qp(c, xq, xqp);
    CHECK(memcmp(xq, xqp, 8 * LN) == 0);
    po_scale_pq_expand(c, xq, xqp);
    po_scale_round_tp(c, xqp, yq);
    for (size_t i = 0; i < LN; i++) CHECK(yq[i] < mod[i / N]);

    /* the PSI of tests/TestBatchedFHEPIE.cpp at reduced size */
    const uint32_t k = 2, e = 8, K = 2, b = 3, E = 4, B = k * e;
    const uint32_t nS = 60, nC = 6;
    uint64_t items[70];
    for (uint32_t i = 0; i < 70; i++) items[i] = 1000 + 7 * i;
    uint64_t client[6] = {items[3], items[17], items[40], items[64], items[66], items[68]};  /* 3 of 6 are server items */
    ph_tab *tab = ph_tab_create(987654321, k + K);
    CHECK(tab != NULL);
    CHECK(ph_tab_hash(tab, 5, 0) != ph_tab_hash(tab, 5, 1));
    uint64_t *tbl = xmalloc(8 * (size_t)k * e * K * b * E);
    CHECK(ph_hct_build(tab, items, nS, k, e, K, b, E, 1, tbl) == 0);
    ph_hct_shuffle_bins(tbl, k, e, K, b, E, 2);
    int64_t *slots = xmalloc(8 * (size_t)K * b * E * B), *mslots = xmalloc(8 * (size_t)b * B);
    ph_pack_db(tbl, k, e, K, b, E, slots);
    ph_masks(t, b, B, 3, mslots);
    uint64_t *ctab = xmalloc(8 * (size_t)k * e);
    CHECK(ph_client_build(tab, client, nC, k, e, 4, ctab) == 0);
    int64_t *index = xmalloc(8 * (size_t)K * E * B), *minus_v = xmalloc(8 * B);
    ph_client_vectors(tab, ctab, k, e, K, E, index, minus_v);
    uint64_t *db = xmalloc(8 * (size_t)K * b * E * LN), *masks = xmalloc(8 * (size_t)b * LN);
    for (size_t i = 0; i < (size_t)K * b * E; i++) CHECK(po_encode(c, slots + i * B, B, NULL, db + i * LN) == 0);
    for (size_t i = 0; i < b; i++) CHECK(po_encode(c, mslots + i * B, B, NULL, masks + i * LN) == 0);
    uint64_t *idx = xmalloc(8 * (size_t)K * E * 2 * LN), *minus = xmalloc(8 * 2 * LN), *res = xmalloc(8 * (size_t)b * 2 * LN);
    for (size_t i = 0; i < (size_t)K * E; i++) {
        CHECK(po_encode(c, index + i * B, B, cf, NULL) == 0);
        po_encrypt_sk(c, sk, cf, 100 + i, idx + i * 2 * LN);
    }
    CHECK(po_encode(c, minus_v, B, cf, NULL) == 0);
    po_encrypt_sk(c, sk, cf, 99, minus);
    po_pie_run(c, K, b, E, idx, minus, db, masks, evk, res, 0, b);
    /* a slice of bin layers gives the same ciphertexts as the whole run */
    uint64_t *res1 = xmalloc(8 * (size_t)b * 2 * LN);
    po_pie_run(c, K, b, E, idx, minus, db, masks, evk, res1, 1, 2);
    CHECK(memcmp(res1 + 2 * LN, res + 2 * LN, 8 * 2 * LN) == 0);
    int64_t *decs = xmalloc(8 * (size_t)b * B);
    for (uint32_t i = 0; i < b; i++) {
        CHECK(po_decrypt(c, sk, res + (size_t)i * 2 * LN, 2, dec) > 0);
        po_decode(c, dec, decs + (size_t)i * B, B);
    }
    uint64_t found[16];
    const size_t nf = ph_client_scan(ctab, k, e, b, decs, found);
    CHECK(nf == 3);
    for (size_t i = 0; i < nf; i++) CHECK(found[i] == items[3] || found[i] == items[17] || found[i] == items[40]);

    ph_tab_destroy(tab);
    po_destroy(c);
    free(fw); free(iv); free(pos); free(a); free(a0); free(sl); free(sl2); free(cf); free(ev); free(sk); free(evk); free(rk);
    free(c1); free(c2); free(c3); free(cr); free(pt2); free(dec); free(xq); free(xqp); free(yq); free(tbl); free(slots);
    free(mslots); free(ctab); free(index); free(minus_v); free(db); free(masks); free(idx); free(minus); free(res); free(res1);
    free(decs);
    printf(fails ? "selftest FAILED (%d checks)\n" : "selftest ok\n", fails);
    return fails ? 1 : 0;
}
