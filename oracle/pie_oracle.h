/*
 * pie_oracle.h -- CPU restatement (plain C) of the server-side batched-FHE PIE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the CPU baseline.  The product (libpiehip.so) never links or calls it.
 *
 * What it restates
 *   control flow / data layout : reference src/Common/Crypto/PrivateIndexedEqualityCheck/
 *                                BatchedFHEHIPPIE.cpp:9-129 (ctor packing + run())
 *   inputs / outputs           : reference src/Client/FHE/BatchedFHEPSIClient.cpp:107-193,249-265
 *   arithmetic                 : the reference delegates ALL arithmetic to OpenFHE
 *                                (openfheorg/openfhe-development; un-vendored, version NOT
 *                                pinned: CMakeLists.txt:10, hint "0.9.2" at CMakeLists.txt:15),
 *                                which is absent from the build container.  This file restates
 *                                the published BFV-RNS algorithms OpenFHE implements for the
 *                                calls at BatchedFHEHIPPIE.cpp:68,81,108,112,113,116,123,126:
 *                                negacyclic NTT (Cooley-Tukey / Gentleman-Sande, bit-reversed
 *                                evaluation order), packed encoding, HPS "P-over-Q" ciphertext
 *                                multiplication (Halevi-Polyakov-Shoup 2019; Kim-Polyakov-Zucca
 *                                2021), BV/RNS-digit relinearisation, automorphism + key switch.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext-bit level -- the reference holds no numeric
 * golden vectors (tests/TestBatchedFHEPIE.cpp:139-149 only prints "Matches") and OpenFHE cannot
 * be built here.  What IS pinned: decrypted-slot semantics (KAT-0/KAT-1/KAT-2 of SURVEY.md 8c),
 * checked in tests/ against this oracle, and this oracle against exact big-integer mathematics
 * (oracle/exact.py).  Floating-point rounding terms of OpenFHE's HPS are replaced by an
 * integer fixed-point rule (60 fractional bits) so that CPU and GPU agree bit for bit.
 */
#ifndef PIE_ORACLE_H
#define PIE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct po_ctx po_ctx;

/* ---- parameter generation ------------------------------------------------------------- */
/* count primes q = 1 (mod 2N), descending, strictly below `below` (pass 1<<bits for the first). */
int po_gen_primes(uint32_t N, uint64_t below, uint32_t count, uint64_t *out);
int po_is_prime(uint64_t n);
/* smallest primitive 2N-th root of unity mod q (0 on failure) */
uint64_t po_min_root(uint64_t q, uint32_t N);

/* q: L primes (basis Q), p: L+1 primes (auxiliary basis P); pass NULL for the default chain
 * (largest primes < 2^60 = 1 mod 2N, then continuing downwards for P).  t prime, t = 1 mod 2N. */
po_ctx *po_create(uint32_t N, uint32_t L, uint64_t t, const uint64_t *q, const uint64_t *p);
void po_destroy(po_ctx *c);
uint32_t po_N(const po_ctx *c);
uint32_t po_L(const po_ctx *c);
uint64_t po_t(const po_ctx *c);
/* moduli in order q_0..q_{L-1}, p_0..p_L, t  (2L+2 entries) */
void po_moduli(const po_ctx *c, uint64_t *out);
uint64_t po_psi(const po_ctx *c, uint32_t mod_index);
/* table exports for cross-checks against the product's own tables */
void po_twiddles(const po_ctx *c, uint32_t mod_index, uint64_t *fwd /*[N]*/, uint64_t *inv /*[N]*/);
void po_slot_positions(const po_ctx *c, uint32_t *pos /*[N]*/);

/* ---- single-limb transforms (in place, canonical [0,q) in and out) -------------------- */
void po_ntt_fwd(const po_ctx *c, uint32_t mod_index, uint64_t *a);
void po_ntt_inv(const po_ctx *c, uint32_t mod_index, uint64_t *a);

/* ---- packed encoding (reference call: MakePackedPlaintext, BatchedFHEHIPPIE.cpp:68,81) -- */
/* slots[nslots] signed values (|v| < t); unused slots are 0.
 * coeff_t: [N] coefficients mod t (may be NULL); eval_q: [L][N] EVALUATION-format limbs (may be NULL) */
int po_encode(const po_ctx *c, const int64_t *slots, uint32_t nslots, uint64_t *coeff_t, uint64_t *eval_q);
/* inverse: coefficients mod t -> signed slots (centered) */
void po_decode(const po_ctx *c, const uint64_t *coeff_t, int64_t *slots, uint32_t nslots);

/* ---- keys, encryption, decryption (client side of the harness) ------------------------ */
/* sk: [L][N] EVALUATION format; deterministic from seed */
void po_keygen(const po_ctx *c, uint64_t seed, uint64_t *sk);
/* BV relinearisation key, one digit per RNS limb: evk[L digits][2 (b,a)][L limbs][N] */
void po_relin_keygen(const po_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *evk);
/* key-switch key for automorphism X -> X^g (same layout as evk) */
void po_rot_keygen(const po_ctx *c, const uint64_t *sk, uint32_t g, uint64_t seed, uint64_t *rk);
/* secret-key encryption (BatchedFHEPSIClient.cpp:156,166): ct [2][L][N] EVALUATION format */
void po_encrypt_sk(const po_ctx *c, const uint64_t *sk, const uint64_t *coeff_t, uint64_t seed, uint64_t *ct);
/* decrypt a 2-component (ncomp=2) or 3-component ciphertext; returns the invariant noise budget
 * in bits (min over coefficients, capped at 58); coeff_t_out [N] */
int po_decrypt(const po_ctx *c, const uint64_t *sk, const uint64_t *ct, uint32_t ncomp, uint64_t *coeff_t_out);

/* ---- homomorphic operations (EVALUATION format in and out) ----------------------------- */
void po_add(const po_ctx *c, const uint64_t *x, const uint64_t *y, uint64_t *out);          /* EvalAdd(ct,ct) */
void po_mul_plain(const po_ctx *c, const uint64_t *x, const uint64_t *pt, uint64_t *out);   /* EvalMult(ct,pt) */
void po_mul_tensor(const po_ctx *c, const uint64_t *x, const uint64_t *y, uint64_t *out3);  /* HPS P-over-Q, no relin: [3][L][N] */
void po_relin(const po_ctx *c, const uint64_t *ct3, const uint64_t *evk, uint64_t *out);    /* BV key switch of d2 */
void po_mul(const po_ctx *c, const uint64_t *x, const uint64_t *y, const uint64_t *evk, uint64_t *out); /* EvalMult(ct,ct) */
/* EvalAtIndex-style: automorphism X->X^g followed by key switch with rk */
void po_automorph(const po_ctx *c, const uint64_t *x, uint32_t g, const uint64_t *rk, uint64_t *out);
uint32_t po_rot_index(const po_ctx *c, int32_t r); /* g = 5^r mod 2N (r<0: inverse) */

/* building blocks exposed for kernel-level parity tests ([..][N] limb arrays, COEFFICIENT format) */
void po_expand_q_to_qp(const po_ctx *c, const uint64_t *xq /*[L][N]*/, uint64_t *xqp /*[2L+1][N]*/);
void po_scale_pq_expand(const po_ctx *c, const uint64_t *xq /*[L][N]*/, uint64_t *xqp /*[2L+1][N]*/);
void po_scale_round_tp(const po_ctx *c, const uint64_t *xqp /*[2L+1][N]*/, uint64_t *xq /*[L][N]*/);

/* ---- the hot path: BatchedFHEHIPPIE::run() (BatchedFHEHIPPIE.cpp:88-129) --------------- */
/* idx   [K][E][2][L][N]  index ciphertexts            (indexMatrix,            .hpp:25)
 * minus [2][L][N]        minus-compare ciphertext     (minusCompareElement,    .hpp:26)
 * db    [K][b][E][L][N]  packed plaintexts            (vectorizedHCT,          .hpp:23)
 * masks [b][L][N]        random mask plaintexts       (preCalcRandomMask,      .hpp:27)
 * evk   [L][2][L][N]     relinearisation key
 * out   [b][2][L][N]     resultList                   (.hpp:24)
 * bin_begin/bin_end select a slice of bin layers (for the sharded run and bounded timing). */
void po_pie_run(const po_ctx *c, uint32_t K, uint32_t b, uint32_t E, const uint64_t *idx, const uint64_t *minus,
                const uint64_t *db, const uint64_t *masks, const uint64_t *evk, uint64_t *out, uint32_t bin_begin,
                uint32_t bin_end);

/* deterministic PRNG used by every sampler above (splitmix64 seeding a xoshiro256**) */
typedef struct {
    uint64_t s[4];
} po_rng;
void po_rng_seed(po_rng *r, uint64_t seed);
uint64_t po_rng_next(po_rng *r);
uint64_t po_rng_below(po_rng *r, uint64_t bound);

#ifdef __cplusplus
}
#endif
#endif
