// piehip_client.cpp -- client-side harness (the client role of src/Client/FHE/BatchedFHEPSIClient.cpp): key generation, packed
// encryption, decryption.  Not part of the server hot path; it produces the hot path's inputs and reads its outputs.
#include "piehip_ctx.hpp"

#include <thread>

using namespace piehip;

namespace {
// xoshiro256** seeded through splitmix64, rejection sampling on the smallest covering mask
struct HostRng {
    u64 s[4];
    explicit HostRng(u64 seed)
    {
        for (int i = 0; i < 4; i++) {
            seed += 0x9E3779B97F4A7C15ULL;
            u64 z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            s[i] = z ^ (z >> 31);
        }
    }
    static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    u64 next()
    {
        const u64 result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    u64 below(u64 bound)
    {
        u64 mask = bound - 1;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        for (;;) {
            const u64 v = next() & mask;
            if (v < bound) return v;
        }
    }
};
void sample_uniform(HostRng &r, const HostParams &hp, u64 *a)  // [L][N], independent per limb
{
    for (u32 i = 0; i < hp.L; i++)
        for (u32 j = 0; j < hp.N; j++) a[(size_t)i * hp.N + j] = r.below(hp.moduli[i]);
}
void sample_error(HostRng &r, u32 N, int32_t *e)  // centred binomial, variance 10
{
    for (u32 j = 0; j < N; j++) {
        const u64 x = r.next();
        e[j] = __builtin_popcountll(x & 0xFFFFF) - __builtin_popcountll((x >> 20) & 0xFFFFF);
    }
}
}  // namespace

extern "C" {

int piehip_client_keygen(piehip_handle h, uint64_t seed, uint64_t *sk)
{
    NEED(h);
    if (!sk) return fail(PIEHIP_EINVAL, "null sk");
    HIPCHK(hipSetDevice(h->device));
    const u32 N = h->hp.N, L = h->hp.L;
    HostRng r(seed);
    std::vector<u64> host((size_t)L * N);
    for (u32 j = 0; j < N; j++) {
        const int v = (int)r.below(3) - 1;
        for (u32 i = 0; i < L; i++) host[(size_t)i * N + j] = v >= 0 ? (u64)v : h->hp.moduli[i] - 1;
    }
    Tmp tmp;
    TMPGET(d, (size_t)L * N);
    HIPCHK(hipMemcpy(d, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
    launch_ntt(h->plan, d, L, 0, L, false, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(sk, d, host.size() * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

// BV key-switching key from s_from to sk (oracle: ks_keygen): row i = (e_i - a_i s + [j == i] s_from, a_i).
// g == 0: s_from = s^2 (EvalMultKeyGen); otherwise s_from = s(X^g) (EvalAtIndexKeyGen / EvalSumKeyGen).
static int client_ks_keygen(piehip_ctx *h, const uint64_t *sk, uint32_t g, uint64_t seed, uint64_t *out)
{
    HIPCHK(hipSetDevice(h->device));
    const u32 N = h->hp.N, L = h->hp.L;
    const size_t LN = h->LN();
    HostRng r(seed);
    std::vector<u64> ks((size_t)L * 2 * LN), e((size_t)L * LN);
    std::vector<int32_t> ev(N);
    for (u32 i = 0; i < L; i++) {
        sample_uniform(r, h->hp, &ks[((size_t)i * 2 + 1) * LN]);
        sample_error(r, N, ev.data());
        for (u32 l = 0; l < L; l++)
            for (u32 j = 0; j < N; j++) e[(size_t)i * LN + (size_t)l * N + j] = ev[j] >= 0 ? (u64)ev[j] : h->hp.moduli[l] - (u64)(-ev[j]);
    }
    Tmp tmp;
    TMPGET(d_ks, ks.size());
    TMPGET(d_e, e.size());
    TMPGET(d_sk, LN);
    TMPGET(d_s2, LN);
    TMPGET(d_mapw, (N + 1) / 2 + 1);
    HIPCHK(hipMemcpy(d_ks, ks.data(), ks.size() * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_e, e.data(), e.size() * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_sk, sk, LN * sizeof(u64), hipMemcpyHostToDevice));
    launch_ntt(h->plan, d_e, L * L, 0, L, false, h->stream);
    if (g) {
        std::vector<u32> map = h->hp.automorph_map(g);
        HIPCHK(hipMemcpy(d_mapw, map.data(), sizeof(u32) * N, hipMemcpyHostToDevice));
        launch_permute(N, d_sk, (const u32 *)d_mapw, d_s2, L, h->stream);
    } else {
        launch_square(h->d_dc, N, L, d_sk, d_s2, h->stream);
    }
    launch_ks_finish(h->d_dc, N, L, d_e, d_sk, d_s2, d_ks, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d_ks, ks.size() * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_client_relin_keygen(piehip_handle h, const uint64_t *sk, uint64_t seed, uint64_t *evk)
{
    NEED(h);
    if (!sk || !evk) return fail(PIEHIP_EINVAL, "null operand");
    return client_ks_keygen(h, sk, 0, seed, evk);
}

int piehip_rotation_galois(piehip_handle h, int32_t index, uint32_t *g)
{
    NEED_RO(h);
    if (!g) return fail(PIEHIP_EINVAL, "null out");
    const u32 N = h->hp.N;
    const u64 m2 = 2ULL * N;
    u64 base = 5;
    if (index < 0) base = powmod(5, N / 2 - 1, m2);  // 5 has order N/2 modulo 2N
    const u64 k = (u64)(index < 0 ? -(int64_t)index : (int64_t)index);
    *g = (u32)powmod(base, k % (N / 2 ? N / 2 : 1), m2);
    return PIEHIP_OK;
}

int piehip_client_rot_keygen(piehip_handle h, const uint64_t *sk, int32_t index, uint64_t seed, uint64_t *rk)
{
    NEED(h);
    if (!sk || !rk) return fail(PIEHIP_EINVAL, "null operand");
    uint32_t g = 1;
    int rc = piehip_rotation_galois(h, index, &g);
    if (rc) return rc;
    if (g == 1) return fail(PIEHIP_EINVAL, "rotation index is a multiple of the row length");
    return client_ks_keygen(h, sk, g, seed, rk);
}

int piehip_client_encrypt(piehip_handle h, const uint64_t *sk, const int64_t *slots, uint32_t nct, uint32_t B,
                          const uint64_t *seeds, uint64_t *out)
{
    NEED(h);
    if (!sk || !slots || !seeds || !out || !nct) return fail(PIEHIP_EINVAL, "null operand");
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size exceeds the ring dimension");
    HIPCHK(hipSetDevice(h->device));
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M;
    const size_t LN = h->LN();
    const u64 t = h->hp.t;
    for (size_t i = 0; i < (size_t)nct * B; i++)
        if ((u64)(slots[i] < 0 ? -slots[i] : slots[i]) >= t) return fail(PIEHIP_EINVAL, "slot value out of range for the plaintext modulus");
    // host staging, not zero-filled (the sampling threads touch their own parts): a[nct][L][N] and e[nct][N]
    std::unique_ptr<u64[]> a_host(new u64[(size_t)nct * LN]);
    std::unique_ptr<int32_t[]> ev(new int32_t[(size_t)nct * N]);
    {
        // every ciphertext has its own seed and draws a (uniform), then e, as a sequential client would: the ciphertexts are
        // independent, so host threads share them out (2.4 M rejection-sampled words for the 29 ciphertexts of a C3 query)
        auto sample = [&](u32 c0, u32 c1) {
            for (u32 c = c0; c < c1; c++) {
                HostRng r(seeds[c]);
                sample_uniform(r, h->hp, &a_host[(size_t)c * LN]);
                sample_error(r, N, &ev[(size_t)c * N]);
            }
        };
        const u32 hw = std::thread::hardware_concurrency();
        const u32 nth = std::max(1u, std::min(std::min(nct, hw ? hw : 1u), 16u));
        std::vector<std::thread> pool;
        for (u32 i = 1; i < nth; i++) pool.emplace_back(sample, (u32)((u64)nct * i / nth), (u32)((u64)nct * (i + 1) / nth));
        sample(0, nct / nth);
        for (auto &th : pool) th.join();
    }
    const size_t ct_words = (size_t)nct * 2 * LN;
    Tmp tmp(h);
    TMPGET(d_out, ct_words);
    TMPGET(d_sk, LN);
    TMPGET(d_slotsw, (size_t)nct * B);
    TMPGET(d_u, (size_t)nct * N);
    TMPGET(d_em, (size_t)nct * LN);
    TMPGET(d_evw, ((size_t)nct * N + 1) / 2 + 1);
    TMPGET(d_a, (size_t)nct * LN);
    HIPCHK(hipMemcpy(d_a, a_host.get(), (size_t)nct * LN * sizeof(u64), hipMemcpyHostToDevice));  // enc_finish puts it into the c1 halves
    HIPCHK(hipMemcpy(d_sk, sk, LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_slotsw, slots, sizeof(int64_t) * (size_t)nct * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_evw, ev.get(), sizeof(int32_t) * (size_t)nct * N, hipMemcpyHostToDevice));
    launch_encode_scatter(h->d_dc, N, M, (const int64_t *)d_slotsw, B, h->d_inv_pos, d_u, nct, h->stream);
    launch_ntt(h->plan, d_u, nct, M, 1, true, h->stream);  // coefficients mod t
    launch_enc_message(h->d_dc, N, L, M, d_u, (const int32_t *)d_evw, d_em, nct, h->stream);
    launch_ntt(h->plan, d_em, nct * L, 0, L, false, h->stream);
    launch_enc_finish(h->d_dc, N, L, d_em, d_sk, d_a, d_out, nct, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d_out, ct_words * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_client_decrypt(piehip_handle h, const uint64_t *sk, const uint64_t *ct, uint32_t nct, uint32_t B, int64_t *slots)
{
    NEED(h);
    if (!sk || !ct || !slots || !nct) return fail(PIEHIP_EINVAL, "null operand");
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size exceeds the ring dimension");
    HIPCHK(hipSetDevice(h->device));
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M;
    const size_t LN = h->LN();
    Tmp tmp;
    TMPGET(d_ct, (size_t)nct * 2 * LN);
    TMPGET(d_sk, LN);
    TMPGET(d_x, (size_t)nct * LN);
    TMPGET(d_u, (size_t)nct * N);
    TMPGET(d_slotsw, (size_t)nct * B);
    TMPGET(d_posw, (N + 1) / 2 + 1);
    HIPCHK(hipMemcpy(d_ct, ct, (size_t)nct * 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_sk, sk, LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_posw, h->hp.slot_pos.data(), sizeof(u32) * N, hipMemcpyHostToDevice));
    launch_dec_dot(h->d_dc, N, L, d_ct, d_sk, d_x, nct, h->stream);
    launch_ntt(h->plan, d_x, nct * L, 0, L, true, h->stream);
    launch_dec_round(h->d_dc, N, L, M, d_x, d_u, nct, h->stream);
    launch_ntt(h->plan, d_u, nct, M, 1, false, h->stream);
    launch_decode_gather(h->d_dc, N, M, d_u, (const u32 *)d_posw, B, (int64_t *)d_slotsw, nct, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(slots, d_slotsw, sizeof(int64_t) * (size_t)nct * B, hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

}  // extern "C"
