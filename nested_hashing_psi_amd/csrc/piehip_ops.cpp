// piehip_ops.cpp -- the OpenFHE primitives under run(), one by one (host buffers in and out, synchronous: kernel-level parity
// tests), the NTT timing loop of the bench tooling and the per-kernel profile of a run().
#include "piehip_ctx.hpp"

using namespace piehip;

extern "C" {

int piehip_ntt(piehip_handle h, uint64_t *limbs, uint32_t nlimbs, uint32_t mod_base, uint32_t mod_count, int inverse)
{
    NEED(h);
    if (!limbs || !mod_count || mod_base + mod_count > h->hp.M + 1) return fail(PIEHIP_EINVAL, "bad modulus range");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const size_t words = (size_t)nlimbs * h->hp.N;
    TMPGET(d, words);
    HIPCHK(hipMemcpy(d, limbs, words * sizeof(u64), hipMemcpyHostToDevice));
    ntt(h, d, nlimbs, mod_base, mod_count, inverse != 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(limbs, d, words * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

static int ew_common(piehip_handle h, const uint64_t *x, const uint64_t *y, uint64_t *out, bool mul)
{
    NEED(h);
    if (!x || !y || !out) return fail(PIEHIP_EINVAL, "null operand");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const size_t LN = h->LN();
    TMPGET(dx, 2 * LN);
    TMPGET(dy, 2 * LN);
    TMPGET(dz, 2 * LN);
    HIPCHK(hipMemcpy(dx, x, 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy, y, (mul ? 1 : 2) * LN * sizeof(u64), hipMemcpyHostToDevice));
    if (mul)
        launch_ct_mul_plain(h->d_dc, h->hp.N, h->hp.L, dx, dy, 0, dz, 1, h->stream);
    else
        launch_ct_add(h->d_dc, h->hp.N, h->hp.L, dx, dy, dz, 1, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, dz, 2 * LN * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}
int piehip_eval_add(piehip_handle h, const uint64_t *x, const uint64_t *y, uint64_t *out) { return ew_common(h, x, y, out, false); }
int piehip_eval_mult_plain(piehip_handle h, const uint64_t *x, const uint64_t *pt, uint64_t *out)
{
    return ew_common(h, x, pt, out, true);
}

int piehip_eval_mult(piehip_handle h, const uint64_t *x, const uint64_t *y, uint32_t nct, int relin, uint64_t *out)
{
    NEED(h);
    if (!x || !y || !out || !nct) return fail(PIEHIP_EINVAL, "null operand");
    if (relin && !h->d_evk) return fail(PIEHIP_ESTATE, "relinearisation key not loaded");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const size_t LN = h->LN();
    const u32 L = h->hp.L;
    TMPGET(dxy, (size_t)nct * 4 * LN);  // [nct][x,y][2][L][N]
    TMPGET(dout, (size_t)nct * 3 * LN);
    for (u32 i = 0; i < nct; i++) {
        HIPCHK(hipMemcpy(dxy + (size_t)i * 4 * LN, x + (size_t)i * 2 * LN, 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dxy + (size_t)i * 4 * LN + 2 * LN, y + (size_t)i * 2 * LN, 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
    }
    MulWs w;
    int rc = ws_alloc(h, w, nct);
    if (rc) {
        ws_free(w);
        return rc;
    }
    const bool xq = xq_reuse(h);
    NttExtra ex;  // operand layout [nct][x, y][2][L]: x is "operand 0" of every pair
    ex.copy_out = w.eqp;
    ex.copy_K = 2;
    ex.copy_L = L;
    ex.copy_M = h->hp.M;
    ntt(h, dxy, nct * 4 * L, 0, L, true, false, true, xq ? &ex : nullptr);
    enqueue_mul(h, w, dxy, 4 * LN, dxy + 2 * LN, 4 * LN, nct, relin != 0, nullptr, dout, xq);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    ws_free(w);
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("eval_mult: ") + hipGetErrorString(e));
    HIPCHK(hipMemcpy(out, dout, (size_t)nct * (relin ? 2 : 3) * LN * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_eval_automorph(piehip_handle h, const uint64_t *x, uint32_t g, const uint64_t *rk, uint64_t *out)
{
    NEED(h);
    if (!x || !rk || !out) return fail(PIEHIP_EINVAL, "null operand");
    if (!(g & 1) || g >= 2 * h->hp.N) return fail(PIEHIP_EINVAL, "automorphism index must be odd and < 2N");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const size_t LN = h->LN();
    const u32 N = h->hp.N, L = h->hp.L;
    TMPGET(dx, 2 * LN);
    TMPGET(dk, (size_t)L * 2 * LN);
    TMPGET(dperm, 2 * LN);
    TMPGET(dout, 2 * LN);
    TMPGET(dmapw, (N + 1) / 2 + 1);
    u32 *dmap = (u32 *)dmapw;
    std::vector<u32> map = h->hp.automorph_map(g);
    HIPCHK(hipMemcpy(dx, x, 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dk, rk, (size_t)L * 2 * LN * sizeof(u64), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dmap, map.data(), sizeof(u32) * N, hipMemcpyHostToDevice));
    MulWs w;
    int rc = ws_alloc(h, w, 1);
    if (rc) {
        ws_free(w);
        return rc;
    }
    {
        ProfScope ps(h, PIEHIP_K_AUTOMORPH, 16.0 * N * 2 * L);
        launch_permute(N, dx, dmap, dperm, 2 * L, h->stream);
    }
    // (sigma(c0), 0) stays in EVALUATION format; sigma(c1) goes through the key switch
    (void)hipMemcpyAsync(w.d01, dperm, LN * sizeof(u64), hipMemcpyDeviceToDevice, h->stream);
    (void)hipMemsetAsync(w.d01 + LN, 0, LN * sizeof(u64), h->stream);
    (void)hipMemcpyAsync(w.d2c, dperm + LN, LN * sizeof(u64), hipMemcpyDeviceToDevice, h->stream);
    ntt(h, w.d2c, L, 0, L, true);
    enqueue_keyswitch(h, w, 1, dk, nullptr, dout);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    ws_free(w);
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("eval_automorph: ") + hipGetErrorString(e));
    HIPCHK(hipMemcpy(out, dout, 2 * LN * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_encode(piehip_handle h, const int64_t *slots, uint32_t npt, uint32_t B, uint64_t *out)
{
    NEED(h);
    if (!slots || !out || !npt) return fail(PIEHIP_EINVAL, "null operand");
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size exceeds the ring dimension");
    const u64 t = h->hp.t;
    for (size_t i = 0; i < (size_t)npt * B; i++) {
        const int64_t v = slots[i];
        if ((u64)(v < 0 ? -v : v) >= t) return fail(PIEHIP_EINVAL, "slot value out of range for the plaintext modulus");
    }
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    TMPGET(dsw, (size_t)npt * B);
    TMPGET(dout, (size_t)npt * h->LN());
    HIPCHK(hipMemcpy(dsw, slots, sizeof(int64_t) * (size_t)npt * B, hipMemcpyHostToDevice));
    int rc = encode_on_device(h, (const int64_t *)dsw, npt, B, dout);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out, dout, sizeof(u64) * (size_t)npt * h->LN(), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_base_convert(piehip_handle h, int which, const uint64_t *in, uint32_t npoly, uint64_t *out)
{
    NEED(h);
    if (!in || !out || !npoly || which < 0 || which > 2) return fail(PIEHIP_EINVAL, "bad argument");
    if (which == 2 && npoly % 3) return fail(PIEHIP_EINVAL, "scale-and-round takes polynomials in triples");
    if (which != 2 && npoly % 2) return fail(PIEHIP_EINVAL, "extension takes polynomials in pairs");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M;
    const size_t LN = h->LN(), MN = (size_t)M * N;
    const size_t win = (size_t)npoly * (which == 2 ? MN : LN), wout = (size_t)npoly * (which == 2 ? LN : MN);
    TMPGET(din, win);
    TMPGET(dout, wout);
    set_small_moduli(h->small_moduli);
    HIPCHK(hipMemcpy(din, in, win * sizeof(u64), hipMemcpyHostToDevice));
    if (which == 0)
        launch_expand_q_to_qp(h->d_dc, N, L, din, 2 * LN, LN, npoly / 2, dout, 2, 0, h->stream);
    else if (which == 1)
        launch_scale_pq_expand(h->d_dc, N, L, din, 2 * LN, LN, npoly / 2, dout, 2, 0, h->stream);
    else
        launch_scale_round(h->d_dc, N, L, din, npoly / 3, dout, 3 * LN, dout + 2 * LN, 3 * LN, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, dout, wout * sizeof(u64), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_bench_ntt(piehip_handle h, uint32_t nlimbs, uint32_t mod_count, int flags, uint32_t iters, double *ms_per_launch)
{
    NEED(h);
    if (!nlimbs || !mod_count || mod_count > h->hp.M || !iters || !ms_per_launch) return fail(PIEHIP_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp;
    const u32 N = h->hp.N;
    const size_t words = (size_t)nlimbs * N;
    TMPGET(d, words);
    {   // residues below the smallest modulus are valid for every limb
        std::vector<u64> host(words);
        u64 lo = h->hp.moduli[0];
        for (u32 a = 1; a < mod_count; a++) lo = h->hp.moduli[a] < lo ? h->hp.moduli[a] : lo;
        u64 s = 0x9E3779B97F4A7C15ULL;
        for (size_t i = 0; i < words; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            host[i] = s % lo;
        }
        HIPCHK(hipMemcpy(d, host.data(), words * sizeof(u64), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const bool inverse = (flags & 1) != 0, sigma = (flags & 2) != 0 && h->sigma_on;
    launch_ntt(h->plan, d, nlimbs, 0, mod_count, inverse, h->stream, sigma);  // warm-up
    HIPCHK(hipEventRecord(e0, h->stream));
    for (u32 i = 0; i < iters; i++) launch_ntt(h->plan, d, nlimbs, 0, mod_count, inverse, h->stream, sigma);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / iters;
    return PIEHIP_OK;
}

int piehip_set_profiling(piehip_handle h, int on)
{
    NEED_RO(h);
    h->profiling = on != 0;
    h->recs.clear();
    h->pool_used = 0;
    return PIEHIP_OK;
}

int piehip_profile_read_n(piehip_handle h, uint32_t n, uint32_t *launches, double *ms, double *alg_bytes)
{
    NEED_RO(h);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n > PIEHIP_NKERNELS) n = PIEHIP_NKERNELS;
    for (u32 k = 0; k < n; k++) {
        if (launches) launches[k] = 0;
        if (ms) ms[k] = 0;
        if (alg_bytes) alg_bytes[k] = 0;
    }
    for (const ProfRec &r : h->recs) {
        if ((u32)r.k >= n) continue;   // a class the caller's arrays have no room for
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
        if (launches) launches[r.k]++;
        if (ms) ms[r.k] += t;
        if (alg_bytes) alg_bytes[r.k] += r.bytes;
    }
    return PIEHIP_OK;
}

// the entry point of version 100: twelve classes, whatever this library knows beyond them
int piehip_profile_read(piehip_handle h, uint32_t *launches, double *ms, double *alg_bytes)
{
    return piehip_profile_read_n(h, PIEHIP_NKERNELS_V100, launches, ms, alg_bytes);
}

}  // extern "C"
