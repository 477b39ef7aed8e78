// piehip.cpp -- C ABI (include/piehip.h) over the gfx950 kernels: the context, keys and database of a handle, its query inputs,
// and the launch schedule of BatchedFHEHIPPIE::run() (reference BatchedFHEHIPPIE.cpp:88-129) on the handle's queues.
// The other entry points live in piehip_host.cpp / _ops.cpp / _fhepie.cpp / _client.cpp / _rccl.cpp (piehip_ctx.hpp lists them).
#include "piehip_ctx.hpp"

#include <random>

using namespace piehip;

static thread_local std::string g_err;
namespace piehip {
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
}  // namespace piehip
static void free_workspace(piehip_ctx *h);

namespace piehip {

void join_pending(piehip_ctx *h)
{
    if (!h->pending_join) return;
    for (size_t g = 0; g < h->ev_join.size(); g++) (void)hipStreamWaitEvent(h->stream, h->ev_join[g], 0);
    h->pending_join = false;
}

void mark_dirty(piehip_ctx *h) { h->inputs_dirty = true; }

void drop_graph(piehip_ctx *h)
{
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    h->gexec = nullptr;
}

}  // namespace piehip
static const char *KNAMES[PIEHIP_NKERNELS] = {"stage_a_mac", "ntt_fwd", "ntt_inv",  "expand",   "tensor",    "scale_round",
                                              "digits",      "relin",   "mask_mul", "encode",   "automorph", "other",
                                              "event_pair"};
namespace piehip {

// ---- profiling helpers -------------------------------------------------------------------------
hipEvent_t prof_event(piehip_ctx *h)
{
    if (h->pool_used == h->pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->pool.push_back(e);
    }
    return h->pool[h->pool_used++];
}

int dev_alloc(u64 **p, size_t words)
{
    *p = nullptr;
    if (!words) return PIEHIP_OK;
    hipError_t e = hipMalloc((void **)p, words * sizeof(u64));
    if (e != hipSuccess) return fail(PIEHIP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return PIEHIP_OK;
}
void dev_free(u64 **p)
{
    if (*p) (void)hipFree(*p);
    *p = nullptr;
}

int ws_alloc(piehip_ctx *h, MulWs &w, u32 nb)
{
    const size_t N = h->hp.N, L = h->hp.L, M = h->hp.M;
    w.nb = nb;
    int rc;
    if ((rc = dev_alloc(&w.eqp, (size_t)nb * 4 * M * N))) return rc;
    if ((rc = dev_alloc(&w.dqp, (size_t)nb * 3 * M * N))) return rc;
    if ((rc = dev_alloc(&w.d01, (size_t)nb * 2 * L * N))) return rc;
    if ((rc = dev_alloc(&w.d2c, (size_t)nb * L * N))) return rc;
    if ((rc = dev_alloc(&w.dig, (size_t)nb * L * L * N))) return rc;
    return PIEHIP_OK;
}
void ws_free(MulWs &w)
{
    dev_free(&w.eqp);
    dev_free(&w.dqp);
    dev_free(&w.d01);
    dev_free(&w.d2c);
    dev_free(&w.dig);
    w.nb = 0;
}

// ---- schedule pieces ----------------------------------------------------------------------------
// sigma: lane order on the EVALUATION side; fold: outer stage applied by the neighbouring kernels (both only
// take effect when the context supports them; callers pass the same flags to those neighbours)
void ntt(piehip_ctx *h, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inv, bool sigma, bool fold, const NttExtra *ex)
{
    ProfScope ps(h, inv ? PIEHIP_K_NTT_INV : PIEHIP_K_NTT_FWD, 16.0 * h->hp.N * nlimbs);
    launch_ntt(h->plan, data, nlimbs, mod_base, mod_count, inv, h->stream, sigma && h->sigma_on, fold && h->fold_on, ex);
}
// The X operand of a ciphertext multiplication is available in EVALUATION format before its inverse transform; when
// the register-blocked kernel runs that transform it also drops a lane-ordered copy into the Q limbs of the QP operand
// array, and the forward transform over QP skips those limbs (8 of 36 per bin layer at L = 4).  (For the first product of a
// query batch stage A has written X there already: enqueue_run_bins, x_direct.)
bool xq_reuse(const piehip_ctx *h) { return h->sigma_on && ntt_supports_extra(h->plan, h->fold_on); }

// BV key switch of the COEFFICIENT-format polynomials at w.d2c with `key`, added to the EVALUATION
// ciphertexts at w.d01, optionally multiplied by mask plaintexts: out[nb][2][L][N]
// sigma: w.d01 and the digits are in lane order, key/mask are lane-ordered copies, out is written in standard order
void enqueue_keyswitch(piehip_ctx *h, MulWs &w, u32 nb, const u64 *key, const u64 *mask, u64 *out, bool sigma, bool fold,
                       size_t key_stride, u32 key_group, bool out_is_result, bool digits_ready)
{
    const u32 N = h->hp.N, L = h->hp.L;
    const size_t LN = h->LN();
    const double W = 8.0 * N;
    set_small_moduli(h->small_moduli);
    bool fused = digits_ready;  // the caller's transform launch of d01 also lifted and transformed the digits
    // digit lift inside the transform's load phase (the 32-coefficient kernel; contexts whose lane order is the 16-coefficient
    // kernel's take the digits kernel + transform below)
    if (!fused && h->sigma_on && h->d_twc && h->hp.logN <= 14 && !(sigma && ntt16_applies(h->plan, fold && h->fold_on))) {
        ProfScope ps(h, PIEHIP_K_NTT_FWD, 16.0 * N * nb * L * L);
        fused = launch_ntt_digits(h->plan, w.d2c, w.dig, nb, L, sigma && h->sigma_on, fold && h->fold_on, h->stream);
    }
    if (!fused) {
        {
            ProfScope ps(h, PIEHIP_K_DIGITS, W * nb * (L + (double)L * L));
            launch_digits(h->d_dc, N, L, w.d2c, LN, nb, w.dig, h->stream, fold && h->fold_on);
        }
        ntt(h, w.dig, nb * L * L, 0, L, false, sigma, fold);
    }
    {
        // the result buffer may still be read by work the caller queued on the handle's stream before this run
        if (h->wait_before_results && out_is_result) (void)hipStreamWaitEvent(h->stream, h->wait_before_results, 0);
        if (h->chain_armed && out_is_result) {  // the next queue group of a host-results run may start (piehip_run_into)
            h->chain_armed = false;
            (void)hipEventRecord(h->ev_chain, h->stream);
        }
        ProfScope ps(h, PIEHIP_K_RELIN, W * (nb * ((double)L * L + 2 * L + 2 * L + (mask ? L : 0)) + 2.0 * L * L));
        launch_relin_mac(h->d_dc, N, L, w.d01, 2 * LN, w.dig, key, mask, out, nb, h->stream,
                         (sigma && h->sigma_on) ? h->d_sigma_inv : nullptr, key_stride, key_group,
                         (sigma && h->sigma_on) ? h->sigma_T : 0, h->sigma_kp, mask ? h->mask_div : 1);
    }
}

// One batched EvalMult(ct,ct) (BatchedFHEHIPPIE.cpp:123): operands in COEFFICIENT format (produced by
// ntt(.., inverse, sigma = false, fold = true): with folding on, their outermost inverse stage is applied here),
// X polynomial (o,c) at x + o*sx + c*LN, Y likewise.  relin: out[nb][2][L][N] (times mask if given);
// otherwise out[nb][3][L][N] holds the EVALUATION-format tensor result.
void enqueue_mul(piehip_ctx *h, MulWs &w, const u64 *x, size_t sx, const u64 *y, size_t sy, u32 nb, bool relin,
                 const u64 *mask, u64 *out, bool xq_ready, bool out_is_result)
{
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M;
    const size_t LN = h->LN();
    const double W = 8.0 * N;
    set_small_moduli(h->small_moduli);
    {
        ProfScope ps(h, PIEHIP_K_EXPAND, W * nb * (4.0 * L + 4.0 * M));
        launch_expand_both(h->d_dc, N, L, x, sx, y, sy, LN, nb, w.eqp, h->stream, h->fold_on, xq_ready);
    }
    // the QP operands and the tensor result never leave the library: lane order, no LDS transposes
    {
        NttExtra ex;
        ex.lazy_out = true;  // the tensor product's Barrett reduction takes any operands < 2^63
        if (xq_ready) {
            ex.skip_L = L;
            ex.skip_M = M;
        }
        ntt(h, w.eqp, nb * (xq_ready ? 4 * M - 2 * L : 4 * M), 0, M, false, true, true, &ex);
    }
    {
        ProfScope ps(h, PIEHIP_K_TENSOR, W * nb * 7.0 * M);
        launch_tensor(h->d_dc, N, M, w.eqp, w.dqp, nb, h->stream);
    }
    ntt(h, w.dqp, nb * 3 * M, 0, M, true, true, true);
    if (relin) {
        {
            ProfScope ps(h, PIEHIP_K_SCALE, W * nb * (3.0 * M + 3.0 * L));
            launch_scale_round(h->d_dc, N, L, w.dqp, nb, w.d01, 2 * LN, w.d2c, LN, h->stream, h->fold_on, false);
        }
        bool digits_ready = false;
        {
            NttExtra ex;
            ex.lazy_out = true;  // the key-switch MAC adds d01 into its accumulator before reducing
            if (h->sigma_on && h->small_moduli && ntt16_applies(h->plan, h->fold_on)) {
                // one launch for both forward transforms in front of the key-switch MAC: d0, d1 and the L * L digits of d2
                // (equal-width primes only: the kernel's lift is a conditional subtraction)
                ProfScope ps(h, PIEHIP_K_NTT_FWD, 16.0 * N * nb * (2.0 * L + (double)L * L));
                Ntt16Digits dg = {w.d2c, LN, w.dig, nb, L};
                digits_ready = launch_ntt16(h->plan, h->fold_on, w.d01, nb * 2 * L, 0, L, false, true, h->stream, &ex, &dg);
            }
            if (!digits_ready) ntt(h, w.d01, nb * 2 * L, 0, L, false, true, true, &ex);
        }
        if (h->key_group > 1)  // a batch whose queries bring their own keys (piehip_load_relin_key_q)
            enqueue_keyswitch(h, w, nb, h->sigma_on ? h->d_evkq_sigma : h->d_evkq, mask, out, true, true, (size_t)L * 2 * LN, h->key_group,
                              out_is_result, digits_ready);
        else
            enqueue_keyswitch(h, w, nb, h->sigma_on ? h->d_evk_sigma : h->d_evk, mask, out, true, true, 0, 1, out_is_result, digits_ready);
    } else {
        {
            ProfScope ps(h, PIEHIP_K_SCALE, W * nb * (3.0 * M + 3.0 * L));
            launch_scale_round(h->d_dc, N, L, w.dqp, nb, out, 3 * LN, out + 2 * LN, 3 * LN, h->stream, h->fold_on, true);
        }
        ntt(h, out, nb * 3 * L, 0, L, false, false, true);
    }
}

}  // namespace piehip

// =================================================================================================
extern "C" {

int piehip_version(void) { return 101; }   // 101: piehip_profile_read_n, piehip_set_transform_slots, piehip_upload_turn_wait, piehip_rccl_abort
const char *piehip_last_error(void) { return g_err.c_str(); }
const char *piehip_kernel_name(int k) { return (k >= 0 && k < PIEHIP_NKERNELS) ? KNAMES[k] : "?"; }

int piehip_default_moduli(uint32_t N, uint32_t L, uint64_t *q, uint64_t *p)
{
    if (!q || !p || L < 1 || L > MAX_L || N < 8 || (N & (N - 1))) return fail(PIEHIP_EINVAL, "bad N/L");
    std::vector<u64> ch(2 * L + 1);
    if (!prime_chain(N, 1ULL << 60, 2 * L + 1, ch.data())) return fail(PIEHIP_EINVAL, "prime chain exhausted");
    memcpy(q, ch.data(), sizeof(u64) * L);
    memcpy(p, ch.data() + L, sizeof(u64) * (L + 1));
    return PIEHIP_OK;
}

int piehip_create(piehip_handle *out, uint32_t N, uint32_t L, uint64_t t, const uint64_t *q, const uint64_t *p, int device,
                  void *stream)
{
    if (!out) return fail(PIEHIP_EINVAL, "null out");
    *out = nullptr;
    piehip_ctx *h = new piehip_ctx();
    std::string err = h->hp.init(N, L, t, q, p);
    if (!err.empty()) {
        delete h;
        return fail(PIEHIP_EINVAL, err);
    }
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev <= 0) {
        delete h;
        return fail(PIEHIP_EHIP, std::string("no HIP device visible (hipGetDeviceCount: ") + hipGetErrorString(de) + ", " +
                                     std::to_string(ndev) + " devices): libpiehip has no CPU fallback");
    }
    h->device = device;
#define CHK_(expr)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            piehip_destroy(h);                                                              \
            return fail(PIEHIP_EHIP, m_);                                                   \
        }                                                                                   \
    } while (0)
    CHK_(hipSetDevice(device));
    if (stream) {
        h->stream = (hipStream_t)stream;
    } else {
        CHK_(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    CHK_(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    const u32 M = h->hp.M;
    CHK_(hipMalloc((void **)&h->d_dc, sizeof(DevConsts)));
    CHK_(hipMemcpy(h->d_dc, &h->hp.dc, sizeof(DevConsts), hipMemcpyHostToDevice));
    CHK_(hipMalloc((void **)&h->d_tables, sizeof(u64) * (size_t)(M + 1) * 4 * N));
    for (u32 a = 0; a <= M; a++) {
        u64 *base = h->d_tables + (size_t)a * 4 * N;
        CHK_(hipMemcpy(base, h->hp.tw[a].data(), sizeof(u64) * N, hipMemcpyHostToDevice));
        CHK_(hipMemcpy(base + N, h->hp.tw_sh[a].data(), sizeof(u64) * N, hipMemcpyHostToDevice));
        CHK_(hipMemcpy(base + 2 * (size_t)N, h->hp.itw[a].data(), sizeof(u64) * N, hipMemcpyHostToDevice));
        CHK_(hipMemcpy(base + 3 * (size_t)N, h->hp.itw_sh[a].data(), sizeof(u64) * N, hipMemcpyHostToDevice));
    }
    {
        std::vector<u64> pairs((size_t)(M + 1) * 4 * N);
        for (u32 a = 0; a <= M; a++)
            for (u32 k = 0; k < N; k++) {
                u64 *f = &pairs[((size_t)a * 2 + 0) * 2 * N + 2 * (size_t)k];
                u64 *i = &pairs[((size_t)a * 2 + 1) * 2 * N + 2 * (size_t)k];
                // the register-blocked kernel uses 63-bit Shoup constants floor(w 2^63 / q) (kernels_ntt_fast.hip)
                f[0] = h->hp.tw[a][k];
                f[1] = h->hp.tw_sh[a][k] >> 1;
                i[0] = h->hp.itw[a][k];
                i[1] = h->hp.itw_sh[a][k] >> 1;
            }
        CHK_(hipMalloc((void **)&h->d_twp, pairs.size() * sizeof(u64)));
        CHK_(hipMemcpy(h->d_twp, pairs.data(), pairs.size() * sizeof(u64), hipMemcpyHostToDevice));
        u32 s0 = ntt_fast_s0(h->hp.logN);
        for (u32 a = 0; a <= M; a++)
            if (h->hp.moduli[a] >> 60) s0 = ~0u;  // lazy residues need 8q < 2^63
        if (s0 != ~0u) {
            std::vector<u64> all, one;
            for (u32 a = 0; a <= M; a++)
                for (u32 dir = 0; dir < 2; dir++) {
                    build_twc_table(&pairs[((size_t)a * 2 + dir) * 2 * N], h->hp.logN, s0, one);
                    all.insert(all.end(), one.begin(), one.end());
                }
            CHK_(hipMalloc((void **)&h->d_twc, all.size() * sizeof(u64)));
            CHK_(hipMemcpy(h->d_twc, all.data(), all.size() * sizeof(u64), hipMemcpyHostToDevice));
            if (h->hp.logN >= 14 && h->hp.logN <= 15) {  // folded configuration: slices of N/2
                all.clear();
                for (u32 a = 0; a <= M; a++)
                    for (u32 dir = 0; dir < 2; dir++) {
                        build_twc_table(&pairs[((size_t)a * 2 + dir) * 2 * N], h->hp.logN, 1, one);
                        all.insert(all.end(), one.begin(), one.end());
                    }
                CHK_(hipMalloc((void **)&h->d_twc_fold, all.size() * sizeof(u64)));
                CHK_(hipMemcpy(h->d_twc_fold, all.data(), all.size() * sizeof(u64), hipMemcpyHostToDevice));
                h->fold_on = true;
            }
            if (h->hp.logN >= 13 && h->hp.logN <= 15) {  // the 16-coefficients-per-thread kernel: slices of 2^13 (rings 2^13, 2^14) or 2^14 (ring 2^15)
                const u32 s16 = h->hp.logN == 13 ? 0u : 1u;
                all.clear();
                for (u32 a = 0; a <= M; a++)
                    for (u32 dir = 0; dir < 2; dir++) {
                        build_twk16_table(&pairs[((size_t)a * 2 + dir) * 2 * N], s16, h->hp.logN - s16, one);
                        all.insert(all.end(), one.begin(), one.end());
                    }
                CHK_(hipMalloc((void **)&h->d_twk16, all.size() * sizeof(u64)));
                CHK_(hipMemcpy(h->d_twk16, all.data(), all.size() * sizeof(u64), hipMemcpyHostToDevice));
            }
        }
    }
    {
        std::vector<u32> inv(N, 0xFFFFFFFFu);
        for (u32 s = 0; s < N; s++) inv[h->hp.slot_pos[s]] = s;
        CHK_(hipMalloc((void **)&h->d_inv_pos, sizeof(u32) * N));
        CHK_(hipMemcpy(h->d_inv_pos, inv.data(), sizeof(u32) * N, hipMemcpyHostToDevice));
    }
#undef CHK_
    h->plan.tables = h->d_tables;
    h->plan.twp = h->d_twp;
    h->plan.logN = h->hp.logN;
    h->plan.force_generic = false;
    if (h->hp.logN == 13) h->plan.twk16 = h->d_twk16;
    if (h->hp.logN == 14 || h->hp.logN == 15) h->plan.twk16_fold = h->d_twk16;
    const bool use16 = ntt16_applies(h->plan, h->fold_on);  // which kernel defines this context's lane order
    {
        std::vector<u32> smap;
        if (use16)
            ntt16_sigma_inverse_map(h->hp.logN, h->fold_on ? 1u : 0u, smap);
        else
            ntt_sigma_inverse_map(h->hp.logN, h->d_twc ? (h->fold_on ? 1u : ntt_fast_s0(h->hp.logN)) : ~0u, smap);
        if (hipMalloc((void **)&h->d_sigma_inv, sizeof(u32) * N) != hipSuccess ||
            hipMemcpy(h->d_sigma_inv, smap.data(), sizeof(u32) * N, hipMemcpyHostToDevice) != hipSuccess) {
            piehip_destroy(h);
            return fail(PIEHIP_EHIP, "sigma map upload failed");
        }
        h->sigma_on = h->d_twc != nullptr;
        if (h->sigma_on) {
            const u32 s0 = h->fold_on ? 1u : ntt_fast_s0(h->hp.logN);
            h->sigma_T = use16 ? (N >> s0) / 16 : (N >> s0) / 32;
            h->sigma_kp = use16 ? 8 : 16;
        }
        h->small_moduli = true;
        for (u32 a = 0; a < M; a++)
            if ((h->hp.moduli[a] >> 59) != 1) h->small_moduli = false;  // the mad paths assume 2^59 < q < 2^60
    }
    h->plan.twp = h->d_twp;
    h->plan.twc = h->d_twc;
    h->plan.twc_fold = h->d_twc_fold;
    h->plan.force_generic = false;
    {
        hipDeviceProp_t prop;
        h->plan.num_cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                              ? (u32)prop.multiProcessorCount : 256u;
    }
    h->plan.dc = h->d_dc;
    h->plan.N = N;
    h->plan.logN = h->hp.logN;
    *out = h;
    return PIEHIP_OK;
}

// a handle that borrowed its database and key (piehip_attach_database) lets go of them: pointers only
static void detach_database(piehip_ctx *h)
{
    if (!h->db_borrowed) return;
    h->d_evk = h->d_evk_sigma = h->d_db = h->d_masks = h->d_masks_sigma = nullptr;
    h->db_borrowed = false;
    if (h->db_owner && h->db_owner->db_borrowers) h->db_owner->db_borrowers--;
    h->db_owner = nullptr;
}

int piehip_destroy(piehip_handle h)
{
    if (!h) return PIEHIP_OK;
    if (h->db_borrowers) return fail(PIEHIP_ESTATE, "destroy: other handles still use this handle's database (destroy them first)");
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)piehip_rccl_destroy(h);
    for (hipEvent_t e : h->pool) (void)hipEventDestroy(e);
    detach_database(h);
    dev_free(&h->d_evk);
    dev_free(&h->d_db);
    dev_free(&h->d_masks);
    dev_free(&h->d_idx_own);
    dev_free(&h->d_minus_own);
    for (u32 q = 0; q < STAGE_A_MAX_QUERIES; q++) {
        dev_free(&h->bq_idx_own[q]);
        dev_free(&h->bq_minus_own[q]);
    }
    free_workspace(h);
    if (h->d_dc) (void)hipFree(h->d_dc);
    if (h->d_tables) (void)hipFree(h->d_tables);
    if (h->d_twp) (void)hipFree(h->d_twp);
    if (h->d_twc) (void)hipFree(h->d_twc);
    if (h->d_twc_fold) (void)hipFree(h->d_twc_fold);
    if (h->d_twk16) (void)hipFree(h->d_twk16);
    if (h->d_inv_pos) (void)hipFree(h->d_inv_pos);
    if (h->d_sigma_inv) (void)hipFree(h->d_sigma_inv);
    dev_free(&h->d_evk_sigma);
    dev_free(&h->d_evkq);
    dev_free(&h->d_evkq_sigma);
    dev_free(&h->d_masks_sigma);
    dev_free(&h->d_hash_tbl);
    dev_free(&h->arena);
    for (auto &kv : h->rotkeys) (void)hipFree(kv.second);
    for (auto &kv : h->rotmaps) (void)hipFree(kv.second);
    dev_free(&h->fp_pt);
    dev_free(&h->fp_mask);
    dev_free(&h->fp_e0);
    dev_free(&h->fp_idx);
    dev_free(&h->fp_out);
    dev_free(&h->fp_negkeys);
    if (h->fp_negmaps) (void)hipFree(h->fp_negmaps);
    for (hipStream_t s : h->side_streams) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    drop_graph(h);
    free_host_path(h);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_chain) (void)hipEventDestroy(h->ev_chain);
    for (hipEvent_t e : h->ev_join) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PIEHIP_OK;
}

int piehip_get_moduli(piehip_handle h, uint64_t *out)
{
    NEED_RO(h);
    memcpy(out, h->hp.moduli.data(), sizeof(u64) * (h->hp.M + 1));
    return PIEHIP_OK;
}
int piehip_get_root(piehip_handle h, uint32_t mi, uint64_t *psi)
{
    NEED_RO(h);
    if (mi > h->hp.M) return fail(PIEHIP_EINVAL, "mod_index out of range");
    *psi = h->hp.psi[mi];
    return PIEHIP_OK;
}
int piehip_get_twiddles(piehip_handle h, uint32_t mi, uint64_t *fwd, uint64_t *inv)
{
    NEED_RO(h);
    if (mi > h->hp.M) return fail(PIEHIP_EINVAL, "mod_index out of range");
    if (fwd) memcpy(fwd, h->hp.tw[mi].data(), sizeof(u64) * h->hp.N);
    if (inv) memcpy(inv, h->hp.itw[mi].data(), sizeof(u64) * h->hp.N);
    return PIEHIP_OK;
}
int piehip_get_slot_positions(piehip_handle h, uint32_t *pos)
{
    NEED_RO(h);
    memcpy(pos, h->hp.slot_pos.data(), sizeof(u32) * h->hp.N);
    return PIEHIP_OK;
}

int piehip_load_relin_key(piehip_handle h, const uint64_t *evk)
{
    NEED(h);
    if (!evk) return fail(PIEHIP_EINVAL, "null evk");
    if (h->db_borrowed) return fail(PIEHIP_EINVAL, "this handle uses another handle's key and database (piehip_attach_database)");
    if (h->db_borrowers) return fail(PIEHIP_ESTATE, "the key cannot be reloaded while other handles are attached to this one");
    HIPCHK(hipSetDevice(h->device));
    const size_t words = (size_t)h->hp.L * 2 * h->LN();
    if (!h->d_evk) {
        int rc = dev_alloc(&h->d_evk, words);
        if (rc) return rc;
    }
    // on the handle's stream: NEED() has ordered it behind every run still in flight (a null-stream copy would not be)
    HIPCHK(hipMemcpyAsync(h->d_evk, evk, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->sigma_on) {  // lane-ordered copy for the key-switch MAC
        if (!h->d_evk_sigma) {
            int rc = dev_alloc(&h->d_evk_sigma, words);
            if (rc) return rc;
        }
        launch_permute(h->hp.N, h->d_evk, h->d_sigma_inv, h->d_evk_sigma, h->hp.L * 2 * h->hp.L, h->stream);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    // per-query key slots (piehip_load_relin_key_q) that no query loaded hold a COPY of the handle's key: they follow it
    for (u32 i = 0; i < h->evkq_n && h->d_evkq; i++) {
        if (h->evkq_loaded >> i & 1) continue;
        HIPCHK(hipMemcpyAsync(h->d_evkq + i * words, h->d_evk, words * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
        if (h->sigma_on && h->d_evkq_sigma)
            HIPCHK(hipMemcpyAsync(h->d_evkq_sigma + i * words, h->d_evk_sigma, words * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
    }
    if (h->d_evkq) HIPCHK(hipStreamSynchronize(h->stream));
    return PIEHIP_OK;
}

// lane-ordered copy of the mask plaintexts for the fused mask multiply of the last key switch
static int make_masks_sigma(piehip_ctx *h)
{
    if (!h->sigma_on) return PIEHIP_OK;
    if (!h->d_masks_sigma) {  // freed with the run buffers when the shape changes
        int rc = dev_alloc(&h->d_masks_sigma, (size_t)h->b * h->LN());
        if (rc) return rc;
    }
    launch_permute(h->hp.N, h->d_masks, h->d_sigma_inv, h->d_masks_sigma, h->b * h->hp.L, h->stream);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("mask permutation: ") + hipGetErrorString(e));
    return PIEHIP_OK;
}

// run() workspace and result rows for b bin layers x nq queries of K inner hash functions
static void free_workspace(piehip_ctx *h)
{
    dev_free(&h->d_acc);
    dev_free(&h->d_prod);
    dev_free(&h->d_out);
    ws_free(h->ws);
    h->ws_cap_rows = h->ws_cap_K = 0;
}
// Every array of the workspace is indexed by row first, so one sized for more rows serves fewer as it stands: a smaller batch
// keeps the larger allocation (a server that alternates between batch sizes would otherwise free and allocate 0.6 GiB per change,
// and what the allocator hands back after such churn is not what it handed out first: uploads into input buffers allocated later
// ran at half the link rate in some sequences of bench.py's legs).
static int alloc_workspace(piehip_ctx *h, u32 K, u32 b)
{
    const size_t LN = h->LN(), rows = (size_t)b * h->nq;
    if (h->d_acc && h->d_out && h->ws.eqp && h->ws_cap_K == K && rows <= h->ws_cap_rows && (K <= 2 || h->d_prod)) {
        h->ws.nb = (u32)rows;
        return PIEHIP_OK;
    }
    free_workspace(h);
    int rc;
    if ((rc = dev_alloc(&h->d_acc, rows * K * 2 * LN))) return rc;
    if ((rc = dev_alloc(&h->d_out, rows * 2 * LN))) return rc;
    if (K > 2 && (rc = dev_alloc(&h->d_prod, rows * 2 * LN))) return rc;
    if ((rc = ws_alloc(h, h->ws, (u32)rows))) return rc;
    h->ws_cap_rows = rows;
    h->ws_cap_K = K;
    return PIEHIP_OK;
}

static int alloc_run_buffers(piehip_ctx *h, u32 K, u32 b, u32 E, bool with_db = true)
{
    drop_graph(h);  // the captured launches hold the addresses and shapes of the buffers below
    if (h->db_borrowed && with_db) {  // a database of its own from here on; the key goes back too (load it again)
        detach_database(h);
        h->K = h->b = h->E = 0;
    }
    if (K < 1) return fail(PIEHIP_EINVAL, "at least one inner hash function");
    if (b < 1 || E < 1) return fail(PIEHIP_EINVAL, "Bin size needs to be at least of size one!");
    // the database and masks are shared with the attached query slots, whose streams are not ordered against this handle's:
    // rewriting them in place (same shape) would race with their runs, reallocating them would leave them dangling
    if (with_db && h->db_borrowers)
        return fail(PIEHIP_ESTATE, "a database cannot be loaded while other handles are attached to this one (destroy or re-home them first)");
    const size_t LN = h->LN();
    if (with_db && h->K == K && h->b == b && h->E == E && h->d_db && h->d_masks && h->d_acc && h->d_out && h->ws.nb == b * h->nq) {
        // same shape as the database being replaced (or reserved): keep the 0.5 GiB of buffers (hipFree + hipMalloc cost
        // ~10 ms); the inputs of the previous database are stale
        h->d_idx = nullptr;
        for (u32 q = 1; q < STAGE_A_MAX_QUERIES; q++) h->bq_idx[q] = nullptr;
        return PIEHIP_OK;
    }
    dev_free(&h->d_db);
    dev_free(&h->d_masks);
    dev_free(&h->d_masks_sigma);
    free_workspace(h);
    h->K = h->b = h->E = 0;
    int rc;
    if (with_db && (rc = dev_alloc(&h->d_db, (size_t)K * b * E * LN))) return rc;
    if (with_db && (rc = dev_alloc(&h->d_masks, (size_t)b * LN))) return rc;
    if ((rc = alloc_workspace(h, K, b))) return rc;
    h->K = K;
    h->b = b;
    h->E = E;
    // inputs depend on K,E: drop stale copies
    h->stage_open = false;
    dev_free(&h->d_idx_own);
    h->d_idx = nullptr;
    for (u32 q = 1; q < STAGE_A_MAX_QUERIES; q++) {
        dev_free(&h->bq_idx_own[q]);
        h->bq_idx[q] = nullptr;
    }
    return PIEHIP_OK;
}

int piehip_load_db(piehip_handle h, uint32_t K, uint32_t b, uint32_t E, const uint64_t *pts, const uint64_t *masks)
{
    NEED(h);
    if (!pts || !masks) return fail(PIEHIP_EINVAL, "null database");
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_run_buffers(h, K, b, E);
    if (rc) return rc;
    const size_t LN = h->LN();
    HIPCHK(hipMemcpyAsync(h->d_db, pts, sizeof(u64) * (size_t)K * b * E * LN, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_masks, masks, sizeof(u64) * (size_t)b * LN, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return make_masks_sigma(h);
}

// Another query slot on the same database: `h` takes `owner`'s relinearisation key, database and masks by reference (device
// pointers; nothing is copied) and gets a run() workspace of its own, so that run() calls on the two handles -- each on its
// own stream -- overlap.  One query's stage A is HBM-bound while another's transforms are ALU-bound: two handles with one
// queue each finish two queries 10 % sooner than one handle with two queues finishes them one after the other (DESIGN.md
// section 6).  The owner must outlive the borrower and must not reload its key or database while the borrower is in use.
int piehip_attach_database(piehip_handle h, piehip_handle owner)
{
    NEED(h);
    if (!owner || owner == h) return fail(PIEHIP_EINVAL, "attach_database: needs another handle");
    if (h->db_borrowers)  // its key and database are in use by the handles attached to it: nothing of them may be freed
        return fail(PIEHIP_ESTATE, "attach_database: other handles are attached to this handle's database (detach or destroy them first)");
    join_pending(owner);
    if (owner->db_borrowed) return fail(PIEHIP_EINVAL, "attach_database: the owner itself borrows its database");
    if (!owner->d_db || (!owner->d_evk && owner->K > 1)) return fail(PIEHIP_ESTATE, "attach_database: the owner has no key or no database yet");
    if (owner->device != h->device) return fail(PIEHIP_EINVAL, "attach_database: handles on different devices");
    if (owner->hp.N != h->hp.N || owner->hp.L != h->hp.L || owner->hp.t != h->hp.t || owner->hp.moduli != h->hp.moduli)
        return fail(PIEHIP_EINVAL, "attach_database: handles with different parameters");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(owner->stream));  // uploads of the owner's key / database have landed
    if (!h->db_borrowed) {
        dev_free(&h->d_evk);
        dev_free(&h->d_evk_sigma);
        dev_free(&h->d_db);
        dev_free(&h->d_masks);
        dev_free(&h->d_masks_sigma);
    }
    detach_database(h);
    h->K = h->b = h->E = 0;
    int rc = alloc_run_buffers(h, owner->K, owner->b, owner->E, false);
    if (rc) return rc;
    h->d_evk = owner->d_evk;
    h->d_evk_sigma = owner->d_evk_sigma;
    h->d_db = owner->d_db;
    h->d_masks = owner->d_masks;
    h->d_masks_sigma = owner->d_masks_sigma;
    h->db_borrowed = true;
    h->db_owner = owner;
    owner->db_borrowers++;
    return PIEHIP_OK;
}

static const u32 ENCODE_CHUNK = 256;  // plaintexts per batch of the device encoder (bounds its mod-t scratch)

// the persistent hash-table buffer [k][e][K][b][E]: reallocated only when the size changes
static int hash_tbl_alloc(piehip_ctx *h, size_t words)
{
    if (h->d_hash_tbl && h->hash_tbl_words == words) return PIEHIP_OK;
    dev_free(&h->d_hash_tbl);
    h->hash_tbl_words = 0;
    int rc = dev_alloc(&h->d_hash_tbl, words);
    if (rc) return rc;
    h->hash_tbl_words = words;
    return PIEHIP_OK;
}

// scratch words piehip_build_db_bins carves (256-byte granules), including the encoder's
static size_t build_db_scratch_words(const piehip_ctx *h, size_t n, u32 k, u32 e, u32 K, u32 b, u32 E)
{
    auto g = [](size_t w) { return ((w ? w : 1) + 31) & ~(size_t)31; };
    const size_t B = (size_t)k * e, npt = (size_t)K * b * E;
    return g((size_t)(k + K) * 16 * 256) + g(n) + 2 * g(n + 1) + g((e + 2) / 2 + 1) + g(1) + g(hash_sort_temp_bytes((u32)n, e) / 8 + 1) +
           g((npt > b ? npt : b) * B) + g((size_t)ENCODE_CHUNK * h->hp.N);
}

// device-side MakePackedPlaintext of npt slot vectors (already on the device) into out[npt][L][N]
}  // extern "C"
namespace piehip {
int encode_on_device(piehip_ctx *h, const int64_t *d_slots, u32 npt, u32 B, u64 *d_out)
{
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M;
    // chunk so the mod-t scratch stays small
    const u32 chunk = ENCODE_CHUNK;
    Tmp tmp(h);
    TMPGET(d_u, (size_t)(npt < chunk ? npt : chunk) * N);
    for (u32 s = 0; s < npt; s += chunk) {
        const u32 c = npt - s < chunk ? npt - s : chunk;
        ProfScope ps(h, PIEHIP_K_ENCODE, 8.0 * c * ((double)B + 2.0 * N + (double)L * N));
        launch_encode_scatter(h->d_dc, N, M, d_slots + (size_t)s * B, B, h->d_inv_pos, d_u, c, h->stream);
        launch_ntt(h->plan, d_u, c, M, 1, true, h->stream);
        launch_encode_lift(h->d_dc, N, L, M, d_u, d_out + (size_t)s * L * N, c, h->stream);
        launch_ntt(h->plan, d_out + (size_t)s * L * N, c * L, 0, L, false, h->stream);
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("encode: ") + hipGetErrorString(e));
    return PIEHIP_OK;
}
}  // namespace piehip
extern "C" {

int piehip_load_db_slots(piehip_handle h, uint32_t K, uint32_t b, uint32_t E, uint32_t B, const int64_t *slots,
                         const int64_t *mask_slots)
{
    NEED(h);
    if (!slots || !mask_slots) return fail(PIEHIP_EINVAL, "null database");
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size exceeds the ring dimension");
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_run_buffers(h, K, b, E);
    if (rc) return rc;
    const size_t npt = (size_t)K * b * E;
    int64_t *d_s = nullptr;
    HIPCHK(hipMalloc((void **)&d_s, sizeof(int64_t) * (npt > b ? npt : b) * B));
    hipError_t e = hipMemcpy(d_s, slots, sizeof(int64_t) * npt * B, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = encode_on_device(h, d_s, (u32)npt, B, h->d_db);
    if (e == hipSuccess && rc == PIEHIP_OK) e = hipMemcpy(d_s, mask_slots, sizeof(int64_t) * (size_t)b * B, hipMemcpyHostToDevice);
    if (e == hipSuccess && rc == PIEHIP_OK) rc = encode_on_device(h, d_s, b, B, h->d_masks);
    (void)hipFree(d_s);
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("load_db_slots: ") + hipGetErrorString(e));
    if (rc) return rc;
    return make_masks_sigma(h);
}

// Last step of the constructor (BatchedFHEHIPPIE.cpp:45-82) for the bin layers [lo, hi) this handle keeps: MakePackedPlaintext
// of the gathered slot vectors d_slots[K][b][E][B] into the database [K][hi - lo][E], then the masks of those layers (drawn per
// layer from mask_seed, so every shard of a sharded server holds the masks the unsharded one would).  Overwrites d_slots.
static int encode_bin_layers(piehip_ctx *h, int64_t *d_slots, u32 K, u32 b, u32 E, u32 B, u32 lo, u32 hi, u64 mask_seed)
{
    const u32 nb = hi - lo;
    const size_t LN = h->LN();
    int rc;
    for (u32 hf = 0; hf < K; hf++)
        if ((rc = encode_on_device(h, d_slots + ((size_t)hf * b + lo) * E * B, nb * E, B, h->d_db + (size_t)hf * nb * E * LN))) return rc;
    launch_mask_slots(h->hp.t, b, B, mask_seed, d_slots, h->stream);
    if ((rc = encode_on_device(h, d_slots + (size_t)lo * B, nb, B, h->d_masks))) return rc;
    return make_masks_sigma(h);
}

// TabulationHashing tables (TabulationHashing.cpp:16-36): [nfun][16][256], drawn in that order from
// std::mt19937(seed) through std::uniform_int_distribution<uint64_t> -- the library types themselves, so the
// stream is the reference's under libstdc++.
static void tabulation_tables(uint64_t seed, uint32_t nfun, std::vector<u64> &tab)
{
    tab.resize((size_t)nfun * 16 * 256);
    std::mt19937 gen(seed);
    std::uniform_int_distribution<uint64_t> dis;
    for (auto &v : tab) v = dis(gen);
}

int piehip_tabulation_hash(uint64_t hash_seed, uint32_t nfun, uint32_t hf, const uint64_t *x, size_t n, uint64_t *out)
{
    if (!x || !out || hf >= nfun) return fail(PIEHIP_EINVAL, "bad argument");
    std::vector<u64> tab;
    tabulation_tables(hash_seed, nfun, tab);
    const u64 *t = tab.data() + (size_t)hf * 16 * 256;
    for (size_t a = 0; a < n; a++) {
        u64 v = x[a], res = 0;
        for (int i = 0; i < 16; i++) {
            res ^= t[i * 256 + (v & 0xff)];
            v >>= 8;
        }
        out[a] = res;
    }
    return PIEHIP_OK;
}

// The client's table (host side, no device): CuckooHashTable(hash, e, k, startingHashId 0, stash 0, multi tables, 1 layer),
// insertAll (BatchedFHEPSIClient.cpp:97-99,109; insert / eviction walk at CuckooHashTable.cpp:72-114, 1000 retries).
int piehip_client_cuckoo_table(uint64_t hash_seed, uint32_t nfun, uint32_t k, uint32_t e, const uint64_t *items, size_t n,
                               uint64_t *table)
{
    if (!items || !table || !k || !e || k > nfun) return fail(PIEHIP_EINVAL, "bad argument");
    std::vector<u64> tab;
    tabulation_tables(hash_seed, nfun, tab);
    auto pos = [&](u64 x, u32 hf) -> size_t {
        const u64 *t = tab.data() + (size_t)hf * 16 * 256;
        u64 v = x, res = 0;
        for (int i = 0; i < 16; i++) {
            res ^= t[i * 256 + (v & 0xff)];
            v >>= 8;
        }
        return (size_t)(res % e);
    };
    std::fill(table, table + (size_t)k * e, 0);
    for (size_t a = 0; a < n; a++) {
        u64 x = items[a];
        bool dup = false;
        for (u32 hf = 0; hf < k; hf++) dup = dup || table[(size_t)hf * e + pos(x, hf)] == x;  // lookUp
        if (dup) continue;
        bool placed = false;
        for (int retry = 0; retry < 1000 && !placed; retry++) {  // numberOfRetries
            for (u32 hf = 0; hf < k; hf++) {
                u64 &slot = table[(size_t)hf * e + pos(x, hf)];
                if (slot == 0) {
                    slot = x;
                    placed = true;
                    break;
                }
                std::swap(x, slot);  // one layer: evict the occupant and carry it to the next table
            }
        }
        if (!placed) return fail(PIEHIP_EHASH, "(Blocked) Cuckoo hashing error");
    }
    return PIEHIP_OK;
}

int piehip_reserve(piehip_handle h, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, uint32_t bin_lo, uint32_t bin_hi)
{
    NEED(h);
    if (k < 1 || e < 1 || bin_lo >= bin_hi || bin_hi > b || !n || n > 0x7FFFFFFFu) return fail(PIEHIP_EINVAL, "bad shape");
    if (K < 2) return fail(PIEHIP_EINVAL, "Cuckoo Table needs more than one hash function!");  // CuckooHashTable.cpp:39-42
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_run_buffers(h, K, bin_hi - bin_lo, E);
    if (rc) return rc;
    if ((rc = hash_tbl_alloc(h, (size_t)k * e * K * b * E))) return rc;
    const size_t need = build_db_scratch_words(h, n, k, e, K, b, E);
    if (h->arena_words < need) {
        if (h->arena_used) return fail(PIEHIP_ESTATE, "scratch in use");
        dev_free(&h->arena);
        h->arena_words = 0;
        if ((rc = dev_alloc(&h->arena, need))) return rc;
        h->arena_words = need;
    }
    // the per-query input buffers too (setIndex / setMinusCompareElement from host memory)
    if (!h->d_idx_own && (rc = dev_alloc(&h->d_idx_own, (size_t)K * E * 2 * h->LN()))) return rc;
    if (!h->d_minus_own && (rc = dev_alloc(&h->d_minus_own, 2 * h->LN()))) return rc;
    return PIEHIP_OK;
}

int piehip_build_db_bins(piehip_handle h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                         uint64_t hash_seed, uint64_t evict_seed, uint64_t shuffle_seed, uint64_t mask_seed, uint32_t bin_lo,
                         uint32_t bin_hi)
{
    NEED(h);
    if (!items || !n || n > 0x7FFFFFFFu) return fail(PIEHIP_EINVAL, "empty or oversized server set");
    if (k < 1 || e < 1) return fail(PIEHIP_EINVAL, "need at least one outer hash function and position");
    if (K < 2) return fail(PIEHIP_EINVAL, "Cuckoo Table needs more than one hash function!");  // CuckooHashTable.cpp:39-42
    if (bin_lo >= bin_hi || bin_hi > b) return fail(PIEHIP_EINVAL, "bin-layer slice must be non-empty and within [0, b)");
    const size_t B = (size_t)k * e;
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size k*e exceeds the ring dimension");
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_run_buffers(h, K, bin_hi - bin_lo, E);
    if (rc) return rc;
    const size_t tbl_words = B * K * b * E;
    if ((rc = hash_tbl_alloc(h, tbl_words))) return rc;
    h->hk = k;
    h->he = e;
    h->hb = b;
    std::vector<u64> tab;
    tabulation_tables(hash_seed, k + K, tab);
    Tmp tmp(h);
    TMPGET(d_tab, tab.size());
    TMPGET(d_items, n);
    TMPGET(d_keys, n + 1);      // 2 n u32
    TMPGET(d_vals, n + 1);      // 2 n u32
    TMPGET(d_start, (e + 2) / 2 + 1);
    TMPGET(d_failw, 1);
    const size_t temp_bytes = hash_sort_temp_bytes((u32)n, e);
    TMPGET(d_temp, temp_bytes / 8 + 1);
    const size_t npt = (size_t)K * b * E;
    TMPGET(d_slotsw, (npt > b ? npt : b) * B);
    int64_t *d_slots = (int64_t *)d_slotsw;
    u32 *d_fail = (u32 *)d_failw;
    HIPCHK(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_items, items, n * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_hash_tbl, 0, tbl_words * sizeof(u64), h->stream));
    HIPCHK(hipMemsetAsync(d_fail, 0, sizeof(u32), h->stream));
    {
        ProfScope ps(h, PIEHIP_K_OTHER, 8.0 * (double)n * k * 4);
        HIPCHK(launch_hash_build(d_tab, d_items, (u32)n, k, e, K, b, E, evict_seed, shuffle_seed, h->d_hash_tbl, (u32 *)d_keys,
                                 (u32 *)d_vals, (u32 *)d_start, d_temp, temp_bytes, d_fail, h->stream));
        launch_gather_slots(h->d_hash_tbl, (u32)B, K, b, E, h->hp.t, d_slots, d_fail, h->stream);
    }
    u32 failed = 0;
    HIPCHK(hipMemcpyAsync(&failed, d_fail, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (failed & 1u) return fail(PIEHIP_EHASH, "(Blocked) Cuckoo hashing error");
    if (failed & 2u) return fail(PIEHIP_EINVAL, "server item does not fit the plaintext modulus");
    return encode_bin_layers(h, d_slots, K, b, E, (u32)B, bin_lo, bin_hi, mask_seed);
}

int piehip_build_db(piehip_handle h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                    uint64_t hash_seed, uint64_t evict_seed, uint64_t shuffle_seed, uint64_t mask_seed)
{
    return piehip_build_db_bins(h, items, n, k, e, K, b, E, hash_seed, evict_seed, shuffle_seed, mask_seed, 0, b);
}

int piehip_load_db_table_bins(piehip_handle h, const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                              uint64_t shuffle_seed, uint64_t mask_seed, uint32_t bin_lo, uint32_t bin_hi)
{
    NEED(h);
    if (!tbl || k < 1 || e < 1) return fail(PIEHIP_EINVAL, "bad hash table");
    if (K < 2) return fail(PIEHIP_EINVAL, "Cuckoo Table needs more than one hash function!");  // CuckooHashTable.cpp:39-42
    if (bin_lo >= bin_hi || bin_hi > b) return fail(PIEHIP_EINVAL, "bin-layer slice must be non-empty and within [0, b)");
    const size_t B = (size_t)k * e;
    if (B > h->hp.N) return fail(PIEHIP_EINVAL, "batch size k*e exceeds the ring dimension");
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_run_buffers(h, K, bin_hi - bin_lo, E);
    if (rc) return rc;
    const size_t tbl_words = B * K * b * E;
    if ((rc = hash_tbl_alloc(h, tbl_words))) return rc;
    h->hk = k;
    h->he = e;
    h->hb = b;
    Tmp tmp(h);
    const size_t npt = (size_t)K * b * E;
    TMPGET(d_slotsw, (npt > b ? npt : b) * B);
    TMPGET(d_failw, 1);
    int64_t *d_slots = (int64_t *)d_slotsw;
    u32 *d_fail = (u32 *)d_failw;
    HIPCHK(hipMemcpyAsync(h->d_hash_tbl, tbl, tbl_words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(d_fail, 0, sizeof(u32), h->stream));
    launch_shuffle_rows(h->d_hash_tbl, (u32)(B * K), b, E, shuffle_seed, h->stream);
    launch_gather_slots(h->d_hash_tbl, (u32)B, K, b, E, h->hp.t, d_slots, d_fail, h->stream);
    u32 failed = 0;
    HIPCHK(hipMemcpyAsync(&failed, d_fail, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (failed & 2u) return fail(PIEHIP_EINVAL, "server item does not fit the plaintext modulus");
    return encode_bin_layers(h, d_slots, K, b, E, (u32)B, bin_lo, bin_hi, mask_seed);
}

int piehip_load_db_table(piehip_handle h, const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                         uint64_t shuffle_seed, uint64_t mask_seed)
{
    return piehip_load_db_table_bins(h, tbl, k, e, K, b, E, shuffle_seed, mask_seed, 0, b);
}

int piehip_get_hash_table(piehip_handle h, uint64_t *tbl)
{
    NEED(h);
    if (!tbl) return fail(PIEHIP_EINVAL, "null out");
    if (!h->d_hash_tbl) return fail(PIEHIP_ESTATE, "no table: call piehip_build_db first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(tbl, h->d_hash_tbl, sizeof(u64) * (size_t)h->hk * h->he * h->K * h->hb * h->E, hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

int piehip_set_index(piehip_handle h, const uint64_t *idx)
{
    NEED(h);
    if (!idx) return fail(PIEHIP_EINVAL, "null index matrix");
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database before the index matrix");
    HIPCHK(hipSetDevice(h->device));
    const size_t words = (size_t)h->K * h->E * 2 * h->LN();
    if (!h->d_idx_own) {
        int rc = dev_alloc(&h->d_idx_own, words);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(h->d_idx_own, idx, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->d_idx = h->d_idx_own;
    return PIEHIP_OK;
}
int piehip_set_minus(piehip_handle h, const uint64_t *minus)
{
    NEED(h);
    if (!minus) return fail(PIEHIP_EINVAL, "null minus element");
    HIPCHK(hipSetDevice(h->device));
    const size_t words = 2 * h->LN();
    if (!h->d_minus_own) {
        int rc = dev_alloc(&h->d_minus_own, words);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(h->d_minus_own, minus, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->d_minus = h->d_minus_own;
    return PIEHIP_OK;
}
int piehip_set_index_device(piehip_handle h, const void *d_idx)
{
    NEED(h);
    if (!d_idx) return fail(PIEHIP_EINVAL, "null index matrix");
    h->d_idx = (const u64 *)d_idx;
    return PIEHIP_OK;
}
int piehip_set_minus_device(piehip_handle h, const void *d_minus)
{
    NEED(h);
    if (!d_minus) return fail(PIEHIP_EINVAL, "null minus element");
    h->d_minus = (const u64 *)d_minus;
    return PIEHIP_OK;
}

// ---- query batches --------------------------------------------------------------------------------------------------------
// run() over nq queries at once (each with its own index matrix and minus element) against the handle's database.  Stage A
// reads every database plaintext once for the batch instead of once per query, and every later launch works on nq times as
// many ciphertexts.  Rows of the workspace and of the results: [bin layer][query].
int piehip_set_query_batch(piehip_handle h, uint32_t nq)
{
    NEED(h);
    if (nq < 1 || nq > STAGE_A_MAX_QUERIES) return fail(PIEHIP_EINVAL, "set_query_batch: between 1 and 8 queries per run()");
    if (nq == h->nq) return PIEHIP_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->stage_open = false;
    // caller-owned device pointers of queries outside the new batch are forgotten (they may be freed by now); a later, larger
    // batch must set them again.  Per-query keys are sized by the batch: load them again after a change.
    for (u32 q = nq; q < STAGE_A_MAX_QUERIES; q++) h->bq_idx[q] = h->bq_minus[q] = nullptr;
    dev_free(&h->d_evkq);
    dev_free(&h->d_evkq_sigma);
    h->evkq_n = h->evkq_loaded = 0;
    h->nq = nq;
    if (!h->K) return PIEHIP_OK;  // the database's arrival sizes the workspace
    return alloc_workspace(h, h->K, h->b);
}
int piehip_get_query_batch(piehip_handle h, uint32_t *nq)
{
    NEED_RO(h);
    if (!nq) return fail(PIEHIP_EINVAL, "null out");
    *nq = h->nq;
    return PIEHIP_OK;
}
static int batch_query_check(piehip_ctx *h, u32 q)
{
    if (q >= h->nq) return fail(PIEHIP_EINVAL, "query index outside the batch (piehip_set_query_batch)");
    return PIEHIP_OK;
}
int piehip_set_index_device_q(piehip_handle h, uint32_t q, const void *d_idx)
{
    if (q == 0) return piehip_set_index_device(h, d_idx);
    NEED(h);
    if (!d_idx) return fail(PIEHIP_EINVAL, "null index matrix");
    int rc = batch_query_check(h, q);
    if (rc) return rc;
    h->bq_idx[q] = (const u64 *)d_idx;
    return PIEHIP_OK;
}
int piehip_set_minus_device_q(piehip_handle h, uint32_t q, const void *d_minus)
{
    if (q == 0) return piehip_set_minus_device(h, d_minus);
    NEED(h);
    if (!d_minus) return fail(PIEHIP_EINVAL, "null minus element");
    int rc = batch_query_check(h, q);
    if (rc) return rc;
    h->bq_minus[q] = (const u64 *)d_minus;
    return PIEHIP_OK;
}
int piehip_set_index_q(piehip_handle h, uint32_t q, const uint64_t *idx)
{
    if (q == 0) return piehip_set_index(h, idx);
    NEED(h);
    if (!idx) return fail(PIEHIP_EINVAL, "null index matrix");
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database before the index matrix");
    int rc = batch_query_check(h, q);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const size_t words = (size_t)h->K * h->E * 2 * h->LN();
    if (!h->bq_idx_own[q] && (rc = dev_alloc(&h->bq_idx_own[q], words))) return rc;
    HIPCHK(hipMemcpyAsync(h->bq_idx_own[q], idx, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->bq_idx[q] = h->bq_idx_own[q];
    return PIEHIP_OK;
}
int piehip_set_minus_q(piehip_handle h, uint32_t q, const uint64_t *minus)
{
    if (q == 0) return piehip_set_minus(h, minus);
    NEED(h);
    if (!minus) return fail(PIEHIP_EINVAL, "null minus element");
    int rc = batch_query_check(h, q);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const size_t words = 2 * h->LN();
    if (!h->bq_minus_own[q] && (rc = dev_alloc(&h->bq_minus_own[q], words))) return rc;
    HIPCHK(hipMemcpyAsync(h->bq_minus_own[q], minus, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->bq_minus[q] = h->bq_minus_own[q];
    return PIEHIP_OK;
}

}  // extern "C"
namespace piehip {
int query_input_buffers(piehip_ctx *h, u32 q, u64 **d_idx, u64 **d_minus)
{
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database before the index matrix");
    if (q >= h->nq) return fail(PIEHIP_EINVAL, "query index outside the batch (piehip_set_query_batch)");
    u64 **pi = q ? &h->bq_idx_own[q] : &h->d_idx_own, **pm = q ? &h->bq_minus_own[q] : &h->d_minus_own;
    int rc;
    if (!*pi && (rc = dev_alloc(pi, (size_t)h->K * h->E * 2 * h->LN()))) return rc;
    if (!*pm && (rc = dev_alloc(pm, 2 * h->LN()))) return rc;
    *d_idx = *pi;
    *d_minus = *pm;
    return PIEHIP_OK;
}
}  // namespace piehip
extern "C" {

// The queries of a batch come from different clients (BatchedFHEPSIServer.cpp:94-95: one client per connection), and every client
// has its own EvalMult key (.cpp:45-49): query q's key switch takes key q.  The key-switch kernel already selects its key per
// ciphertext row (FHEHIPPIE's EvalMerge: one rotation key per position); rows of a batch are [bin layer][query], so row r
// takes key r % nq.
int piehip_load_relin_key_q(piehip_handle h, uint32_t q, const uint64_t *evk)
{
    NEED(h);
    if (!evk) return fail(PIEHIP_EINVAL, "null evk");
    if (q >= h->nq) return fail(PIEHIP_EINVAL, "query index outside the batch (piehip_set_query_batch)");
    HIPCHK(hipSetDevice(h->device));
    const size_t words = (size_t)h->hp.L * 2 * h->LN();
    int rc;
    if (!h->d_evkq || h->evkq_n != h->nq) {
        dev_free(&h->d_evkq);
        dev_free(&h->d_evkq_sigma);
        h->evkq_n = h->evkq_loaded = 0;
        if ((rc = dev_alloc(&h->d_evkq, words * h->nq))) return rc;
        if (h->sigma_on && (rc = dev_alloc(&h->d_evkq_sigma, words * h->nq))) return rc;
        h->evkq_n = h->nq;
        // queries without a key of their own use the handle's (piehip_load_relin_key), if it has one
        for (u32 i = 0; i < h->nq && h->d_evk; i++) {
            HIPCHK(hipMemcpyAsync(h->d_evkq + i * words, h->d_evk, words * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
            if (h->sigma_on)
                HIPCHK(hipMemcpyAsync(h->d_evkq_sigma + i * words, h->d_evk_sigma, words * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    HIPCHK(hipMemcpyAsync(h->d_evkq + q * words, evk, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    if (h->sigma_on) launch_permute(h->hp.N, h->d_evkq + q * words, h->d_sigma_inv, h->d_evkq_sigma + q * words, h->hp.L * 2 * h->hp.L, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->evkq_loaded |= 1u << q;
    return PIEHIP_OK;
}

// Queues of a run().  The default is two when the handle evaluates enough bin layers to fill the chip twice over; below that
// every launch is bound by its own latency, a second queue only interleaves two latency-bound chains on the same CUs, and one
// queue is faster (measured at the C3 ring: 2 layers 115 vs 146 us, 5 layers of the E = 40 row 217 vs 239 us, 7 layers even).
static const u32 MAX_RUN_QUEUES = 2;  // 3 is equal within noise, 4 and more collapse
}  // extern "C"
namespace piehip {
u32 run_queue_count(const piehip_ctx *h)
{
    const u32 want = h->run_streams ? h->run_streams : (h->b >= 8 ? 2u : 1u);
    return std::min(want, std::min(MAX_RUN_QUEUES, h->b));
}
}  // namespace piehip
extern "C" {
// The queues are created when a run first needs them: a handle that runs on one queue (a query slot, a rank's small share)
// then owns one stream, not three -- the runtime multiplexes streams onto a few hardware queues, and streams that share one
// serialise against each other.
}  // extern "C"
namespace piehip {
int ensure_run_queues(piehip_ctx *h, u32 ng)
{
    while (h->side_streams.size() < ng) {
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        h->side_streams.push_back(s);
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ev_join.push_back(e);
    }
    return PIEHIP_OK;
}
}  // namespace piehip
extern "C" {

// bin layers of queue group g of ng.  Two groups take 4/7 and 3/7 of the layers: measured 3.5 % faster than equal halves at
// b = 14 (8 + 6: the ragged transform launches of the two queues fit the workgroup slots better than 7 + 7).
}  // extern "C"
namespace piehip {
u32 run_group_size(u32 b, u32 ng, u32 g)
{
    if (ng == 2) {
        const u32 first = (4 * b + 3) / 7;
        return g == 0 ? first : b - first;
    }
    return b / ng + (g < b % ng ? 1 : 0);
}
}  // namespace piehip
extern "C" {

// Bin layers [b0, b0 + nb) of run() on the handle's current stream: stage A, then the product chain.
static void enqueue_run_bins(piehip_ctx *h, u32 b0, u32 nb, u64 *results)
{
    const u32 N = h->hp.N, L = h->hp.L, M = h->hp.M, K = h->K, b = h->b, E = h->E, nq = h->nq;
    const size_t LN = h->LN();
    const double W = 8.0 * N;
    // rows of the workspace: (bin layer, query) pairs, nq per layer; the product chain sees nb * nq ciphertexts
    const size_t r0 = (size_t)b0 * nq;
    const u32 layers = nb;
    nb *= nq;
    MulWs w = h->ws;  // view of the workspace rows of these bin layers
    w.nb = nb;
    w.eqp += r0 * 4 * M * N;
    w.dqp += r0 * 3 * M * N;
    w.d01 += r0 * 2 * LN;
    w.d2c += r0 * LN;
    w.dig += r0 * L * LN;
    u64 *acc = h->d_acc + r0 * K * 2 * LN;
    u64 *prod = h->d_prod ? h->d_prod + r0 * 2 * LN : nullptr;
    u64 *out = results + r0 * 2 * LN;
    const u64 *masks = (h->sigma_on ? h->d_masks_sigma : h->d_masks) + (size_t)b0 * LN;
    struct MaskDiv {
        piehip_ctx *h;
        ~MaskDiv() { h->mask_div = h->key_group = 1; }
    } mask_div_scope{h};
    h->mask_div = nq;
    h->key_group = (nq > 1 && h->d_evkq && h->evkq_n == nq) ? nq : 1;  // per-query EvalMult keys: row r of a group is query r % nq
    // Operand X of the first ciphertext product (the accumulators of inner hash function 0) is needed twice: in COEFFICIENT
    // form by the base extension, and in EVALUATION form, lane-ordered, as the Q limbs of the QP operand (xq_reuse).  Until r04
    // the inverse transform wrote that second copy; now stage A writes X there in the first place and the transform reads it
    // from there (out of place, lane order in: its fast path) -- 44 MB less per step at the headline shape: the inverse launches
    // 161 -> 150 us per step of three queries, stage A + 2.5 (its X rows leave in 64-byte runs) and the base extension + 2.5.
    // Query batches only: one query's transform launches are single partial rounds that gain 1 us, and its stage A kernel, which
    // runs at the HBM rate, loses 2.5 (profiles/r04/stage_a_writes_x_lane_ordered.txt).
    const bool x_direct = K > 1 && nq > 1 && xq_reuse(h) && h->small_moduli && ntt16_applies(h->plan, h->fold_on);
    if (h->profiling) {  // an empty bracket: what the event pair itself costs on this stream (reported beside the kernels' times)
        ProfScope ps(h, PIEHIP_K_EVENT_PAIR, 0.0);
    }
    {   // stage A: all inner products of these bin layers in one launch (BatchedFHEHIPPIE.cpp:101-116)
        ProfScope ps(h, PIEHIP_K_STAGE_A, W * ((double)layers * K * E * L + nq * ((double)K * E * 2 * L + 2.0 * L + (double)layers * K * 2 * L)));
        StageAQueries qs = {};
        qs.idx[0] = h->d_idx, qs.minus[0] = h->d_minus;
        for (u32 q = 1; q < nq; q++) qs.idx[q] = h->bq_idx[q], qs.minus[q] = h->bq_minus[q];
        const u64 *db = h->d_db + (size_t)b0 * E * LN;
        StageAXOut xo;
        if (x_direct) xo.out = w.eqp, xo.M = M, xo.logns = h->hp.logN - (h->fold_on ? 1 : 0);
        if (nq > 1) {
            launch_stage_a_batch(h->d_dc, N, L, K, layers, E, qs, nq, db, acc, h->stream, h->small_moduli, b, 0, 0, x_direct ? &xo : nullptr);
        } else {
            launch_stage_a(h->d_dc, N, L, K, nb, E, h->d_idx, h->d_minus, db, acc, h->stream, h->small_moduli, b, 0, 0, 1, 0,
                           x_direct ? &xo : nullptr);
        }
    }
    if (K == 1) {
        // one inner hash function: multipliedResult is the inner product itself (BatchedFHEHIPPIE.cpp:117-120), so run() is
        // stage A and the mask multiply (:126) -- no ciphertext product, no transform, no key
        if (h->wait_before_results) (void)hipStreamWaitEvent(h->stream, h->wait_before_results, 0);
        ProfScope ps(h, PIEHIP_K_MASK, W * nb * 5.0 * L);
        launch_ct_mul_plain(h->d_dc, N, L, acc, h->d_masks + (size_t)b0 * LN, LN, out, nb, h->stream, nq);
        return;
    }
    // every accumulator enters a ct x ct product exactly once: switch them all to COEFFICIENT format
    const bool xq = xq_reuse(h);
    NttExtra ex;
    ex.copy_out = w.eqp;
    ex.copy_K = K;
    ex.copy_L = L;
    ex.copy_M = M;
    ex.x_lane_in = x_direct;
    ntt(h, acc, nb * K * 2 * L, 0, L, true, false, true, xq ? &ex : nullptr);
    ex.x_lane_in = false;
    // product chain over the inner hash functions (BatchedFHEHIPPIE.cpp:117-124); the mask multiply
    // (:126) is fused into the last key switch
    const u64 *x = acc;
    size_t sx = (size_t)K * 2 * LN;
    for (u32 hf = 1; hf < K; hf++) {
        const bool last = hf + 1 == K;
        u64 *dst = last ? out : prod;
        enqueue_mul(h, w, x, sx, acc + (size_t)hf * 2 * LN, (size_t)K * 2 * LN, nb, true, last ? masks : nullptr, dst, xq, last);
        if (!last) {
            ex.copy_K = 1;  // the product is the X operand of the next multiplication
            ntt(h, prod, nb * 2 * L, 0, L, true, false, true, xq ? &ex : nullptr);
            x = prod;
            sx = 2 * LN;
        }
    }
}

// the result ciphertexts of bin layers [b0, b0 + nb) (rows [bin layer][query]) to the caller's host array, on the current queue
static hipError_t download_rows(piehip_ctx *h, const u64 *d_results, u32 b0, u32 nb)
{
    const size_t row = (size_t)h->nq * 2 * h->LN();
    return hipMemcpyAsync(h->host_results + (size_t)b0 * row, d_results + (size_t)b0 * row, (size_t)nb * row * sizeof(u64),
                          hipMemcpyDeviceToHost, h->stream);
}

int piehip_run_into(piehip_handle h, void *d_results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!d_results) return fail(PIEHIP_EINVAL, "null result buffer");
    if (!h->K || !h->d_db) return fail(PIEHIP_ESTATE, "run: database not loaded");
    if (!h->d_acc || !h->ws.eqp) return fail(PIEHIP_ESTATE, "run: no workspace (an earlier allocation failed: piehip_set_query_batch / load)");
    if (!run_keys_loaded(h)) return fail(PIEHIP_ESTATE, "run: relinearisation key not loaded");
    if (!h->d_idx || !h->d_minus) return fail(PIEHIP_ESTATE, "run: setIndex / setMinusCompareElement not called");
    for (u32 q = 1; q < h->nq; q++)
        if (!h->bq_idx[q] || !h->bq_minus[q]) return fail(PIEHIP_ESTATE, "run: a query of the batch has no index matrix or minus element");
    HIPCHK(hipSetDevice(h->device));
    const u32 b = h->b;
    h->recs.clear();
    h->pool_used = 0;
    // Bin layers are independent: groups of them go to separate queues.  Most launches of one group leave part of the
    // chip idle (a transform launch is 0.4 .. 2 rounds of workgroups); another group's kernels fill it.
    // Ordering against the handle's stream:
    //   * inputs changed since the last run (setIndex, keys, database): the queues wait for the handle's stream first;
    //   * always: the kernel that writes the results waits for everything the caller queued on the handle's stream
    //     before this call (it may still be reading the result buffer of an earlier run);
    //   * the handle's stream joins the queues lazily, in the next entry point that is not a run (NEED / piehip_join),
    //     so back-to-back runs of one query batch keep every queue busy across run boundaries.
    const u32 ng = run_queue_count(h);
    if (ng > 1) {
        const int qrc = ensure_run_queues(h, ng);
        if (qrc) return qrc;
    }
    if (h->use_graph && !h->profiling && !h->host_results && h->nq == 1) {
        // One graph launch instead of ~13 kernel launches and 2 event operations per queue group: the same two chains, forked
        // from and joined back to the handle's stream inside the graph (so consecutive runs do not overlap each other, which
        // the eager path's lazy join allows).
        if (!h->gexec || h->g_idx != h->d_idx || h->g_minus != h->d_minus || h->g_res != d_results || h->g_ng != ng) {
            drop_graph(h);
            join_pending(h);
            struct Restore {
                piehip_ctx *h;
                hipStream_t s;
                ~Restore() { h->stream = s; }
            } restore{h, h->stream};
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(restore.s, hipStreamCaptureModeThreadLocal));
            hipError_t ce = hipSuccess;
            if (ng > 1) {
                ce = hipEventRecord(h->ev_fork, restore.s);
                u32 b0 = 0;
                for (u32 g = 0; g < ng && ce == hipSuccess; g++) {
                    const u32 nb = run_group_size(b, ng, g);
                    ce = hipStreamWaitEvent(h->side_streams[g], h->ev_fork, 0);
                    h->stream = h->side_streams[g];
                    enqueue_run_bins(h, b0, nb, (u64 *)d_results);
                    if (ce == hipSuccess) ce = hipEventRecord(h->ev_join[g], h->side_streams[g]);
                    if (ce == hipSuccess) ce = hipStreamWaitEvent(restore.s, h->ev_join[g], 0);
                    b0 += nb;
                }
                h->stream = restore.s;
            } else {
                enqueue_run_bins(h, 0, b, (u64 *)d_results);
            }
            const hipError_t ee = hipStreamEndCapture(restore.s, &graph);
            if (ce != hipSuccess || ee != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                return fail(PIEHIP_EHIP, std::string("run: graph capture failed: ") + hipGetErrorString(ce != hipSuccess ? ce : ee));
            }
            const hipError_t ie = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) {
                h->gexec = nullptr;
                return fail(PIEHIP_EHIP, std::string("run: hipGraphInstantiate: ") + hipGetErrorString(ie));
            }
            h->g_idx = h->d_idx, h->g_minus = h->d_minus, h->g_res = d_results, h->g_ng = ng;
        }
        join_pending(h);
        HIPCHK(hipGraphLaunch(h->gexec, h->stream));
        mark_dirty(h);  // the workspace is in use on the handle's stream
        return PIEHIP_OK;
    }
    if (ng > 1) {
        struct Restore {
            piehip_ctx *h;
            hipStream_t s;
            ~Restore()
            {
                h->stream = s;
                h->wait_before_results = nullptr;
            }
        } restore{h, h->stream};
        HIPCHK(hipEventRecord(h->ev_fork, restore.s));
        u32 b0 = 0;
        for (u32 g = 0; g < ng; g++) {
            const u32 nb = run_group_size(b, ng, g);
            if (h->inputs_dirty) HIPCHK(hipStreamWaitEvent(h->side_streams[g], h->ev_fork, 0));
            h->stream = h->side_streams[g];
            h->wait_before_results = h->inputs_dirty ? nullptr : h->ev_fork;
            // Results go down to host memory (piehip_run_staged / piehip_run_host*): the groups do not run side by side but one
            // behind the other -- group g + 1 starts when group g has only its result-writing kernel left, and the download of
            // group g (8 of 14 MiB at C3) travels under the evaluation of group g + 1.  Results in host memory after 0.53 ms
            // instead of 0.60 at C3, 1.30 instead of 1.50 for a batch of three; a stream of queries over several handles is
            // bound by the uploads either way (profiles/r04/online_phase_staggered_groups.txt).
            if (h->host_results && g > 0) HIPCHK(hipStreamWaitEvent(h->side_streams[g], h->ev_chain, 0));
            if (h->host_results && g + 1 < ng) {
                if (!h->ev_chain) HIPCHK(hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming));
                h->chain_armed = true;
            }
            enqueue_run_bins(h, b0, nb, (u64 *)d_results);
            if (h->chain_armed) {  // a chain without a key switch (K = 1): behind all of it
                h->chain_armed = false;
                HIPCHK(hipEventRecord(h->ev_chain, h->side_streams[g]));
            }
            if (h->host_results) HIPCHK(download_rows(h, (const u64 *)d_results, b0, nb));  // this group's slice, on this group's queue
            HIPCHK(hipEventRecord(h->ev_join[g], h->side_streams[g]));
            b0 += nb;
        }
        h->pending_join = true;
        h->inputs_dirty = false;
    } else {
        join_pending(h);
        enqueue_run_bins(h, 0, b, (u64 *)d_results);
        if (h->host_results) HIPCHK(download_rows(h, (const u64 *)d_results, 0, b));
        mark_dirty(h);  // the workspace is now in use on the handle's stream: the queues of a later multi-queue run wait for it
    }
    HIPCHK(hipGetLastError());
    return PIEHIP_OK;
}

int piehip_run(piehip_handle h)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->d_out) return fail(PIEHIP_ESTATE, "run: database not loaded");
    return piehip_run_into(h, h->d_out);
}

int piehip_join(piehip_handle h)
{
    NEED_RO(h);
    return PIEHIP_OK;
}

int piehip_set_graph(piehip_handle h, int on)
{
    NEED_RO(h);
    h->use_graph = on != 0;
    if (!on) drop_graph(h);
    return PIEHIP_OK;
}

int piehip_set_run_streams(piehip_handle h, uint32_t n)
{
    NEED_RO(h);
    h->run_streams = n;
    return PIEHIP_OK;
}

int piehip_set_transform_slots(piehip_handle h, uint32_t n)
{
    NEED_RO(h);
    h->plan.max_slots = n;
    drop_graph(h);   // a captured run() has the old grids baked in
    return PIEHIP_OK;
}

int piehip_get_transform_slots(piehip_handle h, uint32_t *n, uint32_t *device_slots)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (n) *n = h->plan.max_slots;
    if (device_slots) *device_slots = 2 * h->plan.num_cus;
    return PIEHIP_OK;
}

int piehip_sync(piehip_handle h)
{
    NEED_RO(h);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PIEHIP_OK;
}

int piehip_get_results(piehip_handle h, uint64_t *out)
{
    NEED(h);
    if (!out) return fail(PIEHIP_EINVAL, "null out");
    if (!h->d_out) return fail(PIEHIP_ESTATE, "no results");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out, h->d_out, sizeof(u64) * (size_t)h->b * h->nq * 2 * h->LN(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PIEHIP_OK;
}
int piehip_results_device(piehip_handle h, void **d_out)
{
    NEED_RO(h);
    if (!h->d_out) return fail(PIEHIP_ESTATE, "no results");
    *d_out = h->d_out;
    return PIEHIP_OK;
}

int piehip_copy_results_device(piehip_handle h, void *d_dst)
{
    NEED(h);
    if (!d_dst) return fail(PIEHIP_EINVAL, "null destination");
    if (!h->d_out) return fail(PIEHIP_ESTATE, "no results");
    HIPCHK(hipMemcpyAsync(d_dst, h->d_out, sizeof(u64) * (size_t)h->b * h->nq * 2 * h->LN(), hipMemcpyDeviceToDevice, h->stream));
    return PIEHIP_OK;
}

}  // extern "C"
