// piehip_fhepie.cpp -- FHEHIPPIE, the rotation-based sibling operator (reference
// src/Common/Crypto/PrivateIndexedEqualityCheck/FHEHIPPIE.{hpp,cpp}); off the timed path.
#include "piehip_ctx.hpp"

using namespace piehip;

extern "C" {

static int rot_map_device(piehip_ctx *h, int32_t index, u32 **out)
{
    auto it = h->rotmaps.find(index);
    if (it != h->rotmaps.end()) {
        *out = it->second;
        return PIEHIP_OK;
    }
    uint32_t g = 1;
    int rc = piehip_rotation_galois(h, index, &g);
    if (rc) return rc;
    std::vector<u32> map = h->hp.automorph_map(g);
    u32 *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(u32) * h->hp.N) != hipSuccess) return fail(PIEHIP_ENOMEM, "hipMalloc failed (rotation map)");
    if (hipMemcpy(d, map.data(), sizeof(u32) * h->hp.N, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(PIEHIP_EHIP, "rotation map upload failed");
    }
    h->rotmaps[index] = d;
    *out = d;
    return PIEHIP_OK;
}

int piehip_load_rotation_keys(piehip_handle h, uint32_t nkeys, const int32_t *indices, const uint64_t *keys)
{
    NEED(h);
    if (!nkeys || !indices || !keys) return fail(PIEHIP_EINVAL, "null operand");
    HIPCHK(hipSetDevice(h->device));
    const size_t words = (size_t)h->hp.L * 2 * h->LN();
    for (u32 i = 0; i < nkeys; i++) {
        uint32_t g = 1;
        int rc = piehip_rotation_galois(h, indices[i], &g);
        if (rc) return rc;
        if (g == 1) return fail(PIEHIP_EINVAL, "rotation index is a multiple of the row length");
        u64 *&d = h->rotkeys[indices[i]];
        if (!d && (rc = dev_alloc(&d, words))) {
            h->rotkeys.erase(indices[i]);
            return rc;
        }
        HIPCHK(hipMemcpy(d, keys + (size_t)i * words, words * sizeof(u64), hipMemcpyHostToDevice));
    }
    dev_free(&h->fp_negkeys);  // rebuilt from the new keys by the next run
    return PIEHIP_OK;
}

int piehip_fhepie_load_table(piehip_handle h, uint32_t npie, uint32_t K, uint32_t b, uint32_t E, const int64_t *slots,
                             const int64_t *masks)
{
    NEED(h);
    if (!slots || !masks) return fail(PIEHIP_EINVAL, "null operand");
    if (!npie || !K || !b || !E) return fail(PIEHIP_EINVAL, "empty table");
    // FHEHIPPIE.cpp:13-16: the bin size has to equal the number of bins per hash function
    if (b != E) return fail(PIEHIP_EINVAL, "for FHE PIE the size of a cuckoo bin has to be equal to the number of bins per hash function");
    if (E + 1 > h->hp.N / 2) return fail(PIEHIP_EINVAL, "E + 1 slots exceed one row of the packed encoding");
    HIPCHK(hipSetDevice(h->device));
    const u64 t = h->hp.t;
    const size_t LN = h->LN();
    const size_t npt = (size_t)npie * K * b, nmask = (size_t)npie * K;
    for (size_t i = 0; i < npt * (E + 1); i++)
        if ((u64)(slots[i] < 0 ? -slots[i] : slots[i]) >= t) return fail(PIEHIP_EINVAL, "slot value out of range for the plaintext modulus");
    for (size_t i = 0; i < nmask * b; i++)
        if (masks[i] <= 0 || (u64)masks[i] >= t) return fail(PIEHIP_EINVAL, "mask values must lie in [1, t-1]");
    dev_free(&h->fp_pt);
    dev_free(&h->fp_mask);
    dev_free(&h->fp_idx);
    dev_free(&h->fp_out);
    h->fp_npie = 0;
    int rc;
    if ((rc = dev_alloc(&h->fp_pt, npt * LN))) return rc;
    if ((rc = dev_alloc(&h->fp_mask, nmask * LN))) return rc;
    if (!h->fp_e0 && (rc = dev_alloc(&h->fp_e0, LN))) return rc;
    if ((rc = dev_alloc(&h->fp_idx, nmask * 2 * LN))) return rc;
    if ((rc = dev_alloc(&h->fp_out, nmask * 2 * LN))) return rc;
    Tmp tmp;
    TMPGET(d_slots, npt * (E + 1));
    TMPGET(d_masks, nmask * b);
    TMPGET(d_one, 1);
    const int64_t one = 1;
    HIPCHK(hipMemcpy(d_slots, slots, sizeof(int64_t) * npt * (E + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_masks, masks, sizeof(int64_t) * nmask * b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_one, &one, sizeof(int64_t), hipMemcpyHostToDevice));
    if ((rc = encode_on_device(h, (const int64_t *)d_slots, (u32)npt, E + 1, h->fp_pt))) return rc;
    if ((rc = encode_on_device(h, (const int64_t *)d_masks, (u32)nmask, b, h->fp_mask))) return rc;
    if ((rc = encode_on_device(h, (const int64_t *)d_one, 1, 1, h->fp_e0))) return rc;
    h->fp_npie = npie;
    h->fp_K = K;
    h->fp_b = b;
    h->fp_E = E;
    dev_free(&h->fp_negkeys);
    return PIEHIP_OK;
}

int piehip_fhepie_set_index(piehip_handle h, const uint64_t *idx)
{
    NEED(h);
    if (!idx) return fail(PIEHIP_EINVAL, "null index");
    if (!h->fp_npie) return fail(PIEHIP_ESTATE, "no table loaded");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->fp_idx, idx, sizeof(u64) * (size_t)h->fp_npie * h->fp_K * 2 * h->LN(), hipMemcpyHostToDevice));
    return PIEHIP_OK;
}

// FHEHIPPIE::run (FHEHIPPIE.cpp:61-77) for npie operators at once, hash function by hash function:
//   prod[pie][bin] = idx[pie][hf] (.) pt[pie][hf][bin]                       EvalInnerProduct: EvalMult ...
//   R = ceil(log2(b)) times: prod += rotate(prod, 2^r)                       ... then EvalSum over batchSize = #bins
//   merged[pie] = sum_bin rotate(prod[pie][bin] (.) e0, -bin)                EvalMerge
//   out[pie][hf] = merged (.) mask[pie][hf]                                  EvalMult(.., preCalcRandomMask)
int piehip_fhepie_run(piehip_handle h)
{
    NEED(h);
    if (!h->fp_npie) return fail(PIEHIP_ESTATE, "no table loaded");
    HIPCHK(hipSetDevice(h->device));
    const u32 N = h->hp.N, L = h->hp.L, npie = h->fp_npie, K = h->fp_K, b = h->fp_b;
    const size_t LN = h->LN(), keyw = (size_t)L * 2 * LN;
    const u32 nb = npie * b;
    u32 R = 0;
    while ((1u << R) < b) R++;  // EvalSum(ct, batchSize = vectorizedCT[hf].size()): ceil(log2) rotate-and-add steps
    std::vector<const u64 *> sumkeys(R);
    std::vector<u32 *> summaps(R);
    int rc;
    for (u32 r = 0; r < R; r++) {
        auto it = h->rotkeys.find((int32_t)(1u << r));
        if (it == h->rotkeys.end()) return fail(PIEHIP_ESTATE, "EvalSum key for rotation " + std::to_string(1u << r) + " not loaded");
        sumkeys[r] = it->second;
        if ((rc = rot_map_device(h, (int32_t)(1u << r), &summaps[r]))) return rc;
    }
    if (!h->fp_negkeys && b > 1) {  // keys / maps of rotations -1 .. -(b-1), position r holds rotation -r
        if ((rc = dev_alloc(&h->fp_negkeys, (size_t)b * keyw))) return rc;
        if (h->fp_negmaps) (void)hipFree(h->fp_negmaps);
        h->fp_negmaps = nullptr;
        if (hipMalloc((void **)&h->fp_negmaps, sizeof(u32) * (size_t)b * N) != hipSuccess) return fail(PIEHIP_ENOMEM, "hipMalloc failed (merge maps)");
        for (u32 r = 1; r < b; r++) {
            auto it = h->rotkeys.find(-(int32_t)r);
            if (it == h->rotkeys.end()) {
                dev_free(&h->fp_negkeys);
                return fail(PIEHIP_ESTATE, "EvalAtIndex key for rotation -" + std::to_string(r) + " not loaded");
            }
            u32 *dm = nullptr;
            if ((rc = rot_map_device(h, -(int32_t)r, &dm))) return rc;
            HIPCHK(hipMemcpyAsync(h->fp_negkeys + (size_t)r * keyw, it->second, keyw * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->fp_negmaps + (size_t)r * N, dm, sizeof(u32) * N, hipMemcpyDeviceToDevice, h->stream));
        }
        // position 0 is never rotated: its digits are zero, any valid key / map will do
        HIPCHK(hipMemcpyAsync(h->fp_negkeys, h->fp_negkeys + keyw, keyw * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->fp_negmaps, h->fp_negmaps + N, sizeof(u32) * N, hipMemcpyDeviceToDevice, h->stream));
    }
    MulWs w;
    u64 *prod = nullptr;
    auto cleanup = [&]() {
        ws_free(w);
        dev_free(&prod);
    };
    // only d01 / d2c / dig of the workspace are used here
    w.nb = nb;
    if ((rc = dev_alloc(&w.d01, (size_t)nb * 2 * LN)) || (rc = dev_alloc(&w.d2c, (size_t)nb * LN)) ||
        (rc = dev_alloc(&w.dig, (size_t)nb * L * LN)) || (rc = dev_alloc(&prod, (size_t)nb * 2 * LN))) {
        cleanup();
        return rc;
    }
    h->pool_used = 0;
    h->recs.clear();
    const double W = 8.0 * N;
    for (u32 hf = 0; hf < K; hf++) {
        {
            ProfScope ps(h, PIEHIP_K_MASK, W * nb * 5.0 * L);
            launch_bcast_mul_plain(h->d_dc, N, L, h->fp_idx + (size_t)hf * 2 * LN, (size_t)K * 2 * LN, b,
                                   h->fp_pt + (size_t)hf * b * LN, (size_t)K * b * LN, LN, prod, nb, h->stream);
        }
        for (u32 r = 0; r < R; r++) {
            {
                ProfScope ps(h, PIEHIP_K_AUTOMORPH, W * nb * 7.0 * L);
                launch_rot_prepare(h->d_dc, N, L, prod, summaps[r], 1, true, false, w.d01, w.d2c, nb, h->stream);
            }
            ntt(h, w.d2c, nb * L, 0, L, true);
            enqueue_keyswitch(h, w, nb, sumkeys[r], nullptr, prod);
        }
        if (b > 1) {
            {
                ProfScope ps(h, PIEHIP_K_MASK, W * nb * 5.0 * L);
                launch_ct_mul_plain(h->d_dc, N, L, prod, h->fp_e0, 0, prod, nb, h->stream);
            }
            {
                ProfScope ps(h, PIEHIP_K_AUTOMORPH, W * nb * 5.0 * L);
                launch_rot_prepare(h->d_dc, N, L, prod, h->fp_negmaps, b, false, true, w.d01, w.d2c, nb, h->stream);
            }
            ntt(h, w.d2c, nb * L, 0, L, true);
            enqueue_keyswitch(h, w, nb, h->fp_negkeys, nullptr, prod, false, false, keyw, b);
        } else {
            ProfScope ps(h, PIEHIP_K_MASK, W * nb * 5.0 * L);
            launch_ct_mul_plain(h->d_dc, N, L, prod, h->fp_e0, 0, prod, nb, h->stream);
        }
        {
            ProfScope ps(h, PIEHIP_K_MASK, W * (nb * 2.0 * L + npie * 3.0 * L));
            launch_sum_mul_plain(h->d_dc, N, L, prod, b, h->fp_mask + (size_t)hf * LN, (size_t)K * LN, h->fp_out + (size_t)hf * 2 * LN,
                                 (size_t)K * 2 * LN, npie, h->stream);
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(PIEHIP_EHIP, std::string("fhepie_run: ") + hipGetErrorString(e));
    return PIEHIP_OK;
}

int piehip_fhepie_get_results(piehip_handle h, uint64_t *out)
{
    NEED(h);
    if (!out) return fail(PIEHIP_EINVAL, "null out");
    if (!h->fp_npie) return fail(PIEHIP_ESTATE, "no table loaded");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(out, h->fp_out, sizeof(u64) * (size_t)h->fp_npie * h->fp_K * 2 * h->LN(), hipMemcpyDeviceToHost));
    return PIEHIP_OK;
}

}  // extern "C"
