// piehip_ctx.hpp -- what the translation units behind include/piehip.h share: the context behind a piehip_handle, error /
// ordering macros, device scratch, event-bracketed launches, and the schedule pieces of a ciphertext multiplication.
//   piehip.cpp         context, keys, database (offline phase), query inputs, run() and its queues
//   piehip_host.cpp    the host-memory path of a query: page-locked staging, piecewise uploads, run_staged / run_host
//   piehip_ops.cpp     the OpenFHE primitives one by one (parity tests), NTT timing, per-kernel profiling
//   piehip_fhepie.cpp  the rotation-based sibling operator (FHEHIPPIE)
//   piehip_client.cpp  client-side harness (key generation, encryption, decryption)
//   piehip_rccl.cpp    the final gather of a sharded server over RCCL
#pragma once
#include "../../include/piehip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "params.hpp"

namespace piehip {
int fail(int code, const std::string &msg);  // sets piehip_last_error() of this thread, returns code
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return piehip::fail(PIEHIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
struct piehip_ctx;
namespace piehip {
void join_pending(piehip_ctx *h);
void mark_dirty(piehip_ctx *h);
}
// every entry point except piehip_run first orders the handle's stream behind the bin-layer queues of earlier runs
#define NEED_RO(h)                                                       \
    do {                                                                 \
        if (!(h)) return piehip::fail(PIEHIP_EINVAL, "null handle");     \
        piehip::join_pending(h);                                         \
    } while (0)
// ... and, unless it only reads (NEED_RO), may queue work on the handle's stream that the next run has to wait for
#define NEED(h)                                                 \
    do {                                                        \
        NEED_RO(h);                                             \
        piehip::mark_dirty(h);                                  \
    } while (0)

namespace piehip {

struct ProfRec {
    hipEvent_t a, b;
    int k;
    double bytes;
};

// scratch of one batched EvalMult(ct,ct) over nb ciphertext pairs
struct MulWs {
    u32 nb = 0;
    u64 *eqp = nullptr;  // [nb][4][M][N]
    u64 *dqp = nullptr;  // [nb][3][M][N]
    u64 *d01 = nullptr;  // [nb][2][L][N]
    u64 *d2c = nullptr;  // [nb][L][N]
    u64 *dig = nullptr;  // [nb][L][L][N]
};

// one query's way in from host memory (piehip_host_buffers_q, piehip_stage_*_q)
struct QueryStage {
    u64 *pin_idx = nullptr, *pin_minus = nullptr;  // page-locked staging [K][E][2][L][N], [2][L][N]
    std::vector<bool> rows;                        // pieces on their way since the staging sequence began
    std::vector<bool> cts;                         // ... ciphertext by ciphertext (piehip_stage_index_ct_q), [K][E]
    bool minus = false;
};

}  // namespace piehip
using piehip::u32;
using piehip::u64;

struct piehip_ctx {
    piehip::HostParams hp;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<hipStream_t> side_streams;        // extra queues of run(): one group of bin layers each (see piehip_run)
    std::vector<hipEvent_t> ev_join;
    hipEvent_t ev_fork = nullptr;
    // runs that bring their results down to host memory: queue group g + 1 starts when group g has reached the kernel that writes
    // its results, so that the download of group g runs under the evaluation of group g + 1 (piehip_run_into)
    hipEvent_t ev_chain = nullptr;
    bool chain_armed = false;                     // set while such a group is enqueued: recorded in front of its result-writing kernel
    u32 run_streams = 0;                          // piehip_set_run_streams: 0 = all queues
    bool inputs_dirty = true;                     // inputs / keys / database changed on the handle's stream since the last run()
    hipEvent_t wait_before_results = nullptr;     // set while run() enqueues a group: its result-writing kernel waits for this
    bool pending_join = false;                    // run() left work on the bin-layer queues that the handle's stream has not waited for
    // piehip_set_graph: run() as one captured hipGraph per (inputs, result buffer, queue count), replayed on the handle's stream
    bool use_graph = false;
    hipGraphExec_t gexec = nullptr;
    const void *g_idx = nullptr, *g_minus = nullptr, *g_res = nullptr;
    u32 g_ng = 0;
    // the host-memory path (piehip_host.cpp): per query of the batch its page-locked staging and which pieces have been staged;
    // all transfers travel on the handle's own queues (no copy stream: see piehip_host.cpp)
    piehip::QueryStage qstage[piehip::STAGE_A_MAX_QUERIES];
    bool stage_open = false;                      // piehip_stage_*: the uploads of the next run()'s queries have begun
    u64 *pin_up_flag = nullptr;                   // page-locked word: sequence number of the last query of this handle whose uploads have
    u64 up_seq = 0;                               // left host memory (written by a one-thread kernel behind them); the next number
    u64 up_turn_wait_ns = 0;                      // how long the last staging sequence waited for its turn on the device's link
    u64 up_turn_wait_total_ns = 0, up_turn_waits = 0;   // ... and all of them so far (piehip_upload_turn_wait)
    // piehip_set_host_path_timing: three events per query on the handle's stream -- first staged piece, uploads handed over, results down
    bool hp_timing = false;
    hipEvent_t hp_ev[3] = {nullptr, nullptr, nullptr};
    int hp_ev_state = 0;                          // 0 nothing recorded, 1 first piece, 2 handed over, 3 complete sequence recorded
    u64 *host_results = nullptr;                  // set while piehip_run_staged enqueues: every queue group downloads its slice there
    u64 *pin_res = nullptr;                       // [b][nq][2][L][N]
    size_t pin_idx_words = 0, pin_res_words = 0;
    piehip::DevConsts *d_dc = nullptr;
    u64 *d_tables = nullptr;  // [(M+1)][4][N]
    u64 *d_twp = nullptr;     // [(M+1)][2][N][2] interleaved {w, w_shoup}
    u64 *d_twc = nullptr;     // pass-C kernel-order copy of the same pairs
    u64 *d_twc_fold = nullptr;  // ... for the folded configuration (two half-size slices per limb)
    bool fold_on = false;     // outermost NTT stage folded into the coefficient-wise kernels (N >= 2^14)
    u32 *d_inv_pos = nullptr; // EVALUATION position -> slot
    u32 *d_sigma_inv = nullptr;  // lane-order position -> standard position (identity for small rings)
    u32 sigma_T = 0;             // threads per slice of the transform that defines the lane order
    u32 sigma_kp = 16;           // ... and coefficient pairs per thread (16: kernels_ntt_fast.hip, 8: ntt16_kernel.h)
    u64 *d_twk16 = nullptr;      // ntt16_kernel.h tables (ring 2^13 as one slice per limb; rings 2^14, 2^15 as two folded slices)
    bool small_moduli = false;   // all Q and P moduli in (2^59, 2^60): v_mad_u64_u32 column accumulators, one-word Barrett
    bool sigma_on = false;       // the register-blocked NTT (and hence the lane order) applies to this context
    u64 *d_evk_sigma = nullptr, *d_masks_sigma = nullptr;  // lane-ordered copies of key and masks
    u64 *d_hash_tbl = nullptr;   // [k][e][K][b][E] of the last piehip_build_db
    size_t hash_tbl_words = 0;
    // scratch of the offline phase (hashing, packing, encoding), kept between calls: piehip_reserve sizes it up front so that
    // the timed offline phase allocates nothing (hipMalloc / hipFree cost milliseconds and synchronise the device)
    u64 *arena = nullptr;
    size_t arena_words = 0, arena_used = 0;
    u32 hk = 0, he = 0, hb = 0;  // table dimensions ([k][e][K][hb][E]; hb = all bin layers, of which this handle keeps b)
    piehip::NttPlan plan;
    // keys / database / inputs
    bool db_borrowed = false;  // piehip_attach_database: d_evk*, d_db, d_masks* belong to another handle (never freed or written here)
    piehip_ctx *db_owner = nullptr;  // ... that handle
    u32 db_borrowers = 0;            // handles that borrow from this one: its buffers may not move while > 0
    u64 *d_evk = nullptr;
    u32 K = 0, b = 0, E = 0;
    u64 *d_db = nullptr, *d_masks = nullptr;
    u64 *d_idx_own = nullptr, *d_minus_own = nullptr;
    const u64 *d_idx = nullptr, *d_minus = nullptr;
    // query batch (piehip_set_query_batch): run() evaluates nq queries against the database at once; query 0 is d_idx / d_minus
    // above, queries 1 .. nq - 1 are bq_*[q].  Workspace and results hold nq rows per bin layer: [b][nq][..].
    u32 nq = 1;
    u32 mask_div = 1;  // set while run() enqueues a batch: ciphertext row r of the product chain takes mask r / mask_div
    u32 key_group = 1; // ... and, with per-query EvalMult keys, key r % key_group of d_evkq
    const u64 *bq_idx[piehip::STAGE_A_MAX_QUERIES] = {}, *bq_minus[piehip::STAGE_A_MAX_QUERIES] = {};
    u64 *bq_idx_own[piehip::STAGE_A_MAX_QUERIES] = {}, *bq_minus_own[piehip::STAGE_A_MAX_QUERIES] = {};
    // piehip_load_relin_key_q: the queries of a batch come from different clients, each with its own EvalMult key.  [evkq_n] keys
    // [L][2][L][N] one after the other (+ the lane-ordered copy); entries nobody loaded hold the handle's key
    u64 *d_evkq = nullptr, *d_evkq_sigma = nullptr;
    u32 evkq_n = 0, evkq_loaded = 0;    // keys the array holds; bit q: query q loaded its own
    // run() workspace
    u64 *d_acc = nullptr;   // [b][K][2][L][N]
    u64 *d_prod = nullptr;  // [b][2][L][N]  (K > 2 only)
    u64 *d_out = nullptr;   // [b][2][L][N]
    piehip::MulWs ws;
    size_t ws_cap_rows = 0;   // rows (bin layer x query) the workspace arrays were allocated for; ws.nb = rows in use
    u32 ws_cap_K = 0;
    // sharded server (piehip_rccl.cpp): this handle's rank in an RCCL communicator (ncclComm_t; owned if piehip_rccl_init made it)
    void *comm = nullptr;
    bool comm_owned = false;
    int comm_ranks = 0, comm_rank = 0;
    u64 *d_gather = nullptr, *pin_gather = nullptr;   // the root's gathered result list [b_total][nq][2][L][N]: HBM, page-locked host
    size_t gather_words = 0;
    // rotation-based PIE (FHEHIPPIE): rotation keys by index, EVALUATION index maps, packed sub-tables
    std::map<int32_t, u64 *> rotkeys;   // [L][2][L][N] each
    std::map<int32_t, u32 *> rotmaps;   // [N] each
    u32 fp_npie = 0, fp_K = 0, fp_b = 0, fp_E = 0;
    u64 *fp_pt = nullptr;     // [npie][K][b][L][N]
    u64 *fp_mask = nullptr;   // [npie][K][L][N]
    u64 *fp_e0 = nullptr;     // [L][N]: plaintext with slot 0 = 1 (EvalMerge's mask)
    u64 *fp_idx = nullptr;    // [npie][K][2][L][N]
    u64 *fp_out = nullptr;    // [npie][K][2][L][N]
    u64 *fp_negkeys = nullptr;  // [b][L][2][L][N]: key of rotation -r at position r (position 0 unused)
    u32 *fp_negmaps = nullptr;  // [b][N]
    // profiling
    bool profiling = false;
    std::vector<piehip::ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;

    size_t LN() const { return (size_t)hp.L * hp.N; }
};

namespace piehip {

// the key switch of run() has a key for every query: the handle's, or one per query of the batch (piehip_load_relin_key_q)
inline bool run_keys_loaded(const piehip_ctx *h)
{
    return h->K <= 1 || h->d_evk || (h->d_evkq && h->evkq_n == h->nq && h->evkq_loaded == (1u << h->nq) - 1);
}

hipEvent_t prof_event(piehip_ctx *h);
// brackets the launches queued in its scope with HIP events on the handle's current stream when profiling is on
struct ProfScope {
    piehip_ctx *h;
    ProfRec r;
    bool on;
    ProfScope(piehip_ctx *h_, int k, double bytes) : h(h_), on(h_->profiling)
    {
        if (!on) return;
        r.k = k;
        r.bytes = bytes;
        r.a = prof_event(h);
        r.b = prof_event(h);
        if (!r.a || !r.b) {
            on = false;
            return;
        }
        (void)hipEventRecord(r.a, h->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.b, h->stream);
        h->recs.push_back(r);
    }
};

void drop_graph(piehip_ctx *h);
int dev_alloc(u64 **p, size_t words);
void dev_free(u64 **p);

struct Tmp {  // RAII device scratch: carved from the handle's arena while it has room, hipMalloc otherwise
    piehip_ctx *h = nullptr;
    size_t mark = 0;
    std::vector<u64 *> ptrs;
    Tmp() {}
    explicit Tmp(piehip_ctx *h_) : h(h_), mark(h_->arena_used) {}
    ~Tmp()
    {
        for (u64 *p : ptrs) (void)hipFree(p);
        if (h) h->arena_used = mark;
    }
    u64 *get(size_t words)
    {
        if (!words) words = 1;
        const size_t w32 = (words + 31) & ~(size_t)31;  // 256-byte granules
        if (h && h->arena && h->arena_used + w32 <= h->arena_words) {
            u64 *p = h->arena + h->arena_used;
            h->arena_used += w32;
            return p;
        }
        u64 *p = nullptr;
        if (hipMalloc((void **)&p, words * sizeof(u64)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};
#define TMPGET(var, words)                                          \
    piehip::u64 *var = tmp.get(words);                              \
    if (!var) return piehip::fail(PIEHIP_ENOMEM, "hipMalloc failed (scratch)")

int ws_alloc(piehip_ctx *h, MulWs &w, u32 nb);
void ws_free(MulWs &w);

// ---- schedule pieces (piehip.cpp) ------------------------------------------------------------------
// sigma: lane order on the EVALUATION side; fold: outer stage applied by the neighbouring kernels (both only
// take effect when the context supports them; callers pass the same flags to those neighbours)
void ntt(piehip_ctx *h, u64 *data, u32 nlimbs, u32 mod_base, u32 mod_count, bool inv, bool sigma = false, bool fold = false,
         const NttExtra *ex = nullptr);
bool xq_reuse(const piehip_ctx *h);
void enqueue_keyswitch(piehip_ctx *h, MulWs &w, u32 nb, const u64 *key, const u64 *mask, u64 *out, bool sigma = false,
                       bool fold = false, size_t key_stride = 0, u32 key_group = 1, bool out_is_result = false,
                       bool digits_ready = false);
void enqueue_mul(piehip_ctx *h, MulWs &w, const u64 *x, size_t sx, const u64 *y, size_t sy, u32 nb, bool relin,
                 const u64 *mask, u64 *out, bool xq_ready = false, bool out_is_result = false);
int encode_on_device(piehip_ctx *h, const int64_t *d_slots, u32 npt, u32 B, u64 *d_out);
// device input buffers of query q of the batch (owned copies: the host setters and the staged uploads write them)
int query_input_buffers(piehip_ctx *h, u32 q, u64 **d_idx, u64 **d_minus);
void free_host_path(piehip_ctx *h);   // piehip_host.cpp: page-locked staging
// queues of a run() and the bin layers each takes
u32 run_queue_count(const piehip_ctx *h);
int ensure_run_queues(piehip_ctx *h, u32 ng);
u32 run_group_size(u32 b, u32 ng, u32 g);

}  // namespace piehip
