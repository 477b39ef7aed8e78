// piehip_rccl.cpp -- the collectives of a sharded server (one process per GPU) behind the C ABI, over RCCL / xGMI:
//   the final gather of the result ciphertexts to the rank that answers the client (the path's only exchange: bin layers are
//   independent, reference BatchedFHEHIPPIE.cpp:91; sendResult at BatchedFHEPSIServer.cpp:143-152), and the per-query
//   distribution of the inputs from the rank that holds the client's socket (.cpp:94-95,114-141).
// A C++ server (host/ShardedBatchedFHEPSIServer.hpp) needs nothing but this library and RCCL; torch.distributed is only the
// Python harness's way to the same collectives (shard.py).
//
// RCCL is bound at run time (dlopen), not at link time: a one-GPU deployment never loads the 0.5 GB library, and a process that
// already holds a copy (PyTorch bundles its own librccl.so) gets THAT copy, so a communicator made elsewhere in the process can
// be attached (piehip_rccl_attach).
#include "piehip_ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>
#include <thread>

using namespace piehip;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (a library without them still gathers; piehip_rccl_wait then only has its time-out, piehip_rccl_agree is refused)
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    std::string error;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// the process's RCCL: the copy that is already mapped if there is one, else the ROCm installation's
const Rccl *rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    Rccl &r = g_rccl;
    if (r.lib || !r.error.empty()) return r.lib ? &r : nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    for (size_t i = 0; !r.lib && i < sizeof(names) / sizeof(names[0]); i++) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
        r.error = std::string("RCCL is not available: ") + dlerror();
        return nullptr;
    }
#define BIND(field, sym)                                                      \
    do {                                                                      \
        *(void **)(&r.field) = dlsym(r.lib, sym);                             \
        if (!r.field) {                                                       \
            r.error = std::string("RCCL lacks ") + sym;                       \
            r.lib = nullptr;                                                  \
            return nullptr;                                                   \
        }                                                                     \
    } while (0)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
    BIND(Broadcast, "ncclBroadcast");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    *(void **)(&r.CommAbort) = dlsym(r.lib, "ncclCommAbort");
    *(void **)(&r.CommGetAsyncError) = dlsym(r.lib, "ncclCommGetAsyncError");
    *(void **)(&r.AllReduce) = dlsym(r.lib, "ncclAllReduce");
    return &r;
}

int no_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    return fail(PIEHIP_EHIP, g_rccl.error.empty() ? "RCCL is not available" : g_rccl.error);
}

}  // namespace

#define NCCLCHK(R, expr)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(PIEHIP_EHIP, std::string(#expr) + ": " + (R)->GetErrorString(r_));     \
    } while (0)

extern "C" {

int piehip_rccl_unique_id(void *id)
{
    if (!id) return fail(PIEHIP_EINVAL, "null id");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    ncclUniqueId u;
    NCCLCHK(R, R->GetUniqueId(&u));
    static_assert(sizeof(u) == PIEHIP_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof(u));
    return PIEHIP_OK;
}

int piehip_rccl_init(piehip_handle h, const void *id, int nranks, int rank)
{
    NEED_RO(h);
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(PIEHIP_EINVAL, "rccl_init: bad rank or id");
    if (h->comm) return fail(PIEHIP_ESTATE, "rccl_init: the handle already has a communicator");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    NCCLCHK(R, R->CommInitRank(&c, nranks, u, rank));
    h->comm = c;
    h->comm_owned = true;
    h->comm_ranks = nranks;
    h->comm_rank = rank;
    return PIEHIP_OK;
}

int piehip_rccl_attach(piehip_handle h, void *comm, int nranks, int rank)
{
    NEED_RO(h);
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(PIEHIP_EINVAL, "rccl_attach: bad communicator or rank");
    if (h->comm) return fail(PIEHIP_ESTATE, "rccl_attach: the handle already has a communicator");
    if (!rccl()) return no_rccl();
    h->comm = comm;
    h->comm_owned = false;
    h->comm_ranks = nranks;
    h->comm_rank = rank;
    return PIEHIP_OK;
}

int piehip_rccl_destroy(piehip_handle h)
{
    NEED_RO(h);
    if (!h->comm) return PIEHIP_OK;
    const Rccl *R = rccl();
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));  // collectives queued on the handle's stream have drained
    if (h->comm_owned && R) (void)R->CommDestroy((ncclComm_t)h->comm);
    dev_free(&h->d_gather);
    if (h->pin_gather) (void)hipHostFree(h->pin_gather);
    h->pin_gather = nullptr;
    h->gather_words = 0;
    h->comm = nullptr;
    h->comm_owned = false;
    h->comm_ranks = h->comm_rank = 0;
    return PIEHIP_OK;
}

// Giving up.  A collective is queued in the handle's stream and completes when every rank has queued its side; a rank that
// never does (it died, it failed a precondition and returned before queueing anything, it is a programming error away from the
// others) leaves its peers' streams blocked for ever -- the reference aborts the whole process on any error
// (BatchedFHEHIPPIE.cpp:15,20), a hung group of server processes is worse.  So nobody waits without a bound:
//   piehip_rccl_wait    piehip_sync with a time-out: polls the handle's stream and the communicator's asynchronous error state;
//                       when the time is up (or RCCL reports a failed peer) it ABORTS the communicator -- which releases the
//                       blocked stream -- and returns PIEHIP_EHIP.  The communicator is gone afterwards.
//   piehip_rccl_abort   the same on purpose: a rank that cannot take part any more (an exception on its way out) tears its
//                       side down so that its peers' waits end at once instead of at their time-out.
//   piehip_rccl_agree   one word from every rank, everybody learns whether ALL said yes (an all-reduce + wait): the ranks of a
//                       server call it at the end of a phase (database built? key loaded?) so that a failure on one of them
//                       ends the session everywhere before anybody enters a collective the failed rank will not join.
static int abort_comm(piehip_ctx *h, const Rccl *R)
{
    if (!h->comm) return PIEHIP_OK;
    if (R && R->CommAbort && h->comm_owned) (void)R->CommAbort((ncclComm_t)h->comm);
    h->comm = nullptr;
    h->comm_owned = false;
    h->comm_ranks = h->comm_rank = 0;
    return PIEHIP_OK;
}

int piehip_rccl_abort(piehip_handle h)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->comm) return PIEHIP_OK;
    HIPCHK(hipSetDevice(h->device));
    return abort_comm(h, rccl());
}

int piehip_rccl_wait(piehip_handle h, uint32_t timeout_ms)
{
    NEED_RO(h);
    HIPCHK(hipSetDevice(h->device));
    if (!h->comm) {   // nothing of RCCL's can be in the stream: a plain wait
        HIPCHK(hipStreamSynchronize(h->stream));
        return PIEHIP_OK;
    }
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    const auto t0 = std::chrono::steady_clock::now();
    std::string why;
    // The common case is a query's worth of work (well under a millisecond at the headline shape): the stream is polled back to back
    // for the first milliseconds -- this wait sits inside the server's online timer -- and the communicator's error state and the
    // clock are looked at every 64th poll; after 5 ms the loop backs off to one poll per 50 us.
    for (u32 polls = 0;; polls++) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(PIEHIP_EHIP, std::string("rccl_wait: ") + hipGetErrorString(q));
        if ((polls & 63) != 63) {
            std::this_thread::yield();
            continue;
        }
        ncclResult_t ae = ncclSuccess;
        if (R->CommGetAsyncError && R->CommGetAsyncError((ncclComm_t)h->comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
            why = std::string("rccl_wait: the communicator reports ") + R->GetErrorString(ae);
            break;
        }
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(timeout_ms)) {
            why = "rccl_wait: timed out after " + std::to_string(timeout_ms) + " ms -- a rank of the server group did not join the collective";
            break;
        }
        if (waited > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50) * 64);
    }
    if (why.empty()) {
        // (the stream has drained; a transfer that FAILED also drains it: ask once more)
        ncclResult_t ae = ncclSuccess;
        if (R->CommGetAsyncError && R->CommGetAsyncError((ncclComm_t)h->comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress)
            why = std::string("rccl_wait: the communicator reports ") + R->GetErrorString(ae);
        else
            return PIEHIP_OK;
    }
    const bool owned = h->comm_owned;
    abort_comm(h, R);
    if (!owned)   // piehip_rccl_attach: the communicator is the caller's to abort -- until then the stream may still be blocked
        return fail(PIEHIP_EHIP, why + " (attached communicator dropped: the caller must ncclCommAbort it to release the stream)");
    (void)hipStreamSynchronize(h->stream);   // the abort released whatever of the collective sat in the stream
    return fail(PIEHIP_EHIP, why + " (communicator aborted)");
}

int piehip_rccl_agree(piehip_handle h, int ok, int *all_ok, uint32_t timeout_ms)
{
    NEED(h);
    if (!all_ok) return fail(PIEHIP_EINVAL, "null result");
    if (!h->comm) return fail(PIEHIP_ESTATE, "rccl_agree: no communicator");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    if (!R->AllReduce) return fail(PIEHIP_EHIP, "rccl_agree: this RCCL has no ncclAllReduce");
    HIPCHK(hipSetDevice(h->device));
    Tmp tmp(h);
    TMPGET(d_word, 1);
    int32_t *pin = nullptr;
    HIPCHK(hipHostMalloc((void **)&pin, 64, hipHostMallocPortable));
    pin[0] = ok ? 1 : 0;
    int rc = PIEHIP_OK;
    do {
        if (hipMemcpyAsync(d_word, pin, 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(PIEHIP_EHIP, "rccl_agree: upload"); break; }
        const ncclResult_t r = R->AllReduce(d_word, d_word, 1, ncclInt32, ncclMin, (ncclComm_t)h->comm, h->stream);
        if (r != ncclSuccess) { rc = fail(PIEHIP_EHIP, std::string("rccl_agree: ") + R->GetErrorString(r)); break; }
        if (hipMemcpyAsync(pin, d_word, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc = fail(PIEHIP_EHIP, "rccl_agree: download"); break; }
        rc = piehip_rccl_wait(h, timeout_ms);
    } while (0);
    if (rc == PIEHIP_OK) *all_ok = pin[0] != 0;
    (void)hipHostFree(pin);
    return rc;
}

// [lo, hi): the bin layers of rank r of G (contiguous, sizes differ by at most one: shard.bin_slice, ShardedBatchedFHEHIPPIE)
static void rank_slice(u32 b, int r, int G, u32 *lo, u32 *hi)
{
    *lo = (u32)((u64)b * (u64)r / (u64)G);
    *hi = (u32)((u64)b * (u64)(r + 1) / (u64)G);
}

int piehip_rccl_bin_slice(uint32_t b_total, int nranks, int rank, uint32_t *bin_lo, uint32_t *bin_hi)
{
    if (!bin_lo || !bin_hi || nranks < 1 || rank < 0 || rank >= nranks) return fail(PIEHIP_EINVAL, "bad rank");
    rank_slice(b_total, rank, nranks, bin_lo, bin_hi);
    return PIEHIP_OK;
}

int piehip_gather_results(piehip_handle h, uint32_t b_total, int root, void *d_out)
{
    NEED(h);   // the handle's stream is behind the run whose results travel
    if (!h->comm) return fail(PIEHIP_ESTATE, "gather_results: no communicator (piehip_rccl_init / piehip_rccl_attach)");
    const int G = h->comm_ranks, me = h->comm_rank;
    if (root < 0 || root >= G) return fail(PIEHIP_EINVAL, "gather_results: root outside the communicator");
    u32 lo, hi;
    rank_slice(b_total, me, G, &lo, &hi);
    if (hi - lo != (h->d_out ? h->b : 0u))
        return fail(PIEHIP_EINVAL, "gather_results: this handle does not evaluate its rank's slice of the bin layers (piehip_rccl_bin_slice)");
    if (me == root && !d_out) return fail(PIEHIP_EINVAL, "gather_results: the root needs a destination");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    HIPCHK(hipSetDevice(h->device));
    const size_t row = (size_t)h->nq * 2 * h->LN();   // words per bin layer: its nq result ciphertexts
    const ncclComm_t comm = (ncclComm_t)h->comm;
    // exact sizes, no padding: the root posts one receive per peer straight into that peer's rows of d_out, every other rank one
    // send of its result buffer; the root's own rows are a device copy.  Each slice crosses its own xGMI link once.
    NCCLCHK(R, R->GroupStart());
    ncclResult_t gr = ncclSuccess;
    if (me == root) {
        for (int r = 0; r < G && gr == ncclSuccess; r++) {
            u32 rlo, rhi;
            rank_slice(b_total, r, G, &rlo, &rhi);
            if (r == me || rhi == rlo) continue;
            gr = R->Recv((u64 *)d_out + (size_t)rlo * row, (size_t)(rhi - rlo) * row, ncclUint64, r, comm, h->stream);
        }
    } else if (hi > lo) {
        gr = R->Send(h->d_out, (size_t)(hi - lo) * row, ncclUint64, root, comm, h->stream);
    }
    const ncclResult_t ge = R->GroupEnd();
    if (gr != ncclSuccess || ge != ncclSuccess)
        return fail(PIEHIP_EHIP, std::string("gather_results: ") + R->GetErrorString(gr != ncclSuccess ? gr : ge));
    if (me == root && hi > lo)
        HIPCHK(hipMemcpyAsync((u64 *)d_out + (size_t)lo * row, h->d_out, (size_t)(hi - lo) * row * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
    return PIEHIP_OK;
}

int piehip_gather_results_host(piehip_handle h, uint32_t b_total, int root, uint64_t **results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->comm) return fail(PIEHIP_ESTATE, "gather_results: no communicator (piehip_rccl_init / piehip_rccl_attach)");
    const bool is_root = h->comm_rank == root;
    if (is_root && !results) return fail(PIEHIP_EINVAL, "gather_results: the root needs a destination");
    if (is_root) {
        HIPCHK(hipSetDevice(h->device));
        const size_t words = (size_t)b_total * h->nq * 2 * h->LN();
        if (h->gather_words != words) {   // first use, or another shape
            HIPCHK(hipStreamSynchronize(h->stream));
            dev_free(&h->d_gather);
            if (h->pin_gather) (void)hipHostFree(h->pin_gather);
            h->pin_gather = nullptr;
            h->gather_words = 0;
            int rc = dev_alloc(&h->d_gather, words);
            if (rc) return rc;
            HIPCHK(hipHostMalloc((void **)&h->pin_gather, words * sizeof(u64), hipHostMallocPortable));
            h->gather_words = words;
        }
    }
    int rc = piehip_gather_results(h, b_total, root, is_root ? h->d_gather : nullptr);
    if (rc) return rc;
    if (is_root) {
        HIPCHK(hipMemcpyAsync(h->pin_gather, h->d_gather, h->gather_words * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        *results = h->pin_gather;
    } else if (results) {
        *results = nullptr;
    }
    return PIEHIP_OK;
}

int piehip_rccl_broadcast(piehip_handle h, int root, void *d_buf, size_t bytes)
{
    NEED(h);
    if (!h->comm) return fail(PIEHIP_ESTATE, "rccl_broadcast: no communicator");
    if (!d_buf && bytes) return fail(PIEHIP_EINVAL, "null buffer");
    if (root < 0 || root >= h->comm_ranks) return fail(PIEHIP_EINVAL, "rccl_broadcast: root outside the communicator");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    HIPCHK(hipSetDevice(h->device));
    if (bytes) NCCLCHK(R, R->Broadcast(d_buf, d_buf, bytes, ncclChar, root, (ncclComm_t)h->comm, h->stream));
    return PIEHIP_OK;
}

int piehip_rccl_broadcast_query(piehip_handle h, int root)
{
    NEED(h);
    if (!h->comm) return fail(PIEHIP_ESTATE, "rccl_broadcast_query: no communicator");
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database before the query");
    if (root < 0 || root >= h->comm_ranks) return fail(PIEHIP_EINVAL, "rccl_broadcast_query: root outside the communicator");
    const Rccl *R = rccl();
    if (!R) return no_rccl();
    HIPCHK(hipSetDevice(h->device));
    // the root's staged uploads are on its stream (piehip_stage_*: in order); the collective is queued behind them.  A root
    // with a half-staged query is a call-order error everywhere (the other ranks would wait for a broadcast that never comes),
    // so it is checked before anything is queued.
    if (h->comm_rank == root) {
        if (!h->stage_open) return fail(PIEHIP_ESTATE, "rccl_broadcast_query: the root has no staged query");
        for (u32 q = 0; q < h->nq; q++) {
            if (!h->qstage[q].minus) return fail(PIEHIP_ESTATE, "rccl_broadcast_query: minus element not staged");
            for (u32 hf = 0; hf < h->K; hf++)
                if (!h->qstage[q].rows[hf]) return fail(PIEHIP_ESTATE, "rccl_broadcast_query: index matrix row not staged");
        }
    }
    const size_t iw = (size_t)h->K * h->E * 2 * h->LN(), mw = 2 * h->LN();
    const ncclComm_t comm = (ncclComm_t)h->comm;
    int rc;
    NCCLCHK(R, R->GroupStart());
    ncclResult_t gr = ncclSuccess;
    for (u32 q = 0; q < h->nq && gr == ncclSuccess; q++) {
        u64 *di = nullptr, *dm = nullptr;
        if ((rc = query_input_buffers(h, q, &di, &dm))) {
            (void)R->GroupEnd();
            return rc;
        }
        gr = R->Broadcast(di, di, iw, ncclUint64, root, comm, h->stream);
        if (gr == ncclSuccess) gr = R->Broadcast(dm, dm, mw, ncclUint64, root, comm, h->stream);
    }
    const ncclResult_t ge = R->GroupEnd();
    if (gr != ncclSuccess || ge != ncclSuccess)
        return fail(PIEHIP_EHIP, std::string("rccl_broadcast_query: ") + R->GetErrorString(gr != ncclSuccess ? gr : ge));
    // every rank now evaluates the received copy: as if the query had been staged here
    h->stage_open = false;
    h->d_idx = h->d_idx_own;
    h->d_minus = h->d_minus_own;
    for (u32 q = 1; q < h->nq; q++) h->bq_idx[q] = h->bq_idx_own[q], h->bq_minus[q] = h->bq_minus_own[q];
    return PIEHIP_OK;
}

}  // extern "C"
