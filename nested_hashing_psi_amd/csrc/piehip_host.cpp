// piehip_host.cpp -- the host-memory path of a query (include/piehip.h: piehip_host_buffers*, piehip_stage_*, piehip_run_staged,
// piehip_run_host*): the reference server holds the query as deserialised ciphertexts in host memory when its timer starts
// (src/Server/FHE/BatchedFHEPSIServer.cpp:94-108); here every piece crosses PCIe from page-locked staging as soon as it exists.
// A handle that evaluates a batch of nq queries per run() (piehip_set_query_batch) stages every query of the batch the same
// way -- the queries of a batch are different clients' (.cpp:94-95: one per connection), so their pieces arrive interleaved.
//
// Queues.  Everything of a query travels on the handle's OWN queues, in order: the uploads on the handle's stream, the evaluation
// behind them (run()'s queues fork from that stream), and each queue group's slice of the result list on the queue that computed
// it, straight behind its last kernel.  No copy stream, no cross-stream event waits.  Rounds 2-3 gave every handle a private copy
// stream; a rocprofv3 trace of a stream of queries over several query slots (profiles/r04/host_stream_trace.txt) showed why that
// ran at 0.72-0.86 ms per C3 query where the link's duplex rate allows 0.57: the HIP runtime multiplexes a process's streams
// onto four hardware queues, a stream's wait for another stream's event is a barrier packet in its hardware queue, and a
// barrier blocks every later packet of that hardware queue -- also those of the OTHER streams mapped to it.  One slot's "wait for
// my download" or "wait for my run" stalled another slot's uploads and kernels.  With one in-order chain per handle the slots
// only meet on the PCIe link and on the CUs.
//
// Upload order.  Uploads of different handles that are in flight together share the link, finish together, evaluate together and
// download together: a stream of queries over several slots falls into lock-step -- every slot uploading, then every slot
// computing, then every slot downloading, never one direction busy while the other is (same trace).  The queries of a device
// therefore go up one after the other at the full link rate: a staging sequence does not begin before the query another handle
// of the device handed over last (piehip_run_staged) has left host memory.  While query i + 1 crosses PCIe, query i evaluates
// and its results come down.  How the host learns "has left host memory":
//   * not from a stream-side event wait -- a barrier packet again: 0.64 -> 1.4 ms per query over three slots once the
//     process owns more than four streams;
//   * not from hipEventSynchronize on an event recorded behind the uploads -- with kernels queued behind that event it returns
//     when the stream's whole backlog is done (measured: the wait covered the other slot's evaluation and download too);
//   * from a word in page-locked memory that a one-thread kernel, queued between the uploads and the evaluation, sets to the
//     query's sequence number.  The staging host thread polls it.
#include "piehip_ctx.hpp"

#include <chrono>
#include <mutex>
#include <thread>

using namespace piehip;

namespace {
// the page-locked word of a handle, shared with whoever still polls it (a handle may be destroyed while another waits its turn)
struct UpWord {
    u64 *w = nullptr;
    ~UpWord()
    {
        if (w) (void)hipHostFree(w);
    }
};
// Per DEVICE (the upload order is a matter of one device's PCIe link; until r05 one process-wide mutex was held while polling, so a
// host thread waiting for device 0's word also blocked staging, hand-over and destruction of handles on every other device):
//   turn   held by the one staging thread that waits for its turn -- the others of this device queue up behind it, nobody jumps
//   m      guards `last`; never held while waiting
struct DeviceUploads {
    std::mutex turn, m;
    std::shared_ptr<UpWord> flag;   // the word of the handle that handed a query over last ...
    u64 seq = 0;                    // ... reads >= seq once that query's uploads are done
    const piehip_ctx *owner = nullptr;
};
std::mutex g_up_map_mutex;
std::map<int, std::unique_ptr<DeviceUploads>> g_up;   // per device; entries live as long as the process
DeviceUploads &device_uploads(int device)
{
    std::lock_guard<std::mutex> lock(g_up_map_mutex);
    std::unique_ptr<DeviceUploads> &p = g_up[device];
    if (!p) p.reset(new DeviceUploads);
    return *p;
}
struct UpWordHolder {   // piehip_ctx keeps the raw pointer (piehip_ctx.hpp is plain data); the owning reference lives here
    std::mutex m;
    std::map<const piehip_ctx *, std::shared_ptr<UpWord>> of;
} g_words;
}  // namespace

// the first piece of a staging sequence waits (on the host) for the query another handle handed over last on this device
static int upload_turn(piehip_ctx *h)
{
    if (!h->pin_up_flag) {
        // coherent: the word is written by a kernel (system-scope release) and polled by the host while that stream keeps running
        std::shared_ptr<UpWord> wd(new UpWord);
        HIPCHK(hipHostMalloc((void **)&wd->w, 64, hipHostMallocPortable | hipHostMallocCoherent));
        *wd->w = 0;
        h->pin_up_flag = wd->w;
        std::lock_guard<std::mutex> lock(g_words.m);
        g_words.of[h] = wd;
    }
    DeviceUploads &du = device_uploads(h->device);
    std::lock_guard<std::mutex> turn(du.turn);
    std::shared_ptr<UpWord> flag;
    u64 seq = 0;
    {
        std::lock_guard<std::mutex> lock(du.m);
        if (du.owner != h) flag = du.flag, seq = du.seq;
    }
    h->up_turn_wait_ns = 0;
    if (flag) {
        const auto t0 = std::chrono::steady_clock::now();
        bool late = false;
        while (__atomic_load_n((const volatile u64 *)flag->w, __ATOMIC_ACQUIRE) < seq) {
            std::this_thread::yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                late = true;
                break;
            }
        }
        const u64 ns = (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        h->up_turn_wait_ns = ns;
        h->up_turn_wait_total_ns += ns;
        h->up_turn_waits++;
        if (late) {
            // the other handle's stream never reached its flag kernel: forget that hand-over (the next sequence must not pay
            // five seconds again) and tell the caller -- the upload order is broken, and so, most likely, is the device
            std::lock_guard<std::mutex> lock(du.m);
            if (du.flag == flag && du.seq == seq) du.flag.reset(), du.owner = nullptr;
            return fail(PIEHIP_EHIP, "staging: the query another handle of this device handed over 5 s ago has not left host memory");
        }
    }
    return PIEHIP_OK;
}
// behind the staged pieces of a query, in front of its evaluation
static int upload_handed_over(piehip_ctx *h)
{
    void *dev = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dev, h->pin_up_flag, 0));
    (void)hipGetLastError();
    launch_host_flag((u64 *)dev, h->up_seq + 1, h->stream);
    HIPCHK(hipGetLastError());   // a launch that failed never sets the word: nobody may be told to wait for it
    ++h->up_seq;
    std::shared_ptr<UpWord> wd;
    {
        std::lock_guard<std::mutex> lock(g_words.m);
        wd = g_words.of[h];
    }
    DeviceUploads &du = device_uploads(h->device);
    std::lock_guard<std::mutex> lock(du.m);
    du.flag = wd, du.seq = h->up_seq, du.owner = h;
    return PIEHIP_OK;
}

namespace piehip {

void free_host_path(piehip_ctx *h)
{
    if (h->pin_up_flag) {
        {
            DeviceUploads &du = device_uploads(h->device);
            std::lock_guard<std::mutex> lock(du.m);
            if (du.owner == h) du.flag.reset(), du.owner = nullptr;
        }
        // (a thread that is polling the word right now holds a reference of its own: the page is freed when it lets go)
        std::lock_guard<std::mutex> lock(g_words.m);
        g_words.of.erase(h);
        h->pin_up_flag = nullptr;
    }
    for (QueryStage &s : h->qstage) {
        if (s.pin_idx) (void)hipHostFree(s.pin_idx);
        if (s.pin_minus) (void)hipHostFree(s.pin_minus);
        s.pin_idx = s.pin_minus = nullptr;
    }
    if (h->pin_res) (void)hipHostFree(h->pin_res);
    h->pin_res = nullptr;
    for (hipEvent_t &e : h->hp_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    h->hp_timing = false;
}

}  // namespace piehip

extern "C" int piehip_upload_turn_wait(piehip_handle h, double *last_ms, double *total_ms, uint64_t *waits)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (last_ms) *last_ms = (double)h->up_turn_wait_ns * 1e-6;
    if (total_ms) *total_ms = (double)h->up_turn_wait_total_ns * 1e-6;
    if (waits) *waits = h->up_turn_waits;
    return PIEHIP_OK;
}

// One run()'s uploads, piece by piece (piehip_stage_*): every piece is one asynchronous copy from host memory on the handle's
// stream, i.e. behind whatever the handle still has in flight (an earlier run that reads the input buffers included).
static int stage_begin(piehip_ctx *h, u32 q)
{
    if (!h->K || !h->d_db) return fail(PIEHIP_ESTATE, "run: database not loaded");
    if (q >= h->nq) return fail(PIEHIP_EINVAL, "query index outside the batch (piehip_set_query_batch)");
    HIPCHK(hipSetDevice(h->device));
    join_pending(h);
    mark_dirty(h);
    if (h->stage_open) return PIEHIP_OK;
    int rc = upload_turn(h);
    if (rc) return rc;
    h->stage_open = true;
    if (h->hp_timing) {
        h->hp_ev_state = 0;
        if (hipEventRecord(h->hp_ev[0], h->stream) == hipSuccess) h->hp_ev_state = 1;
    }
    for (u32 i = 0; i < h->nq; i++) {
        h->qstage[i].minus = false;
        h->qstage[i].rows.assign(h->K, false);
        h->qstage[i].cts.clear();
    }
    return PIEHIP_OK;
}

extern "C" {

int piehip_host_buffers_q(piehip_handle h, uint32_t q, uint64_t **idx, uint64_t **minus, uint64_t **results)
{
    NEED_RO(h);
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database first (the buffer sizes depend on K, E and b)");
    if (q >= h->nq) return fail(PIEHIP_EINVAL, "query index outside the batch (piehip_set_query_batch)");
    HIPCHK(hipSetDevice(h->device));
    const size_t iw = (size_t)h->K * h->E * 2 * h->LN(), rw = (size_t)h->b * h->nq * 2 * h->LN();
    if (h->pin_idx_words != iw)  // another database shape: every query's index staging goes
        for (QueryStage &s : h->qstage)
            if (s.pin_idx) {
                (void)hipHostFree(s.pin_idx);
                s.pin_idx = nullptr;
            }
    if (h->pin_res && h->pin_res_words != rw) {
        (void)hipHostFree(h->pin_res);
        h->pin_res = nullptr;
    }
    // Portable: one process may drive several devices from the same staging arrays (host/ShardedBatchedFHEHIPPIE.hpp uploads
    // shard 0's arrays to every device).  Only what the caller asks for: such a shard needs a result array only.
    QueryStage &s = h->qstage[q];
    if (idx && !s.pin_idx) HIPCHK(hipHostMalloc((void **)&s.pin_idx, iw * sizeof(u64), hipHostMallocPortable));
    if (minus && !s.pin_minus) HIPCHK(hipHostMalloc((void **)&s.pin_minus, 2 * h->LN() * sizeof(u64), hipHostMallocPortable));
    if (results && !h->pin_res) HIPCHK(hipHostMalloc((void **)&h->pin_res, rw * sizeof(u64), hipHostMallocPortable));
    h->pin_idx_words = iw;
    h->pin_res_words = rw;
    // whoever asks for the staging arrays is about to run queries from host memory: create the device-side input buffers and the
    // run queues now (the offline phase), not inside the first timed query
    int rc;
    u64 *di = nullptr, *dm = nullptr;
    if ((rc = query_input_buffers(h, q, &di, &dm))) return rc;
    const u32 ng = run_queue_count(h);
    if (ng > 1 && (rc = ensure_run_queues(h, ng))) return rc;
    if (idx) *idx = s.pin_idx;
    if (minus) *minus = s.pin_minus;
    if (results) *results = h->pin_res;
    return PIEHIP_OK;
}

int piehip_host_buffers(piehip_handle h, uint64_t **idx, uint64_t **minus, uint64_t **results)
{
    return piehip_host_buffers_q(h, 0, idx, minus, results);
}

int piehip_stage_minus_q(piehip_handle h, uint32_t q, const uint64_t *minus)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!minus) return fail(PIEHIP_EINVAL, "null input");
    int rc = stage_begin(h, q);
    if (rc) return rc;
    u64 *di = nullptr, *dm = nullptr;
    if ((rc = query_input_buffers(h, q, &di, &dm))) return rc;
    HIPCHK(hipMemcpyAsync(dm, minus, 2 * h->LN() * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    h->qstage[q].minus = true;
    return PIEHIP_OK;
}

int piehip_stage_index_row_q(piehip_handle h, uint32_t q, uint32_t row, const uint64_t *row_data)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!row_data) return fail(PIEHIP_EINVAL, "null input");
    int rc = stage_begin(h, q);
    if (rc) return rc;
    if (row >= h->K) return fail(PIEHIP_EINVAL, "stage_index_row: the index matrix has one row per inner hash function");
    u64 *di = nullptr, *dm = nullptr;
    if ((rc = query_input_buffers(h, q, &di, &dm))) return rc;
    const size_t words = (size_t)h->E * 2 * h->LN();
    HIPCHK(hipMemcpyAsync(di + (size_t)row * words, row_data, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    h->qstage[q].rows[row] = true;
    return PIEHIP_OK;
}

int piehip_stage_index_ct_q(piehip_handle h, uint32_t q, uint32_t row, uint32_t j, const uint64_t *ct)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!ct) return fail(PIEHIP_EINVAL, "null input");
    int rc = stage_begin(h, q);
    if (rc) return rc;
    if (row >= h->K || j >= h->E) return fail(PIEHIP_EINVAL, "stage_index_ct: position outside the [K][E] index matrix");
    u64 *di = nullptr, *dm = nullptr;
    if ((rc = query_input_buffers(h, q, &di, &dm))) return rc;
    const size_t words = 2 * h->LN();
    HIPCHK(hipMemcpyAsync(di + ((size_t)row * h->E + j) * words, ct, words * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    QueryStage &s = h->qstage[q];
    if (s.cts.size() != (size_t)h->K * h->E) s.cts.assign((size_t)h->K * h->E, false);
    s.cts[(size_t)row * h->E + j] = true;
    bool all = true;
    for (u32 i = 0; i < h->E && all; i++) all = s.cts[(size_t)row * h->E + i];
    if (all) s.rows[row] = true;
    return PIEHIP_OK;
}

int piehip_stage_minus(piehip_handle h, const uint64_t *minus) { return piehip_stage_minus_q(h, 0, minus); }
int piehip_stage_index_row(piehip_handle h, uint32_t row, const uint64_t *row_data) { return piehip_stage_index_row_q(h, 0, row, row_data); }

int piehip_stage_reset(piehip_handle h)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    h->stage_open = false;  // copies already queued still land (in buffers nothing reads until they are staged again)
    for (QueryStage &s : h->qstage) {
        s.minus = false;
        s.rows.assign(s.rows.size(), false);
        s.cts.clear();
    }
    return PIEHIP_OK;
}

int piehip_run_staged(piehip_handle h, uint64_t *results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->K || !h->d_db) return fail(PIEHIP_ESTATE, "run: database not loaded");
    if (!run_keys_loaded(h)) return fail(PIEHIP_ESTATE, "run: relinearisation key not loaded");
    if (!h->stage_open) return fail(PIEHIP_ESTATE, "run_staged: query not staged");
    const u32 nq = h->nq;
    for (u32 q = 0; q < nq; q++) {
        if (!h->qstage[q].minus) return fail(PIEHIP_ESTATE, "run_staged: minus element not staged");
        for (u32 hf = 0; hf < h->K; hf++)
            if (!h->qstage[q].rows[hf]) return fail(PIEHIP_ESTATE, "run_staged: index matrix row not staged");
    }
    HIPCHK(hipSetDevice(h->device));
    h->stage_open = false;
    h->d_idx = h->d_idx_own;
    h->d_minus = h->d_minus_own;
    for (u32 q = 1; q < nq; q++) h->bq_idx[q] = h->bq_idx_own[q], h->bq_minus[q] = h->bq_minus_own[q];
    // the uploads are on the handle's stream and the run's queues start behind it (the inputs changed); every queue group's slice
    // of the result list leaves on that group's queue as soon as the group is done
    mark_dirty(h);
    int rc = upload_handed_over(h);
    if (rc) return rc;
    if (h->hp_timing && h->hp_ev_state == 1 && hipEventRecord(h->hp_ev[1], h->stream) == hipSuccess) h->hp_ev_state = 2;
    h->host_results = results;
    rc = piehip_run_into(h, h->d_out);
    h->host_results = nullptr;
    if (!rc && h->hp_timing && h->hp_ev_state == 2) {
        // the third event goes into the stream NOW (behind the join with the run's queues), not when the host comes back to wait:
        // what it stamps is then the device's time alone -- a host thread that is late does not show up in it
        join_pending(h);
        if (hipEventRecord(h->hp_ev[2], h->stream) == hipSuccess) h->hp_ev_state = 3;
    }
    return rc;
}

int piehip_run_host_async(piehip_handle h, const uint64_t *idx, const uint64_t *minus, uint64_t *results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!idx || !minus) return fail(PIEHIP_EINVAL, "null input");
    if (h->K && !run_keys_loaded(h)) return fail(PIEHIP_ESTATE, "run: relinearisation key not loaded");
    h->stage_open = false;  // queries of its own: pieces staged earlier and never run are dropped
    const size_t row = (size_t)h->E * 2 * h->LN();
    int rc = PIEHIP_OK;
    for (u32 q = 0; q < h->nq && !rc; q++) rc = piehip_stage_minus_q(h, q, minus + (size_t)q * 2 * h->LN());
    for (u32 q = 0; q < h->nq && !rc; q++)
        for (u32 hf = 0; hf < h->K && !rc; hf++) rc = piehip_stage_index_row_q(h, q, hf, idx + ((size_t)q * h->K + hf) * row);
    if (rc) {
        h->stage_open = false;
        return rc;
    }
    return piehip_run_staged(h, results);
}

int piehip_run_host_wait(piehip_handle h)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->device));
    join_pending(h);  // uploads, evaluation and downloads are all behind the handle's stream now
    HIPCHK(hipStreamSynchronize(h->stream));
    return PIEHIP_OK;
}

// Where a host-memory query spends its time ON THE DEVICE SIDE: events on the handle's stream at the first staged piece, behind the
// last upload (piehip_run_staged) and behind the last download (piehip_run_host_wait).  upload_ms is the time the query's pieces
// took to cross PCIe (from the moment the stream reached the first one), rest_ms the evaluation plus the result list's way down --
// all three stamped by the device in stream order, independent of when the host calls what.
int piehip_set_host_path_timing(piehip_handle h, int on)
{
    NEED_RO(h);
    HIPCHK(hipSetDevice(h->device));
    if (on && !h->hp_ev[0])
        for (hipEvent_t &e : h->hp_ev) HIPCHK(hipEventCreate(&e));
    h->hp_timing = on != 0;
    h->hp_ev_state = 0;
    return PIEHIP_OK;
}

int piehip_host_path_times(piehip_handle h, double *upload_ms, double *rest_ms)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->hp_timing || h->hp_ev_state != 3) return fail(PIEHIP_ESTATE, "host_path_times: no complete timed query (piehip_set_host_path_timing, then stage / run_staged / run_host_wait)");
    float a = 0, b = 0;
    HIPCHK(hipEventElapsedTime(&a, h->hp_ev[0], h->hp_ev[1]));
    HIPCHK(hipEventElapsedTime(&b, h->hp_ev[1], h->hp_ev[2]));
    if (upload_ms) *upload_ms = a;
    if (rest_ms) *rest_ms = b;
    return PIEHIP_OK;
}

int piehip_run_host(piehip_handle h, const uint64_t *idx, const uint64_t *minus, uint64_t *results)
{
    const int rc = piehip_run_host_async(h, idx, minus, results);
    return rc ? rc : piehip_run_host_wait(h);
}

}  // extern "C"
