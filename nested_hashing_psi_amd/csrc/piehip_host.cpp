// piehip_host.cpp -- the host-memory path of a query (include/piehip.h: piehip_host_buffers, piehip_stage_*, piehip_run_staged,
// piehip_run_host*): the reference server holds the query as deserialised ciphertexts in host memory when its timer starts
// (src/Server/FHE/BatchedFHEPSIServer.cpp:94-108); here every piece crosses PCIe from page-locked staging as soon as it exists.
#include "piehip_ctx.hpp"

using namespace piehip;

extern "C" {

static int host_path_setup(piehip_ctx *h)
{
    if (h->nq > 1) return fail(PIEHIP_ESTATE, "the host-buffer path takes one query per run() (piehip_set_query_batch(h, 1))");
    if (!h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->ev_copy_gate) HIPCHK(hipEventCreateWithFlags(&h->ev_copy_gate, hipEventDisableTiming));
    if (!h->ev_minus_h2d) HIPCHK(hipEventCreateWithFlags(&h->ev_minus_h2d, hipEventDisableTiming));
    while (h->ev_h2d.size() < h->K) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ev_h2d.push_back(e);
    }
    return PIEHIP_OK;
}

int piehip_host_buffers(piehip_handle h, uint64_t **idx, uint64_t **minus, uint64_t **results)
{
    NEED_RO(h);
    if (!h->K) return fail(PIEHIP_ESTATE, "load the database first (the buffer sizes depend on K, E and b)");
    HIPCHK(hipSetDevice(h->device));
    const size_t iw = (size_t)h->K * h->E * 2 * h->LN(), rw = (size_t)h->b * 2 * h->LN();
    if (h->pin_idx && h->pin_idx_words != iw) {
        (void)hipHostFree(h->pin_idx);
        h->pin_idx = nullptr;
    }
    if (h->pin_res && h->pin_res_words != rw) {
        (void)hipHostFree(h->pin_res);
        h->pin_res = nullptr;
    }
    if (!h->pin_idx) HIPCHK(hipHostMalloc((void **)&h->pin_idx, iw * sizeof(u64), hipHostMallocDefault));
    if (!h->pin_minus) HIPCHK(hipHostMalloc((void **)&h->pin_minus, 2 * h->LN() * sizeof(u64), hipHostMallocDefault));
    if (!h->pin_res) HIPCHK(hipHostMalloc((void **)&h->pin_res, rw * sizeof(u64), hipHostMallocDefault));
    h->pin_idx_words = iw;
    h->pin_res_words = rw;
    // whoever asks for the staging arrays is about to run queries from host memory: create the copy queue, its events and the
    // run queues now (the offline phase), not inside the first timed query
    int rc = host_path_setup(h);
    if (rc) return rc;
    const u32 ng = run_queue_count(h);
    if (ng > 1 && (rc = ensure_run_queues(h, ng))) return rc;
    if (idx) *idx = h->pin_idx;
    if (minus) *minus = h->pin_minus;
    if (results) *results = h->pin_res;
    return PIEHIP_OK;
}

// One query's uploads, piece by piece (piehip_stage_*): the copy queue is gated once behind everything queued so far -- the uploads
// may not overtake a run that still reads the input buffers -- and every piece is one asynchronous copy from host memory.
static int stage_begin(piehip_ctx *h)
{
    if (!h->K || !h->d_db) return fail(PIEHIP_ESTATE, "run: database not loaded");
    HIPCHK(hipSetDevice(h->device));
    if (h->stage_open) return PIEHIP_OK;
    int rc = host_path_setup(h);
    if (rc) return rc;
    const size_t LN = h->LN(), row = (size_t)h->E * 2 * LN;
    if (!h->d_idx_own && (rc = dev_alloc(&h->d_idx_own, (size_t)h->K * row))) return rc;
    if (!h->d_minus_own && (rc = dev_alloc(&h->d_minus_own, 2 * LN))) return rc;
    join_pending(h);
    HIPCHK(hipEventRecord(h->ev_copy_gate, h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_copy_gate, 0));
    h->stage_open = true;
    h->staged_minus = false;
    h->staged_rows.assign(h->K, false);
    return PIEHIP_OK;
}

int piehip_stage_minus(piehip_handle h, const uint64_t *minus)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!minus) return fail(PIEHIP_EINVAL, "null input");
    int rc = stage_begin(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->d_minus_own, minus, 2 * h->LN() * sizeof(u64), hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(hipEventRecord(h->ev_minus_h2d, h->copy_stream));
    h->staged_minus = true;
    return PIEHIP_OK;
}

int piehip_stage_index_row(piehip_handle h, uint32_t row, const uint64_t *row_data)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!row_data) return fail(PIEHIP_EINVAL, "null input");
    int rc = stage_begin(h);
    if (rc) return rc;
    if (row >= h->K) return fail(PIEHIP_EINVAL, "stage_index_row: the index matrix has one row per inner hash function");
    const size_t words = (size_t)h->E * 2 * h->LN();
    HIPCHK(hipMemcpyAsync(h->d_idx_own + (size_t)row * words, row_data, words * sizeof(u64), hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(hipEventRecord(h->ev_h2d[row], h->copy_stream));
    h->staged_rows[row] = true;
    return PIEHIP_OK;
}

int piehip_run_staged(piehip_handle h, uint64_t *results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!h->K || !h->d_db) return fail(PIEHIP_ESTATE, "run: database not loaded");
    if (!h->d_evk && h->K > 1) return fail(PIEHIP_ESTATE, "run: relinearisation key not loaded");
    if (!h->stage_open || !h->staged_minus) return fail(PIEHIP_ESTATE, "run_staged: minus element not staged");
    for (u32 hf = 0; hf < h->K; hf++)
        if (!h->staged_rows[hf]) return fail(PIEHIP_ESTATE, "run_staged: index matrix row not staged");
    HIPCHK(hipSetDevice(h->device));
    const size_t LN = h->LN();
    h->stage_open = false;
    h->d_idx = h->d_idx_own;
    h->d_minus = h->d_minus_own;
    // the minus element enters at the end of every stage A launch: the handle's stream (and, through the fork, every queue)
    // waits for it; row h of the index matrix is waited for by stage A of row h only
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_minus_h2d, 0));
    mark_dirty(h);
    h->row_events = h->ev_h2d.data();
    int rc = piehip_run_into(h, h->d_out);
    h->row_events = nullptr;
    if (rc) return rc;
    if (results) {
        // every queue group's slice of the result list leaves as soon as that group is done
        if (h->pending_join) {
            const u32 ng = run_queue_count(h);
            u32 b0 = 0;
            for (u32 g = 0; g < ng; g++) {
                const u32 nb = run_group_size(h->b, ng, g);
                HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_join[g], 0));
                HIPCHK(hipMemcpyAsync(results + (size_t)b0 * 2 * LN, h->d_out + (size_t)b0 * 2 * LN, (size_t)nb * 2 * LN * sizeof(u64),
                                      hipMemcpyDeviceToHost, h->copy_stream));
                b0 += nb;
            }
        } else {
            HIPCHK(hipEventRecord(h->ev_copy_gate, h->stream));
            HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_copy_gate, 0));
            HIPCHK(hipMemcpyAsync(results, h->d_out, (size_t)h->b * 2 * LN * sizeof(u64), hipMemcpyDeviceToHost, h->copy_stream));
        }
    }
    return PIEHIP_OK;
}

int piehip_run_host_async(piehip_handle h, const uint64_t *idx, const uint64_t *minus, uint64_t *results)
{
    if (!h) return fail(PIEHIP_EINVAL, "null handle");
    if (!idx || !minus) return fail(PIEHIP_EINVAL, "null input");
    if (h->K && !h->d_evk && h->K > 1) return fail(PIEHIP_ESTATE, "run: relinearisation key not loaded");
    h->stage_open = false;  // a query of its own: pieces staged earlier and never run are dropped
    int rc = piehip_stage_minus(h, minus);
    const size_t row = (size_t)h->E * 2 * h->LN();
    for (u32 hf = 0; hf < h->K && !rc; hf++) rc = piehip_stage_index_row(h, hf, idx + (size_t)hf * row);
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
    if (h->copy_stream) HIPCHK(hipStreamSynchronize(h->copy_stream));
    join_pending(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return PIEHIP_OK;
}

int piehip_run_host(piehip_handle h, const uint64_t *idx, const uint64_t *minus, uint64_t *results)
{
    const int rc = piehip_run_host_async(h, idx, minus, results);
    return rc ? rc : piehip_run_host_wait(h);
}

}  // extern "C"
