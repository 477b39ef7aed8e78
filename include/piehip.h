/*
 * piehip.h -- C ABI of libpiehip.so: the MI355X (gfx950) implementation of the server-side
 * batched-FHE private-indexed-equality evaluation of SAP/nested-hashing-psi.
 *
 * Drop-in boundary.  The reference operator is the concrete class
 *     class BatchedFHEHIPPIE            src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.hpp:18-49
 * constructed at src/Server/FHE/BatchedFHEPSIServer.cpp:86 and driven at :101-103,108 in the order
 * setMinusCompareElement, setIndex, run, getResultList.  Every arithmetic instruction of run()
 * (BatchedFHEHIPPIE.cpp:88-129) is an OpenFHE call on lbcrypto::Ciphertext<DCRTPoly> /
 * lbcrypto::Plaintext objects; a DCRTPoly is L contiguous uint64_t[N] limbs ("towers") in
 * EVALUATION format, so the ABI below sees only such limb arrays: plain pointers and sizes, no
 * OpenFHE, no torch types.  INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every array is uint64_t, C-contiguous, limb-major [..][limb][N], residues canonical in
 *     [0, modulus); ciphertext/plaintext/key polynomials are in EVALUATION (NTT, bit-reversed) format,
 *     exactly as OpenFHE stores them after Encrypt / MakePackedPlaintext+SetFormat(EVALUATION).
 *   - every function returns 0 on success or a negative PIEHIP_E* code and never throws;
 *     piehip_last_error() gives the text (thread-local).  The C++ facade converts codes back into
 *     the exception types the reference throws (BatchedFHEHIPPIE.cpp:13-21: std::invalid_argument).
 *   - a handle is not thread-safe: one host thread per handle, as the reference uses one
 *     BatchedFHEHIPPIE per server process (BatchedFHEPSIServer.hpp:23).
 *   - host pointers are copied on load/set; *_device variants take pointers to HBM the caller owns
 *     (they must stay valid until the next run() has completed).
 */
#ifndef PIEHIP_H
#define PIEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIEHIP_OK 0
#define PIEHIP_EINVAL (-1)   /* bad argument (reference: std::invalid_argument) */
#define PIEHIP_ESTATE (-2)   /* call order violated (run before keys/DB/inputs are set) */
#define PIEHIP_EHIP (-3)     /* HIP runtime error (no device, allocation, launch) */
#define PIEHIP_ENOMEM (-4)

typedef struct piehip_ctx *piehip_handle;

/* 100 = rounds 1-4; 101 (round 5) adds piehip_profile_read_n, piehip_set/get_transform_slots, piehip_upload_turn_wait,
 * piehip_set_host_path_timing / piehip_host_path_times, piehip_rccl_wait / _abort / _agree; nothing of 100 changed its signature or
 * its meaning (piehip_profile_read keeps writing the twelve kernel classes of version 100). */
int piehip_version(void);
const char *piehip_last_error(void);

/* Default RNS chain: the largest primes < 2^60 that are 1 mod 2N, descending -- L for Q, then L+1
 * more for the auxiliary basis P of the HPS multiplication (what GenCryptoContext derives for the
 * reference at src/Client/FHE/BatchedFHEPSIClient.cpp:72-78). */
int piehip_default_moduli(uint32_t N, uint32_t L, uint64_t *q /*[L]*/, uint64_t *p /*[L+1]*/);

/* Replaces the CryptoContext reference held at BatchedFHEHIPPIE.hpp:21 (cryptoContext member):
 * ring dimension N, L primes q (basis Q), L+1 primes p (basis P; NULL,NULL = default chain),
 * plaintext modulus t (GetPlaintextModulus, BatchedFHEHIPPIE.cpp:43).
 * device: HIP device ordinal; stream: a hipStream_t to enqueue on, or NULL for a private stream. */
int piehip_create(piehip_handle *out, uint32_t N, uint32_t L, uint64_t t, const uint64_t *q, const uint64_t *p,
                  int device, void *stream);
int piehip_destroy(piehip_handle h);

/* parameter read-back (moduli: q_0..q_{L-1}, p_0..p_L, t = 2L+2 entries) */
int piehip_get_moduli(piehip_handle h, uint64_t *out);
int piehip_get_root(piehip_handle h, uint32_t mod_index, uint64_t *psi);
int piehip_get_twiddles(piehip_handle h, uint32_t mod_index, uint64_t *fwd /*[N]*/, uint64_t *inv /*[N]*/);
int piehip_get_slot_positions(piehip_handle h, uint32_t *pos /*[N]*/);

/* Replaces the EvalMult key the context holds after DeserializeEvalMultKey
 * (src/Server/FHE/BatchedFHEPSIServer.cpp:49): BV key, one digit per RNS limb,
 * evk[L digits][2 (b,a)][L limbs][N]. */
int piehip_load_relin_key(piehip_handle h, const uint64_t *evk);

/* Replaces the constructor's packed database (BatchedFHEHIPPIE.cpp:37-82):
 *   pts   [K][b][E][L][N]  vectorizedHCT      (BatchedFHEHIPPIE.hpp:23), EVALUATION format
 *   masks [b][L][N]        preCalcRandomMask  (BatchedFHEHIPPIE.hpp:27)
 * b is the number of bin layers THIS handle evaluates (a shard of the reference's eachBinSize). */
int piehip_load_db(piehip_handle h, uint32_t K, uint32_t b, uint32_t E, const uint64_t *pts, const uint64_t *masks);
/* Same, from raw slot values: the device performs MakePackedPlaintext (BatchedFHEHIPPIE.cpp:68,81):
 *   slots [K][b][E][B] int64 (negative = t-|v|), mask_slots [b][B]; B <= N slots used. */
int piehip_load_db_slots(piehip_handle h, uint32_t K, uint32_t b, uint32_t E, uint32_t B, const int64_t *slots,
                         const int64_t *mask_slots);

/* The server's whole offline phase on the device (src/Server/FHE/BatchedFHEPSIServer.cpp:75-90):
 * HierarchicalCuckooHashTable::insertAll (HierarchicalCuckooHashTable.cpp:55-72: k outer tabulation hashes into e
 * positions, each a blocked Cuckoo table K x b x E with inner hash ids k..k+K-1) followed by the BatchedFHEHIPPIE
 * constructor (BatchedFHEHIPPIE.cpp:23-82: bin-layer shuffle, gather into K*b*E packed plaintexts of B = k*e slots,
 * b mask plaintexts).  items[n]: non-zero values < t (0 is the empty-cell sentinel, CuckooHashTable.cpp:89-90).
 * hash_seed seeds TabulationHashing (TabulationHashing.cpp:16-36: std::mt19937 + uniform_int_distribution<uint64_t>);
 * the reference seeds evictions, shuffle and masks from std::random_device -- here they are explicit.
 * Errors: PIEHIP_EINVAL for bad sizes / items >= t, PIEHIP_EHASH when an insertion fails
 * (the reference throws runtime_error("(Blocked) Cuckoo hashing error"), CuckooHashTable.cpp:113). */
#define PIEHIP_EHASH (-5)
int piehip_build_db(piehip_handle h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b,
                    uint32_t E, uint64_t hash_seed, uint64_t evict_seed, uint64_t shuffle_seed, uint64_t mask_seed);
/* Sharded server: the same offline phase, but this handle keeps -- gathers, encodes, evaluates -- only the bin
 * layers [bin_lo, bin_hi) of the b the table has.  The table itself is built and shuffled whole, so handles given the same
 * seeds hold slices of one and the same database, and the masks of layer beta are those the unsharded call draws for it. */
int piehip_build_db_bins(piehip_handle h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b,
                         uint32_t E, uint64_t hash_seed, uint64_t evict_seed, uint64_t shuffle_seed, uint64_t mask_seed,
                         uint32_t bin_lo, uint32_t bin_hi);
/* Allocates, ahead of the offline phase, everything piehip_build_db(_bins) of this shape and a query need (database, run
 * workspace, hash table, hashing / packing scratch, input buffers), so that the timed phases do not call hipMalloc.  The
 * reference server knows these sizes when it is constructed (HashTableParameter, BatchedFHEPSIServer.hpp:24).  Optional. */
int piehip_reserve(piehip_handle h, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, uint32_t bin_lo,
                   uint32_t bin_hi);
/* Only the BatchedFHEHIPPIE constructor (BatchedFHEHIPPIE.cpp:23-82) on the device, for a hash table the caller
 * built itself (the reference's HierarchicalCuckooHashTable): tbl[k][e][K][b][E] = hierarchicalCuckooTable[i][p]
 * .cuckooTable[h][bin][j]; shuffles the bin layers, gathers, draws the masks and encodes. */
int piehip_load_db_table(piehip_handle h, const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                         uint64_t shuffle_seed, uint64_t mask_seed);
int piehip_load_db_table_bins(piehip_handle h, const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E,
                              uint64_t shuffle_seed, uint64_t mask_seed, uint32_t bin_lo, uint32_t bin_hi);
/* the table built by piehip_build_db, after the bin shuffle: tbl[k][e][K][b][E] (hierarchicalCuckooTable[i][p].cuckooTable) */
int piehip_get_hash_table(piehip_handle h, uint64_t *tbl);
/* TabulationHashing::hashWithIndicator for n inputs (host-side; no device needed) */
int piehip_tabulation_hash(uint64_t hash_seed, uint32_t nfun, uint32_t hf, const uint64_t *x, size_t n, uint64_t *out);
/* the client's Cuckoo table (client harness; host-side): k single-layer tables of e positions, hash functions 0..k-1 of the
 * nfun-function family, items inserted in order with the reference's eviction walk (BatchedFHEPSIClient.cpp:97-99,109;
 * CuckooHashTable.cpp:72-114).  table[k][e], 0 = empty.  PIEHIP_EHASH when an insertion fails after 1000 retries. */
int piehip_client_cuckoo_table(uint64_t hash_seed, uint32_t nfun, uint32_t k, uint32_t e, const uint64_t *items, size_t n,
                               uint64_t *table);

/* setIndex (BatchedFHEHIPPIE.hpp:40-43): idx[K][E][2][L][N];
 * setMinusCompareElement (BatchedFHEHIPPIE.hpp:45-48): minus[2][L][N]. */
int piehip_set_index(piehip_handle h, const uint64_t *idx);
int piehip_set_minus(piehip_handle h, const uint64_t *minus);
/* the same for inputs that already live in HBM (no copy is taken).  The arrays must have been written on the stream
 * given to piehip_create, or be complete, at the time of the call; call again whenever their contents change, so that
 * the next run() waits for the writer (see "Stream order" below). */
int piehip_set_index_device(piehip_handle h, const void *d_idx);
int piehip_set_minus_device(piehip_handle h, const void *d_minus);

/* run() (BatchedFHEHIPPIE.cpp:88-129): enqueue the whole evaluation on the handle's stream.
 * Asynchronous; piehip_sync() or piehip_get_results() waits for it.
 * K = 1 (a database loaded with piehip_load_db / _slots; the hashing entry points refuse it as CuckooHashTable.cpp:39-42 does):
 * multipliedResult is the inner product itself (.cpp:117-120), so run() is stage A and the mask multiply (.cpp:126); no key needed. */
int piehip_run(piehip_handle h);
/* the same, with the result ciphertexts written straight into caller-owned HBM d_results[b][2][L][N] (e.g. the RCCL
 * gather buffer; d_results[b][nq][2][L][N] for a query batch) instead of the handle's result buffer */
int piehip_run_into(piehip_handle h, void *d_results);
/* The online phase in one call, as the reference server times it (BatchedFHEPSIServer.cpp:98-108): setMinusCompareElement +
 * setIndex + run + getResultList with the query in HOST memory.  The index matrix is uploaded one inner hash function (row of
 * E ciphertexts) at a time on a copy queue, stage A of row h starts as soon as that row has landed, and each queue group's
 * slice of the result list is downloaded as soon as the group is done.  Synchronous: results (may be NULL) are complete on
 * return.  Fastest with piehip_host_buffers' pinned arrays (a deserialiser writes the towers straight into them); any host
 * pointers work. */
int piehip_run_host(piehip_handle h, const uint64_t *idx /*[K][E][2][L][N]*/, const uint64_t *minus /*[2][L][N]*/,
                    uint64_t *results /*[b][2][L][N]*/);
/* The same in two halves, for a server that keeps several queries in flight (one handle per query slot, see
 * piehip_attach_database): _async returns once the uploads, the evaluation and the downloads are queued -- with page-locked
 * arrays (piehip_host_buffers) nothing in it waits for the device -- and _wait blocks until `results` is complete.  While
 * slot A evaluates, slot B's query crosses PCIe: the path is bound by the 29 MiB upload per query, not by upload + run +
 * download.  The arrays must stay valid and unchanged until _wait returns.
 * A handle with a batch of nq queries (piehip_set_query_batch) takes idx[nq][K][E][2][L][N], minus[nq][2][L][N] and writes
 * results[b][nq][2][L][N]. */
int piehip_run_host_async(piehip_handle h, const uint64_t *idx, const uint64_t *minus, uint64_t *results);
int piehip_run_host_wait(piehip_handle h);
/* page-locked staging arrays owned by the handle (valid until the database shape or the batch size changes, or the handle is
 * destroyed).  _q: the staging of query q of a batch -- every query has its own index matrix and minus element; `results` is the
 * one result array of the handle, [b][nq][2][L][N] (the nq result ciphertexts of a bin layer are adjacent), for every q. */
int piehip_host_buffers(piehip_handle h, uint64_t **idx, uint64_t **minus, uint64_t **results);
int piehip_host_buffers_q(piehip_handle h, uint32_t q, uint64_t **idx, uint64_t **minus, uint64_t **results);
/* piehip_run_host_async piece by piece, for a server that receives the query message by message -- one message per ciphertext,
 * the minus element first (BatchedFHEPSIServer.cpp:94-95,114-141): every piece starts its upload as soon as the deserialiser has
 * written it (into the page-locked arrays above, or any host memory that stays valid until piehip_run_host_wait), so the 29 MiB of
 * a C3 query cross PCIe while the remaining messages are still arriving, i.e. before the reference's timer starts (.cpp:98-99).
 *   piehip_stage_minus(_q)      the minus element [2][L][N] (of query q of the batch; pieces of different queries in any order)
 *   piehip_stage_index_row(_q)  row `row` (one inner hash function) of the index matrix, [E][2][L][N]
 *   piehip_stage_index_ct_q     one ciphertext (row, j) of the index matrix, [2][L][N] -- one message of the reference's receive loop
 *                               (.cpp:124-141): when the timer starts (.cpp:98) at most the last message's megabyte is still on
 *                               its way, not the last row's fourteen
 *   piehip_run_staged           setMinusCompareElement + setIndex + run + getResultList on the staged pieces: stage A of row h waits
 *                               for row h of every query of the batch only; results ([b][nq][2][L][N], may be NULL) are complete
 *                               after piehip_run_host_wait.
 *   piehip_stage_reset          drops a partial staging sequence (a receive failed between two pieces): the next piece begins a
 *                               new one.  Staging a piece again before the run simply replaces it.
 * PIEHIP_ESTATE when a piece is missing.  piehip_run_host_async is exactly: stage_minus, stage_index_row for every row, run_staged.
 * Queries of one device go up one after the other, in the order they were staged: the first piece of a staging sequence waits
 * (on the host) until the pieces other handles staged before it have left host memory -- uploads that share the link finish
 * together and the slots fall into lock-step (piehip_host.cpp). */
int piehip_stage_minus(piehip_handle h, const uint64_t *minus);
int piehip_stage_index_row(piehip_handle h, uint32_t row, const uint64_t *row_data);
int piehip_stage_minus_q(piehip_handle h, uint32_t q, const uint64_t *minus);
int piehip_stage_index_row_q(piehip_handle h, uint32_t q, uint32_t row, const uint64_t *row_data);
int piehip_stage_index_ct_q(piehip_handle h, uint32_t q, uint32_t row, uint32_t j, const uint64_t *ct);
int piehip_stage_reset(piehip_handle h);
int piehip_run_staged(piehip_handle h, uint64_t *results);
/* how long this handle's staging sequences waited (on the host) for their turn on the device's link: the last one, all of them,
 * and how many waited at all.  A wait that reaches 5 s returns PIEHIP_EHIP from the staging call instead of going on silently. */
int piehip_upload_turn_wait(piehip_handle h, double *last_ms, double *total_ms, uint64_t *waits);
/* measurement: with timing on, every host-memory query records three events on the handle's stream (first staged piece, uploads
 * handed over, results down); after piehip_run_host_wait piehip_host_path_times gives the last query's upload time and the rest
 * (evaluation + the result list's way down) -- which side of a slow query is slow */
int piehip_set_host_path_timing(piehip_handle h, int on);
int piehip_host_path_times(piehip_handle h, double *upload_ms, double *rest_ms);
/* Stream order.  run() works on the handle's own queues (piehip_set_run_streams); it waits for the handle's stream
 * where it must (new inputs; the result buffer), and the handle's stream waits for the run in the next call of any
 * other entry point -- piehip_join() does only that, piehip_sync() also blocks the host.  Work queued on the handle's
 * stream after such a call sees the results; consecutive run() calls pipeline. */
int piehip_join(piehip_handle h);
int piehip_sync(piehip_handle h);
/* run() spreads the (independent) bin layers over up to n HIP streams of the handle so that the partial workgroup rounds of
 * one group's launches are filled by another group's; n = 1 serialises everything on the handle's stream (per-kernel
 * timing), 0 = the default (2).  The results are complete on the handle's stream either way. */
int piehip_set_run_streams(piehip_handle h, uint32_t n);
/* The transforms are persistent grids that fill every workgroup slot of the device (two per CU) for 60-100 us at a time, with the
 * CU's whole register file.  A sharded server's RCCL kernels (broadcast of the next query, gather of the previous results) need
 * CUs during exactly those launches: n > 0 caps the transforms' grids at n workgroups (n < 2 x CUs leaves (2 x CUs - n) / 2 CUs
 * free; results are bit-identical for every n), 0 = the default, every slot.  Applies to this handle's later runs. */
int piehip_set_transform_slots(piehip_handle h, uint32_t n);
int piehip_get_transform_slots(piehip_handle h, uint32_t *n /* the cap, 0 = none */, uint32_t *device_slots /* 2 x CUs */);
/* on != 0: run() / run_into() replay one captured hipGraph (all launches of both queue groups, forked from and joined back to
 * the handle's stream) instead of enqueueing ~26 launches; re-captured when the input arrays, the result buffer or the queue
 * count change.  Amortises the launch path when a handle evaluates few bin layers (one rank's share of a sharded server);
 * consecutive runs then do not overlap each other.  Off by default. */
int piehip_set_graph(piehip_handle h, int on);
/* A further query slot on the same database.  `h` (same device, same parameters) takes `owner`'s relinearisation key, packed
 * database and masks by reference -- nothing is copied -- and gets a run() workspace of its own, so that queries set and
 * run on the two handles (each on its own stream) overlap: one query's plaintext-ciphertext stage is HBM-bound while the
 * other's transforms are ALU-bound.  The reference operator evaluates one query at a time (BatchedFHEHIPPIE.cpp:88-129);
 * this is how a server with several clients keeps the GPU full.  While handles are attached, `owner` refuses (PIEHIP_ESTATE)
 * to be destroyed, to load or build a database of any shape, to reload its key and to attach itself elsewhere: the attached
 * handles read those buffers on streams of their own.  Loading a database into `h` itself returns it to a private copy
 * (load_relin_key is refused while attached). */
int piehip_attach_database(piehip_handle h, piehip_handle owner);
/* Query batches: one run() evaluates nq (1 .. 8) queries -- each with its own index matrix and minus element -- against the
 * handle's database.  The reference operator takes one query per run() (BatchedFHEHIPPIE.hpp:40-48, .cpp:88-129); a server
 * with several clients waiting hands them over together: stage A then reads every database plaintext once for the batch
 * instead of once per query (the database is 3/4 of that stage's traffic), and every later launch carries nq times as many
 * ciphertexts.  Each query's results are bit-identical to its own run().
 *   piehip_set_query_batch   sizes the workspace for nq queries per run() (default 1; may be called before or after the database
 *                            is loaded; synchronises the handle's stream)
 *   piehip_set_*_q           inputs of query q < nq (q = 0: the plain setters above)
 *   results                  piehip_run / piehip_run_into / piehip_get_results then use rows [b][nq][2][L][N]: the nq result
 *                            ciphertexts of a bin layer are adjacent
 *   piehip_load_relin_key_q  the EvalMult key of query q's client (BatchedFHEPSIServer.cpp:45-49: every client sends its own);
 *                            queries without one use the handle's key.  Sized by the batch: load again after piehip_set_query_batch
 * The host-memory path takes batches too: piehip_host_buffers_q / piehip_stage_*_q / piehip_run_staged / piehip_run_host* below.
 * Only the captured graph (piehip_set_graph) is limited to one query per run(). */
int piehip_set_query_batch(piehip_handle h, uint32_t nq);
int piehip_get_query_batch(piehip_handle h, uint32_t *nq);
int piehip_load_relin_key_q(piehip_handle h, uint32_t q, const uint64_t *evk /*[L][2][L][N]*/);
int piehip_set_index_q(piehip_handle h, uint32_t q, const uint64_t *idx /*[K][E][2][L][N]*/);
int piehip_set_minus_q(piehip_handle h, uint32_t q, const uint64_t *minus /*[2][L][N]*/);
int piehip_set_index_device_q(piehip_handle h, uint32_t q, const void *d_idx);
int piehip_set_minus_device_q(piehip_handle h, uint32_t q, const void *d_minus);
/* getResultList (BatchedFHEHIPPIE.hpp:35-38): out[b][2][L][N] (out[b][nq][2][L][N] for a query batch) */
int piehip_get_results(piehip_handle h, uint64_t *out);
/* device address of the result buffer [b][2][L][N] ([b][nq][2][L][N] for a query batch; valid until the database shape or the
 * batch size changes); for the RCCL gather */
int piehip_results_device(piehip_handle h, void **d_out);
/* enqueue a device-to-device copy of the results into caller-owned HBM (e.g. the RCCL gather buffer) */
int piehip_copy_results_device(piehip_handle h, void *d_dst);

/* ---- sharded server: one process per GPU, RCCL over xGMI ---------------------------------------------
 * The outer loop of run() over bin layers (BatchedFHEHIPPIE.cpp:91) has independent iterations: rank r of G evaluates the
 * bin layers [b r / G, b (r + 1) / G) of the same table (piehip_build_db_bins / piehip_load_db_table_bins with the same seeds)
 * and holds the EvalMult key(s).  Two exchanges per query, both queued on the handle's stream:
 *   piehip_rccl_broadcast_query  in: the query reaches the server on ONE rank -- the process that holds the client's socket
 *                                (BatchedFHEPSIServer.cpp:94-95,114-141).  That rank stages it (piehip_stage_*); then EVERY rank
 *                                calls this, and every rank's input buffers hold the query (all queries of a batch).
 *   piehip_gather_results        out: the only exchange of the evaluation itself.  After piehip_run on every rank, each rank's
 *                                result rows travel to `root` (the rank that answers the client: sendResult, .cpp:143-152) --
 *                                exact sizes, one transfer per rank, each over its own xGMI link.  d_out on the root:
 *                                [b_total][nq][2][L][N] in bin order, caller-owned HBM; ignored elsewhere.
 * RCCL is bound at run time (the copy already in the process, else /opt/rocm's): a one-GPU deployment never loads it.
 *   piehip_rccl_unique_id   ncclGetUniqueId: 128 bytes made on one rank and handed to the others by the caller's own means
 *   piehip_rccl_init        ncclCommInitRank on the handle's device (collective over the nranks processes); the handle owns it
 *   piehip_rccl_attach      or: a communicator (ncclComm_t) the caller made with the process's RCCL; not destroyed here
 *   piehip_rccl_bin_slice   the bin layers of rank `rank` (what the rank's database must be built for)
 *   piehip_rccl_broadcast   any device buffer from root to all (set-up traffic: key, seeds) */
#define PIEHIP_RCCL_ID_BYTES 128
int piehip_rccl_unique_id(void *id /*[PIEHIP_RCCL_ID_BYTES]*/);
int piehip_rccl_init(piehip_handle h, const void *id, int nranks, int rank);
int piehip_rccl_attach(piehip_handle h, void *nccl_comm, int nranks, int rank);
int piehip_rccl_destroy(piehip_handle h);
int piehip_rccl_bin_slice(uint32_t b_total, int nranks, int rank, uint32_t *bin_lo, uint32_t *bin_hi);
int piehip_rccl_broadcast(piehip_handle h, int root, void *d_buf, size_t bytes);
int piehip_rccl_broadcast_query(piehip_handle h, int root);
int piehip_gather_results(piehip_handle h, uint32_t b_total, int root, void *d_out);
/* the same, with the gathered list brought down to host memory on the root: *results (root only; NULL elsewhere) is a page-locked
 * array [b_total][nq][2][L][N] owned by the handle, complete after piehip_sync -- what sendResult (.cpp:143-152) reads */
int piehip_gather_results_host(piehip_handle h, uint32_t b_total, int root, uint64_t **results);
/* Nobody waits for the other ranks without a bound.  A collective completes when EVERY rank has queued its side; a rank that
 * never does (it died; it returned an error from one of the calls above before anything was queued) would leave its peers'
 * streams blocked for ever.  The reference ends the process on any error (BatchedFHEHIPPIE.cpp:15,20); a sharded server does the
 * same for the whole group:
 *   piehip_rccl_wait    piehip_sync with a time-out (milliseconds): polls the handle's stream and the communicator's asynchronous
 *                       error state; when the time is up or RCCL reports a failed peer it aborts the communicator (ncclCommAbort
 *                       releases the blocked stream) and returns PIEHIP_EHIP -- the handle has no communicator afterwards
 *                       (a communicator given with piehip_rccl_attach is only dropped: aborting it is its owner's business)
 *   piehip_rccl_abort   the same on purpose: a rank on its way out tears its side down, its peers' waits end at once
 *   piehip_rccl_agree   *all_ok = (every rank passed ok != 0): one all-reduced word + piehip_rccl_wait; called at the end of a phase
 *                       (database built, key loaded) so that a failure on one rank ends the session on all of them BEFORE anybody
 *                       enters a collective the failed rank will not join.  Not part of the per-query path. */
int piehip_rccl_wait(piehip_handle h, uint32_t timeout_ms);
int piehip_rccl_abort(piehip_handle h);
int piehip_rccl_agree(piehip_handle h, int ok, int *all_ok, uint32_t timeout_ms);

/* ---- the OpenFHE primitives under run(), exposed one by one for kernel-level parity tests ------
 * (host buffers in, host buffers out; synchronous) */
/* DCRTPoly::SetFormat on nlimbs limbs [nlimbs][N]; limb i uses modulus mod_base + (i % mod_count) */
int piehip_ntt(piehip_handle h, uint64_t *limbs, uint32_t nlimbs, uint32_t mod_base, uint32_t mod_count, int inverse);
/* EvalAdd(ct,ct) BatchedFHEHIPPIE.cpp:112,116 / EvalMult(ct,pt) :108,113,126 */
int piehip_eval_add(piehip_handle h, const uint64_t *x, const uint64_t *y, uint64_t *out);
int piehip_eval_mult_plain(piehip_handle h, const uint64_t *x, const uint64_t *pt, uint64_t *out);
/* EvalMult(ct,ct) BatchedFHEHIPPIE.cpp:123 (HPS P-over-Q tensor + BV relinearisation with the loaded
 * key); nct independent pairs x[nct][2][L][N], y[nct][2][L][N] -> out[nct][2][L][N].
 * relin=0 returns the 3-component tensor result out[nct][3][L][N] instead. */
int piehip_eval_mult(piehip_handle h, const uint64_t *x, const uint64_t *y, uint32_t nct, int relin, uint64_t *out);
/* EvalAtIndex-style rotation (reference call sites FHEHIPPIE.cpp:71,74 via EvalInnerProduct/EvalMerge;
 * NOT on the batched path): automorphism X -> X^g then key switch with rk[L][2][L][N] */
int piehip_eval_automorph(piehip_handle h, const uint64_t *x, uint32_t g, const uint64_t *rk, uint64_t *out);
/* MakePackedPlaintext for npt plaintexts: slots[npt][B] -> out[npt][L][N] EVALUATION format */
int piehip_encode(piehip_handle h, const int64_t *slots, uint32_t npt, uint32_t B, uint64_t *out);
/* base conversions of the HPS multiplication on npoly polynomials (COEFFICIENT format):
 * which = 0: Q -> QP centred extension        in[npoly][L][N]    -> out[npoly][2L+1][N]
 * which = 1: scale by P/Q into P, extend to QP in[npoly][L][N]    -> out[npoly][2L+1][N]
 * which = 2: scale by t/P from QP into Q      in[npoly][2L+1][N] -> out[npoly][L][N] */
int piehip_base_convert(piehip_handle h, int which, const uint64_t *in, uint32_t npoly, uint64_t *out);

/* ---- FHEHIPPIE: the rotation-based sibling operator ---------------------------------
 * Reference: src/Common/Crypto/PrivateIndexedEqualityCheck/FHEHIPPIE.{hpp,cpp}; one operator per client slot
 * (FHEHIPPIECollection, PIECollection.hpp); `npie` operators are evaluated as one batch here.
 * An operator holds a flat blocked Cuckoo table T[K][b][E] with b == E (FHEHIPPIE.cpp:13-16), packs row (hf, bin)
 * as the E table cells followed by a 1 (FHEHIPPIE.cpp:41-51), and evaluates per hash function
 *   EvalMult(EvalMerge_bin(EvalInnerProduct(index[hf], row[hf][bin], b)), mask[hf])        (FHEHIPPIE.cpp:61-77)
 * where EvalInnerProduct = EvalMult(ct, pt) + ceil(log2 b) rotate-by-2^r-and-add steps and EvalMerge masks
 * slot 0 of ciphertext i and rotates it by -i [OFHE-UNVERIFIED: OpenFHE is not part of the reference tree]. */
/* Galois element 5^index mod 2N of the row rotation by `index` (negative = to the right) */
int piehip_rotation_galois(piehip_handle h, int32_t index, uint32_t *g);
/* rotation keys (EvalSumKeyGen / EvalRotateKeyGen products, SimpleFHEPSIClient.cpp:80-89), BV layout as the
 * relinearisation key: keys[nkeys][L][2][L][N], indices[nkeys] signed rotation amounts.  run() needs
 * 2^r (r < ceil(log2 b)) and -1 .. -(b-1). */
int piehip_load_rotation_keys(piehip_handle h, uint32_t nkeys, const int32_t *indices, const uint64_t *keys);
/* constructor packing (FHEHIPPIE.cpp:23-59) for npie operators: slots[npie][K][b][E+1] (already in the bin
 * order the caller's permutation vector chose), masks[npie][K][b] in [1, t-1]; encoded on the device.
 * PIEHIP_EINVAL if b != E (the reference's invalid_argument). */
int piehip_fhepie_load_table(piehip_handle h, uint32_t npie, uint32_t K, uint32_t b, uint32_t E, const int64_t *slots,
                             const int64_t *masks);
/* setIndex (FHEHIPPIE.hpp:45-48): idx[npie][K][2][L][N] */
int piehip_fhepie_set_index(piehip_handle h, const uint64_t *idx);
/* run() (FHEHIPPIE.cpp:61-77), synchronous; getResultList (FHEHIPPIE.hpp:40-43): out[npie][K][2][L][N] in hash
 * function order (the caller applies its result permutation) */
int piehip_fhepie_run(piehip_handle h);
int piehip_fhepie_get_results(piehip_handle h, uint64_t *out);

/* ---- client-side harness -----------------------------------------------------------------------------
 * Not part of the server hot path: the client role of src/Client/FHE/BatchedFHEPSIClient.cpp, needed to
 * produce the hot path's inputs, to read its outputs and to measure the end-to-end PSI wall-clock
 * (SURVEY.md 8f-1).  Deterministic samplers (xoshiro256** streams seeded per call) run on the host, the
 * polynomial arithmetic on the device.  BFV conventions: secret key uniform ternary; noise centred
 * binomial (sigma 3.16); fresh ciphertext (c0, c1) = (-a s + e + round(Q m / t), a); all in EVALUATION format. */
/* KeyGen (BatchedFHEPSIClient.cpp:88): sk[L][N] */
int piehip_client_keygen(piehip_handle h, uint64_t seed, uint64_t *sk);
/* EvalMultKeyGen (BatchedFHEPSIClient.cpp:91): BV key, evk[L][2][L][N] */
int piehip_client_relin_keygen(piehip_handle h, const uint64_t *sk, uint64_t seed, uint64_t *evk);
/* EvalSumKeyGen / EvalRotateKeyGen (SimpleFHEPSIClient.cpp:80-89): BV key from s(X^g) to s for the row rotation
 * by `index`, rk[L][2][L][N] */
int piehip_client_rot_keygen(piehip_handle h, const uint64_t *sk, int32_t index, uint64_t seed, uint64_t *rk);
/* MakePackedPlaintext + Encrypt(secretKey, .) (BatchedFHEPSIClient.cpp:155-156,161-168) of nct slot vectors
 * slots[nct][B]; seeds[nct] one sampler seed per ciphertext; out[nct][2][L][N] */
int piehip_client_encrypt(piehip_handle h, const uint64_t *sk, const int64_t *slots, uint32_t nct, uint32_t B,
                          const uint64_t *seeds, uint64_t *out);
/* Decrypt + GetPackedValue (BatchedFHEPSIClient.cpp:249-265): ct[nct][2][L][N] -> slots[nct][B] (centred) */
int piehip_client_decrypt(piehip_handle h, const uint64_t *sk, const uint64_t *ct, uint32_t nct, uint32_t B, int64_t *slots);

/* ---- measurement ------------------------------------------------------------------------------
 * With profiling on, run() brackets every kernel launch with HIP events on the handle's stream.
 * piehip_profile_read returns, per kernel class, the launch count, total milliseconds, and the
 * algorithmic bytes (SURVEY.md 8d formulas) of the last run. */
#define PIEHIP_NKERNELS 13
enum {
    PIEHIP_K_STAGE_A = 0,   /* fused ct x pt multiply-accumulate + minus add  (A3+A4)          */
    PIEHIP_K_NTT_FWD = 1,   /* forward negacyclic NTT                          (A1)             */
    PIEHIP_K_NTT_INV = 2,   /* inverse negacyclic NTT                          (A1)             */
    PIEHIP_K_EXPAND = 3,    /* Q->QP extension and P/Q scaling                 (A6)             */
    PIEHIP_K_TENSOR = 4,    /* tensor product over QP                          (A5 step 4)      */
    PIEHIP_K_SCALE = 5,     /* scale-and-round by t/P                          (A6)             */
    PIEHIP_K_DIGITS = 6,    /* BV digit decomposition                          (A7)             */
    PIEHIP_K_RELIN = 7,     /* key-switch multiply-accumulate (+ mask multiply) (A7, A3)        */
    PIEHIP_K_MASK = 8,      /* final ct x pt mask multiply when not fused                       */
    PIEHIP_K_ENCODE = 9,    /* packed encoding                                 (A2)             */
    PIEHIP_K_AUTOMORPH = 10,/* automorphism permutation                        (A9)             */
    PIEHIP_K_OTHER = 11,
    PIEHIP_K_EVENT_PAIR = 12 /* no launch: the two events of a bracket back to back -- what the bracket itself reads on this stream */
};
/* times `iters` back-to-back NTT launches over nlimbs random limbs (moduli cycle over mod_count from 0);
 * flags: bit 0 = inverse transform, bit 1 = EVALUATION side in the library's internal lane order.
 * Returns the average milliseconds per launch.  Used by bench tooling to sweep batch sizes. */
int piehip_bench_ntt(piehip_handle h, uint32_t nlimbs, uint32_t mod_count, int flags, uint32_t iters, double *ms_per_launch);
int piehip_set_profiling(piehip_handle h, int on);
/* piehip_profile_read_n fills the first min(n, PIEHIP_NKERNELS) entries of the caller's arrays: pass the length the caller was
 * BUILT with, so that a library with more classes never writes past them (piehip_version() >= 101).
 * piehip_profile_read is the entry point of version 100 and keeps that version's contract: exactly PIEHIP_NKERNELS_V100 = 12
 * entries, the kernel classes only.  Class 12 (PIEHIP_K_EVENT_PAIR) is not a kernel -- it is what an event bracket reads with
 * nothing between its two events -- and must not be added to a sum of kernel times. */
#define PIEHIP_NKERNELS_V100 12
int piehip_profile_read_n(piehip_handle h, uint32_t n, uint32_t *launches /*[n]*/, double *ms /*[n]*/, double *alg_bytes /*[n]*/);
int piehip_profile_read(piehip_handle h, uint32_t *launches /*[12]*/, double *ms /*[12]*/, double *alg_bytes /*[12]*/);
const char *piehip_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif
