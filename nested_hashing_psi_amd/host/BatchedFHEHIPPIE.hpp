// BatchedFHEHIPPIE.hpp -- C++ host facade over the C ABI (include/piehip.h).
//
// Same class shape as the reference operator
//     src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.hpp:18-49
// (constructor, run(), getResultList(), setIndex(&&), setMinusCompareElement()), the same call order
// the server uses (src/Server/FHE/BatchedFHEPSIServer.cpp:86,101-103,108) and the same error behaviour
// (std::invalid_argument for a stash or combined tables, BatchedFHEHIPPIE.cpp:13-21; std::runtime_error
// for everything the library reports at run time).
//
// The reference class is written against lbcrypto::Ciphertext<DCRTPoly> / lbcrypto::Plaintext.  OpenFHE is
// not available to this build, so ciphertexts are piehip::LimbCt here: the RNS towers of a DCRTPoly pair (EVALUATION
// format, uint64_t[N] each) as one flat array.  INTEGRATION.md gives the OpenFHE binding
// (DCRTPoly::GetElementAtIndex(i).GetValues() <-> limb arrays), which is the only code a maintainer adds.
//
// Host path.  The query lives in page-locked staging arrays owned by the library (piehip_host_buffers); setIndex /
// setMinusCompareElement copy each ciphertext there exactly once and start the upload of every piece as soon as it is complete
// (piehip_stage_*), run() is piehip_run_staged (+ wait), and the result list is read from the page-locked result array.  A
// deserialiser can skip the copy altogether: it writes the towers straight into indexStaging(h, j) / minusStaging() and calls
// stageIndexCiphertext(h, j) / stageMinus() (BatchedFHEPSIServer.hpp does; INTEGRATION.md section 2 shows it with OpenFHE).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/piehip.h"

namespace piehip {

// flat limb containers: the layout of a DCRTPoly's towers
struct LimbPt {                       // plaintext: [L][N]
    std::vector<uint64_t> limbs;
};
struct LimbCt {                       // ciphertext: [2][L][N]
    std::vector<uint64_t> limbs;
};

// What the reference reads from its HierarchicalCuckooHashTable in the constructor
// (BatchedFHEHIPPIE.cpp:13-21,37-41,48-66): sizes, the two argument checks, and the raw table
// tbl[k][e][K][b][E] (outer hash fn, outer position, inner hash fn, bin layer, inner position).
struct HashTableView {
    uint32_t numberOfSimpleTables = 0;      // k
    uint32_t eachSimpleTableSize = 0;       // e
    uint32_t numberOfCuckooTables = 0;      // K
    uint32_t eachBinSize = 0;               // b
    uint32_t eachCuckooTableSize = 0;       // E
    uint64_t serverStashSize = 0;
    bool simpleMultiTables = true, cuckooMultiTables = true;
    const uint64_t *table = nullptr;        // [k][e][K][b][E]
};

// Owner of the device context: the role lbcrypto::CryptoContext<DCRTPoly> plays for the reference
// (BatchedFHEPSIServer.hpp:21; created from the client's serialized context at .cpp:21-54).
class PieContext {
public:
    PieContext(uint32_t N, uint32_t L, uint64_t t, const uint64_t *q = nullptr, const uint64_t *p = nullptr, int device = 0,
               void *stream = nullptr)
        : N_(N), L_(L), t_(t)
    {
        check(piehip_create(&h_, N, L, t, q, p, device, stream));
    }
    ~PieContext() { piehip_destroy(h_); }
    PieContext(const PieContext &) = delete;
    PieContext &operator=(const PieContext &) = delete;

    // InsertEvalMultKey / DeserializeEvalMultKey (BatchedFHEPSIServer.cpp:49): evk[L][2][L][N]
    void setEvalMultKey(const uint64_t *evk) { check(piehip_load_relin_key(h_, evk)); }
    uint64_t GetPlaintextModulus() const { return t_; }
    uint32_t ringDimension() const { return N_; }
    uint32_t towers() const { return L_; }
    piehip_handle handle() const { return h_; }

    static void check(int rc)
    {
        if (rc == PIEHIP_OK) return;
        const std::string msg = piehip_last_error();
        if (rc == PIEHIP_EINVAL) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }

private:
    piehip_handle h_ = nullptr;
    uint32_t N_, L_;
    uint64_t t_;
};

class BatchedFHEHIPPIE {
public:
    // Seeds of the bin-layer shuffle (.cpp:23-35) and of the random masks (.cpp:72-82).  The masks are what hides
    // prod_h (item - x) of a non-matching slot from the client, so they must be secret: the default constructor draws
    // both seeds from std::random_device as the reference does (.cpp:25-26).  Fixed seeds are for parity tests only.
    struct Seeds {
        uint64_t shuffle, mask;
        static Seeds fromRandomDevice()
        {
            std::random_device rd;
            auto u64 = [&rd] { return ((uint64_t)rd() << 32) ^ (uint64_t)rd(); };
            return Seeds{u64(), u64()};
        }
    };

    // BatchedFHEHIPPIE(cryptoContext, pK, hct), BatchedFHEHIPPIE.cpp:9-86.  The public key is unused by the
    // reference constructor and run() (it is only stored, .hpp:22), so it does not appear here.
    BatchedFHEHIPPIE(PieContext &cryptoContext, const HashTableView &hct) : BatchedFHEHIPPIE(cryptoContext, hct, Seeds::fromRandomDevice()) {}
    // test-only: reproducible shuffle and masks
    BatchedFHEHIPPIE(PieContext &cryptoContext, const HashTableView &hct, const Seeds &testSeeds)
        : cc(cryptoContext)
    {
        const uint64_t shuffleSeed = testSeeds.shuffle, maskSeed = testSeeds.mask;
        if (hct.serverStashSize != 0) throw std::invalid_argument("Error, batched FHE PIE does not support a stash (yet).");
        if (!hct.simpleMultiTables || !hct.cuckooMultiTables)
            throw std::invalid_argument("Error, batched FHE PIE currently does not support combined tables.");
        K = hct.numberOfCuckooTables;
        b = hct.eachBinSize;
        E = hct.eachCuckooTableSize;
        const uint32_t k = hct.numberOfSimpleTables, e = hct.eachSimpleTableSize;
        const size_t B = (size_t)k * e;  // batch size, .cpp:41 (assumes simple multi table)
        if (B > cc.ringDimension()) throw std::invalid_argument("batch size exceeds the ring dimension");
        // bin-layer shuffle (.cpp:23-35), gather (.cpp:45-70), masks (.cpp:72-82) and MakePackedPlaintext (.cpp:68,81),
        // all on the device
        PieContext::check(piehip_load_db_table(cc.handle(), hct.table, k, e, K, b, E, shuffleSeed, maskSeed));
        initStaging();
    }

    // A further query slot on `database`'s packed table and on its context's key (piehip_attach_database): `cryptoContext`
    // is a second context with the same parameters, its own stream and run() workspace.  Queries set on the two operators
    // evaluate at the same time (enqueue() on each, then collect() on each); the reference operator has no counterpart --
    // it evaluates one query at a time -- and `database` must outlive the slot.
    BatchedFHEHIPPIE(PieContext &cryptoContext, const BatchedFHEHIPPIE &database) : cc(cryptoContext)
    {
        K = database.K, b = database.b, E = database.E;
        PieContext::check(piehip_attach_database(cc.handle(), database.cc.handle()));
        initStaging();
    }

    void run()  // BatchedFHEHIPPIE.cpp:88-129
    {
        enqueue();
        collect();
    }
    // run() in two halves: the evaluation and the download of the result list are asynchronous; collect() waits for them
    void enqueue()
    {
        // whatever happens below, the next query starts from a clean slate: a refused or failed run() must not leave half a
        // query counted (its pieces would never be staged again)
        struct Reset {
            BatchedFHEHIPPIE &o;
            bool ok = false;
            ~Reset()
            {
                o.clearStaged();
                if (!ok) piehip_stage_reset(o.cc.handle());
            }
        } reset{*this};
        if (minusStaged && rowsStaged == K) {
            PieContext::check(piehip_run_staged(cc.handle(), pinRes));
        } else if (minusStaged || rowsStaged || arrived) {
            throw std::runtime_error("run: setMinusCompareElement and setIndex must both precede run()");
        } else {
            // the query of the previous run() again (its inputs are still in HBM)
            PieContext::check(piehip_run(cc.handle()));
            rerun = true;
        }
        reset.ok = true;
    }
    void collect()
    {
        if (rerun) PieContext::check(piehip_get_results(cc.handle(), pinRes));
        else PieContext::check(piehip_run_host_wait(cc.handle()));
        rerun = false;
        listStale = true;
    }

    // .hpp:35-38.  The ciphertexts are materialised from the page-locked result array on the first call after a run() (the
    // reference's timer has stopped by then: BatchedFHEPSIServer.cpp:105-108); resultTowers(i) reads them in place.
    std::vector<LimbCt> &getResultList()
    {
        if (listStale) {
            const size_t ct = ctWords();
            for (uint32_t i = 0; i < b; i++) resultList[i].limbs.assign(pinRes + (size_t)i * ct, pinRes + (size_t)(i + 1) * ct);
            listStale = false;
        }
        return resultList;
    }
    const uint64_t *resultTowers(uint32_t i) const { return pinRes + (size_t)i * ctWords(); }  // [2][L][N], valid until the next run()

    void setIndex(std::vector<std::vector<LimbCt>> &&indexMatrix)  // .hpp:40-43, [K][E] ciphertexts
    {
        const size_t ct = ctWords();
        if (indexMatrix.size() != K) throw std::invalid_argument("index matrix must have one row per inner hash function");
        for (uint32_t h = 0; h < K; h++) {
            if (indexMatrix[h].size() != E) throw std::invalid_argument("index matrix row length must be eachCuckooTableSize");
            for (uint32_t j = 0; j < E; j++)
                if (indexMatrix[h][j].limbs.size() != ct) throw std::invalid_argument("ciphertext does not match the context");
        }
        restartIndex();  // the reference's setter overwrites the matrix (.hpp:40-43): a second call before run() replaces the first
        for (uint32_t h = 0; h < K; h++)
            for (uint32_t j = 0; j < E; j++) {  // one copy, straight into the staging array; row h uploads while row h + 1 is copied
                std::memcpy(indexStaging(h, j), indexMatrix[h][j].limbs.data(), ct * sizeof(uint64_t));
                stageIndexCiphertext(h, j);
            }
    }

    void setMinusCompareElement(LimbCt minusCompareElement)  // .hpp:45-48
    {
        if (minusCompareElement.limbs.size() != ctWords()) throw std::invalid_argument("ciphertext does not match the context");
        if (minusStaged) drainUploads();  // the previous element may still be crossing PCIe from this very array
        std::memcpy(minusStaging(), minusCompareElement.limbs.data(), ctWords() * sizeof(uint64_t));
        stageMinus();
    }

    // ---- zero-copy variant of the two setters, for a deserialiser ------------------------------------------------------------
    // Call restartIndex() before writing a NEW index matrix over one whose pieces were already handed over and not yet run
    // (setIndex does): the uploads in flight are waited for and every row is staged afresh.
    uint64_t *indexStaging(uint32_t h, uint32_t j) { return pinIdx + ((size_t)h * E + j) * ctWords(); }  // [2][L][N] of idx[h][j]
    uint64_t *minusStaging() { return pinMinus; }
    void restartIndex()
    {
        if (!arrived && !rowsStaged) return;
        drainUploads();
        std::fill(rowCount.begin(), rowCount.end(), 0u);
        rowsStaged = arrived = 0;
    }
    // ciphertext (h, j) has been written to indexStaging(h, j): its upload starts now
    void stageIndexCiphertext(uint32_t h, uint32_t j)
    {
        if (h >= K || j >= E) throw std::invalid_argument("index matrix position out of range");
        PieContext::check(piehip_stage_index_ct_q(cc.handle(), 0, h, j, indexStaging(h, j)));  // leaves at once
        arrived++;
        if (++rowCount[h] == E) rowsStaged++;
    }
    void stageMinus()
    {
        PieContext::check(piehip_stage_minus(cc.handle(), pinMinus));
        minusStaged = true;
    }

    friend class BatchedFHEHIPPIEQueryBatch;

protected:
    size_t ctWords() const { return 2 * (size_t)cc.towers() * cc.ringDimension(); }
    void initStaging()
    {
        PieContext::check(piehip_host_buffers(cc.handle(), &pinIdx, &pinMinus, &pinRes));
        resultList.resize(b);
        rowCount.assign(K, 0u);
    }
    void clearStaged()
    {
        minusStaged = false;
        rowsStaged = arrived = 0;
        std::fill(rowCount.begin(), rowCount.end(), 0u);
    }
    void drainUploads() { PieContext::check(piehip_run_host_wait(cc.handle())); }  // the copy queue is idle afterwards
    PieContext &cc;
    uint32_t K = 0, b = 0, E = 0;
    std::vector<LimbCt> resultList;
    uint64_t *pinIdx = nullptr, *pinMinus = nullptr, *pinRes = nullptr;  // page-locked, owned by the library
    std::vector<uint32_t> rowCount;
    uint32_t rowsStaged = 0, arrived = 0;
    bool minusStaged = false, rerun = false, listStale = false;
};

// Several queries per run() (piehip_set_query_batch): a server with clients waiting evaluates their queries together -- stage A
// then streams the packed database once for the batch, and every later launch carries nq times the ciphertexts.  The reference
// operator has one query per run() (BatchedFHEHIPPIE.hpp:40-48) and the reference server one client per process
// (BatchedFHEPSIServer.cpp:94-95); this class keeps the operator's call order per query:
//     setEvalMultKey(q, ..) once per client; then per batch setMinusCompareElement(q, ..), setIndex(q, ..) for every q < nq,
//     run(), getResultList(q).
// Every query's result list is bit-identical to what BatchedFHEHIPPIE::run() gives for that query alone under that client's key.
// Host path as in BatchedFHEHIPPIE: every query has page-locked staging of its own (piehip_host_buffers_q), a piece starts its
// upload when it is complete (piehip_stage_*_q; pieces of different queries in any order -- the clients' messages interleave),
// run() is piehip_run_staged + wait, and the result lists are read from the page-locked result array [b][nq].
// `cryptoContext` is a context of its own (same parameters as `database`'s, its own stream and workspace) attached to
// `database`'s packed table; `database` must outlive this object.
class BatchedFHEHIPPIEQueryBatch {
public:
    BatchedFHEHIPPIEQueryBatch(PieContext &cryptoContext, const BatchedFHEHIPPIE &database, uint32_t queriesPerRun)
        : cc(cryptoContext), K(database.K), b(database.b), E(database.E), nq(queriesPerRun)
    {
        PieContext::check(piehip_attach_database(cc.handle(), database.cc.handle()));
        init();
    }
    uint32_t queriesPerRun() const { return nq; }

    // the EvalMult key of query q's client, evk[L][2][L][N] (BatchedFHEPSIServer.cpp:45-49); queries without one use the key
    // of the context the database lives on
    void setEvalMultKey(uint32_t q, const uint64_t *evk) { PieContext::check(piehip_load_relin_key_q(cc.handle(), q, evk)); }

    void setIndex(uint32_t q, std::vector<std::vector<LimbCt>> &&indexMatrix)  // [K][E] ciphertexts of query q
    {
        const size_t ct = ctWords();
        checkQuery(q);
        if (indexMatrix.size() != K) throw std::invalid_argument("index matrix must have one row per inner hash function");
        for (uint32_t h = 0; h < K; h++) {
            if (indexMatrix[h].size() != E) throw std::invalid_argument("index matrix row length must be eachCuckooTableSize");
            for (uint32_t j = 0; j < E; j++)
                if (indexMatrix[h][j].limbs.size() != ct) throw std::invalid_argument("ciphertext does not match the context");
        }
        restartIndex(q);
        for (uint32_t h = 0; h < K; h++)
            for (uint32_t j = 0; j < E; j++) {
                std::memcpy(indexStaging(q, h, j), indexMatrix[h][j].limbs.data(), ct * sizeof(uint64_t));
                stageIndexCiphertext(q, h, j);
            }
    }
    void setMinusCompareElement(uint32_t q, const LimbCt &minusCompareElement)
    {
        checkQuery(q);
        if (minusCompareElement.limbs.size() != ctWords()) throw std::invalid_argument("ciphertext does not match the context");
        if (st[q].minusStaged) drainUploads();
        std::memcpy(minusStaging(q), minusCompareElement.limbs.data(), ctWords() * sizeof(uint64_t));
        stageMinus(q);
    }
    // zero-copy variant for a deserialiser (see BatchedFHEHIPPIE)
    uint64_t *indexStaging(uint32_t q, uint32_t h, uint32_t j) { return st[q].pinIdx + ((size_t)h * E + j) * ctWords(); }
    uint64_t *minusStaging(uint32_t q) { return st[q].pinMinus; }
    void restartIndex(uint32_t q)
    {
        checkQuery(q);
        if (!st[q].arrived && !st[q].rowsStaged) return;
        drainUploads();
        std::fill(st[q].rowCount.begin(), st[q].rowCount.end(), 0u);
        st[q].rowsStaged = st[q].arrived = 0;
    }
    void stageIndexCiphertext(uint32_t q, uint32_t h, uint32_t j)
    {
        checkQuery(q);
        if (h >= K || j >= E) throw std::invalid_argument("index matrix position out of range");
        PieContext::check(piehip_stage_index_ct_q(cc.handle(), q, h, j, indexStaging(q, h, j)));
        st[q].arrived++;
        if (++st[q].rowCount[h] == E) st[q].rowsStaged++;
    }
    void stageMinus(uint32_t q)
    {
        checkQuery(q);
        PieContext::check(piehip_stage_minus_q(cc.handle(), q, st[q].pinMinus));
        st[q].minusStaged = true;
    }

    void run()  // BatchedFHEHIPPIE.cpp:88-129 for every query of the batch
    {
        enqueue();
        collect();
    }
    void enqueue()
    {
        struct Reset {
            BatchedFHEHIPPIEQueryBatch &o;
            bool ok = false;
            ~Reset()
            {
                for (auto &s : o.st) {
                    s.minusStaged = false;
                    s.rowsStaged = s.arrived = 0;
                    std::fill(s.rowCount.begin(), s.rowCount.end(), 0u);
                }
                if (!ok) piehip_stage_reset(o.cc.handle());
            }
        } reset{*this};
        uint32_t complete = 0, touched = 0;
        for (const auto &s : st) {
            complete += (s.minusStaged && s.rowsStaged == K) ? 1u : 0u;
            touched += (s.minusStaged || s.rowsStaged || s.arrived) ? 1u : 0u;
        }
        if (complete == nq) {
            PieContext::check(piehip_run_staged(cc.handle(), pinRes));
        } else if (touched) {
            throw std::runtime_error("run: setMinusCompareElement and setIndex of every query of the batch must precede run()");
        } else {
            PieContext::check(piehip_run(cc.handle()));  // the previous batch again
            rerun = true;
        }
        reset.ok = true;
    }
    void collect()
    {
        if (rerun) PieContext::check(piehip_get_results(cc.handle(), pinRes));
        else PieContext::check(piehip_run_host_wait(cc.handle()));
        rerun = false;
        std::fill(listStale.begin(), listStale.end(), true);
    }
    // the b result ciphertexts of query q (materialised on the first call after a run(); resultTowers reads them in place)
    std::vector<LimbCt> &getResultList(uint32_t q)
    {
        checkQuery(q);
        if (listStale[q]) {
            const size_t ct = ctWords();
            for (uint32_t i = 0; i < b; i++) lists[q][i].limbs.assign(resultTowers(q, i), resultTowers(q, i) + ct);
            listStale[q] = false;
        }
        return lists[q];
    }
    const uint64_t *resultTowers(uint32_t q, uint32_t i) const { return pinRes + ((size_t)i * nq + q) * ctWords(); }  // rows [bin layer][query]

private:
    struct QueryState {
        uint64_t *pinIdx = nullptr, *pinMinus = nullptr;
        std::vector<uint32_t> rowCount;
        uint32_t rowsStaged = 0, arrived = 0;
        bool minusStaged = false;
    };
    void init()
    {
        if (nq < 1) throw std::invalid_argument("at least one query per run()");
        PieContext::check(piehip_set_query_batch(cc.handle(), nq));
        st.resize(nq);
        for (uint32_t q = 0; q < nq; q++) {
            PieContext::check(piehip_host_buffers_q(cc.handle(), q, &st[q].pinIdx, &st[q].pinMinus, &pinRes));
            st[q].rowCount.assign(K, 0u);
        }
        lists.assign(nq, std::vector<LimbCt>(b));
        listStale.assign(nq, false);
    }
    void checkQuery(uint32_t q) const
    {
        if (q >= nq) throw std::invalid_argument("query index outside the batch");
    }
    void drainUploads() { PieContext::check(piehip_run_host_wait(cc.handle())); }
    size_t ctWords() const { return 2 * (size_t)cc.towers() * cc.ringDimension(); }
    PieContext &cc;
    uint32_t K, b, E, nq;
    std::vector<QueryState> st;
    uint64_t *pinRes = nullptr;  // [b][nq][2][L][N], page-locked, owned by the library
    std::vector<std::vector<LimbCt>> lists;
    std::vector<bool> listStale;
    bool rerun = false;
};

}  // namespace piehip
