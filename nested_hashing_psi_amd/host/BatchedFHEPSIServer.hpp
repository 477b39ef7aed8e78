// BatchedFHEPSIServer.hpp -- the caller of the hot path, in the reference's shape, over the C ABI and WireFraming.hpp.
//
// Mirrors src/Server/FHE/BatchedFHEPSIServer.{hpp,cpp} and the phase driver src/Server/PSIServer.hpp:66-87:
//     run():  runSetUpPhase(); signalPhaseOver(); runOfflinePhase(); signalPhaseOver(); runOnlinePhase();
//     setup   (.cpp:56-74)   three messages: context, public key (unused by the operator), EvalMult key
//     offline (.cpp:75-90)   insertAll(serverSet) + BatchedFHEHIPPIE construction  -> piehip_build_db (all on the device)
//     online  (.cpp:92-112)  minus ciphertext, K*E index ciphertexts (one message each, .cpp:124-141); timer around
//                            setMinusCompareElement / setIndex / run (.cpp:98-106); b results, one message each (.cpp:143-152)
// Payloads are the flat limb format of WireFraming.hpp instead of OpenFHE's cereal blobs (OpenFHE is not available here);
// with OpenFHE the same class deserialises the blobs and copies DCRTPoly towers.
//
// Several clients at once.  The reference server serves ONE client per process (.cpp:94-95: one channel).  Given a list of
// channels this class runs the same three phases with every client -- each sends its own context (they must agree), its own
// EvalMult key, its own query -- builds the database once, and evaluates the clients' queries as ONE batch
// (piehip_set_query_batch): stage A streams the packed database once for all of them.  Every client's answer is bit-identical to
// what a server of its own would have sent.
#pragma once
#include <chrono>
#include <memory>
#include <random>

#include "BatchedFHEHIPPIE.hpp"
#include "WireFraming.hpp"

namespace piehip {

struct HashTableParameter {  // src/Common/Parameter/HashTableParameter.hpp
    uint32_t numberOfSimpleHashFunctions, eachSimpleTableSize, numberOfCuckooHashFunctions, eachCuckooTableSize, maxItemsPerPosition;
    uint64_t serverStashSize = 0;
};

// first setup message: what the reference's serialized CryptoContext carries for this path
struct ContextMessage {
    uint32_t N, L;
    uint64_t t;
    uint64_t moduli[2 * 7 + 1];  // q_0..q_{L-1}, p_0..p_L
};

class BatchedFHEPSIServer {
public:
    // Cuckoo evictions, the bin-layer shuffle and the random masks are seeded from std::random_device, as in the reference
    // (CuckooHashTable.cpp:51-52, BatchedFHEHIPPIE.cpp:25-26): the masks hide non-matching slots from the client and the
    // shuffle hides the bin order.  setSecretSeedsForTesting() makes a run reproducible (parity tests only).
    BatchedFHEPSIServer(int channel_fd, const std::vector<uint64_t> &serverSet, const HashTableParameter &htParams, uint64_t hashSeed = 987654321)
        : BatchedFHEPSIServer(std::vector<int>{channel_fd}, serverSet, htParams, hashSeed)
    {
    }
    // one channel per client; their queries are evaluated together (at most 8: the library's batch limit)
    BatchedFHEPSIServer(const std::vector<int> &channel_fds, const std::vector<uint64_t> &serverSet, const HashTableParameter &htParams,
                        uint64_t hashSeed = 987654321)
        : fds(channel_fds), serverSet(serverSet), ht(htParams), hashSeed(hashSeed)
    {
        if (fds.empty() || fds.size() > 8) throw std::invalid_argument("between one and eight client channels");
        std::random_device rd;
        auto u64 = [&rd] { return ((uint64_t)rd() << 32) ^ (uint64_t)rd(); };
        evictSeed = u64();
        shuffleSeed = u64();
        maskSeed = u64();
        if (ht.serverStashSize != 0) throw std::invalid_argument("Error, batched FHE PIE does not support a stash (yet).");
    }

    void run()  // PSIServer.hpp:66-87
    {
        runSetUpPhase();
        for (int fd : fds) wire::signalPhaseOver(fd);
        runOfflinePhase();
        for (int fd : fds) wire::signalPhaseOver(fd);
        runOnlinePhase();
    }

    void setSecretSeedsForTesting(uint64_t evict, uint64_t shuffle, uint64_t mask)
    {
        evictSeed = evict;
        shuffleSeed = shuffle;
        maskSeed = mask;
    }

    long long offlineComputation = 0, onlineComputation = 0;  // microseconds, PSIServer.hpp:89-103

    void runSetUpPhase()  // receiveAndSetContextAndKeys, BatchedFHEPSIServer.cpp:21-54, once per client
    {
        const uint32_t nq = (uint32_t)fds.size();
        std::vector<uint8_t> m;
        ContextMessage c0;
        for (uint32_t q = 0; q < nq; q++) {
            const int fd = fds[q];
            wire::readWithSizeIntoVector(fd, m);
            if (m.size() != sizeof(ContextMessage)) throw std::runtime_error("context message size");
            ContextMessage c;
            std::memcpy(&c, m.data(), sizeof(c));
            if (c.L < 1 || c.L > 7) throw std::invalid_argument("context: L out of range");
            if (q == 0) {
                c0 = c;
                cc.reset(new PieContext(c.N, c.L, c.t, c.moduli, c.moduli + c.L));
                qMod.assign(c.moduli, c.moduli + c.L);
                // the table sizes are known since construction (htParams): allocate the database, workspace and scratch now, so
                // the timed offline phase does not pay for hipMalloc
                PieContext::check(piehip_set_query_batch(cc->handle(), nq));
                PieContext::check(piehip_reserve(cc->handle(), serverSet.size(), ht.numberOfSimpleHashFunctions, ht.eachSimpleTableSize,
                                                 ht.numberOfCuckooHashFunctions, ht.maxItemsPerPosition, ht.eachCuckooTableSize, 0,
                                                 ht.maxItemsPerPosition));
            } else if (c.N != c0.N || c.L != c0.L || c.t != c0.t || std::memcmp(c.moduli, c0.moduli, sizeof(uint64_t) * (2 * c.L + 1))) {
                throw std::invalid_argument("the clients of one batch must use the same crypto context parameters");
            }
            wire::readWithSizeIntoVector(fd, m);  // public key: stored by the reference, never used by the operator
            wire::readWithSizeIntoVector(fd, m);  // EvalMult key [L][2][L][N]
            const size_t words = (size_t)c.L * 2 * c.L * c.N;
            if (m.size() != words * sizeof(uint64_t)) throw std::runtime_error("EvalMult key message size");
            std::vector<uint64_t> evk(words);
            std::memcpy(evk.data(), m.data(), m.size());
            // [L][2][L][N], the modulus index is the innermost L: the key-switch accumulators take canonical residues only
            wire::checkCanonical(evk.data(), (size_t)c.L * 2 * c.L, c.L, c.N, qMod.data(), "EvalMult key");
            if (nq == 1) cc->setEvalMultKey(evk.data());
            else PieContext::check(piehip_load_relin_key_q(cc->handle(), q, evk.data()));  // every client's own key
        }
    }

    void runOfflinePhase()  // BatchedFHEPSIServer.cpp:75-90
    {
        const auto begin = std::chrono::steady_clock::now();
        // nested hashing (HierarchicalCuckooHashTable::insertAll) + the operator's constructor, on the device
        PieContext::check(piehip_build_db(cc->handle(), serverSet.data(), serverSet.size(), ht.numberOfSimpleHashFunctions,
                                          ht.eachSimpleTableSize, ht.numberOfCuckooHashFunctions, ht.maxItemsPerPosition,
                                          ht.eachCuckooTableSize, hashSeed, evictSeed, shuffleSeed, maskSeed));
        PieContext::check(piehip_sync(cc->handle()));
        // One evaluation of an all-zero batch while nobody waits for it: the first launch of every kernel loads its code object
        // and the first run creates the queues -- milliseconds that would otherwise land in the first client's online phase.
        {
            const uint32_t L = cc->towers(), N = cc->ringDimension(), K = ht.numberOfCuckooHashFunctions, E = ht.eachCuckooTableSize;
            const size_t ct = 2 * (size_t)L * N;
            uint64_t *pinRes = nullptr;
            for (uint32_t q = 0; q < fds.size(); q++) {
                uint64_t *pinIdx = nullptr, *pinMinus = nullptr;
                PieContext::check(piehip_host_buffers_q(cc->handle(), q, &pinIdx, &pinMinus, &pinRes));
                std::memset(pinMinus, 0, ct * sizeof(uint64_t));
                std::memset(pinIdx, 0, (size_t)K * E * ct * sizeof(uint64_t));
                PieContext::check(piehip_stage_minus_q(cc->handle(), q, pinMinus));
                for (uint32_t h = 0; h < K; h++) PieContext::check(piehip_stage_index_row_q(cc->handle(), q, h, pinIdx + (size_t)h * E * ct));
            }
            PieContext::check(piehip_run_staged(cc->handle(), pinRes));
            PieContext::check(piehip_run_host_wait(cc->handle()));
        }
        offlineComputation = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
    }

    void runOnlinePhase()  // BatchedFHEPSIServer.cpp:92-112
    {
        const uint32_t L = cc->towers(), N = cc->ringDimension(), K = ht.numberOfCuckooHashFunctions, E = ht.eachCuckooTableSize,
                       b = ht.maxItemsPerPosition, nq = (uint32_t)fds.size();
        const size_t ct = 2 * (size_t)L * N;
        // Every message is unpacked (and range-checked) straight into the library's page-locked staging arrays, and a piece's
        // upload starts as soon as it has landed: every message's ciphertext at once -- the 29 MiB of a C3 query cross PCIe
        // underneath the receive loop, and when the timer starts only the last megabyte is still on its way (the reference deserialises into
        // Ciphertext objects in the same place, .cpp:114-141, before its timer starts at .cpp:98).  A failed receive drops the
        // partial staging (piehip_stage_reset) before the exception leaves.
        uint64_t *pinRes = nullptr;
        std::vector<uint8_t> m;
        try {
            for (uint32_t q = 0; q < nq; q++) {
                const int fd = fds[q];
                uint64_t *pinIdx = nullptr, *pinMinus = nullptr;
                PieContext::check(piehip_host_buffers_q(cc->handle(), q, &pinIdx, &pinMinus, &pinRes));
                wire::readWithSizeIntoVector(fd, m);  // receiveEncryptedMinusElements, .cpp:114-122
                wire::unpackCiphertextsInto(m, L, N, pinMinus, 1, qMod.data());
                PieContext::check(piehip_stage_minus_q(cc->handle(), q, pinMinus));
                for (uint32_t h = 0; h < K; h++)  // receiveIndexMatrix, .cpp:124-141: one message per ciphertext, staged as it lands
                    for (uint32_t j = 0; j < E; j++) {
                        wire::readWithSizeIntoVector(fd, m);
                        wire::unpackCiphertextsInto(m, L, N, pinIdx + ((size_t)h * E + j) * ct, 1, qMod.data());
                        PieContext::check(piehip_stage_index_ct_q(cc->handle(), q, h, j, pinIdx + ((size_t)h * E + j) * ct));
                    }
            }
        } catch (...) {
            piehip_stage_reset(cc->handle());
            throw;
        }
        const auto begin = std::chrono::steady_clock::now();
        // setMinusCompareElement / setIndex / run (.cpp:101-103) on the staged queries; the result lists are in host memory at the end
        PieContext::check(piehip_run_staged(cc->handle(), pinRes));
        PieContext::check(piehip_run_host_wait(cc->handle()));
        onlineComputation = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
        for (uint32_t q = 0; q < nq; q++)
            for (uint32_t i = 0; i < b; i++) {  // sendResult, .cpp:143-152; rows of the result array are [bin layer][query]
                const auto out = wire::packCiphertexts(pinRes + ((size_t)i * nq + q) * ct, 1, L, N);
                wire::writeWithSize(fds[q], out.data(), out.size());
            }
    }

private:
    std::vector<int> fds;   // one channel per client of the batch
    std::vector<uint64_t> serverSet;
    HashTableParameter ht;
    uint64_t hashSeed;
    uint64_t evictSeed = 0, shuffleSeed = 0, maskSeed = 0;
    std::vector<uint64_t> qMod;
    std::unique_ptr<PieContext> cc;
};

}  // namespace piehip
