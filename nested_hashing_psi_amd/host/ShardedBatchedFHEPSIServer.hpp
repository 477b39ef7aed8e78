// ShardedBatchedFHEPSIServer.hpp -- the reference's server (src/Server/FHE/BatchedFHEPSIServer.{hpp,cpp}, phases of
// src/Server/PSIServer.hpp:66-87) as ONE PROCESS PER GPU, in C++ over the C ABI only: no Python, no torch.
//
// Rank 0 is the process that holds the client's channel (the reference server, .cpp:94-95); ranks 1 .. G-1 are workers on the
// other GPUs of the node.  Every rank keeps a contiguous slice of the bin layers of the same table (the outer
// loop of run(), BatchedFHEHIPPIE.cpp:91, has independent iterations) and a copy of the EvalMult key.
//
//   set-up   rank 0 receives context, public key, EvalMult key from the client (.cpp:21-54) and forwards context, key, the
//            table seeds and the RCCL unique id to the workers over the side channels (one connected socket per worker, the
//            framing of WireFraming.hpp -- host-side, once per session); every rank creates its context and joins the
//            communicator (piehip_rccl_init)
//   offline  every rank hashes the server set and packs ITS bin layers (piehip_build_db_bins, same seeds: slices of one table)
//   online   rank 0 receives the query and stages every piece as it lands (PCIe under the receive loop); then, inside the
//            reference's timer (.cpp:98-106): the query goes to every rank over xGMI (piehip_rccl_broadcast_query), every rank
//            runs its layers (piehip_run), the result ciphertexts are gathered to rank 0 (piehip_gather_results -- the
//            evaluation's only exchange) and come down to host memory; rank 0 answers the client (.cpp:143-152)
// Error handling is the reference's -- exceptions end the process (BatchedFHEHIPPIE.cpp:15,20 throw, nothing catches) -- made safe
// for a GROUP of processes: a collective only completes when every rank joins it, so
//   * at the end of the offline phase the ranks agree that everybody built its slice (piehip_rccl_agree): a rank whose build threw
//     says no on its way out and all of them end the session there, before anybody waits in a collective for it;
//   * every wait for a collective has a bound (piehip_rccl_wait, collectiveTimeoutMs): a rank that died in the online phase costs
//     its peers that time-out and an exception, not a hung group of server processes;
//   * a rank that leaves run() by exception aborts its side of the communicator first (piehip_rccl_abort), which ends its peers'
//     waits at once.
#pragma once
#include <chrono>
#include <exception>
#include <memory>
#include <random>

#include "BatchedFHEPSIServer.hpp"

namespace piehip {

class ShardedBatchedFHEPSIServer {
public:
    // rank 0: client_fd = the client's channel, side_fds = one connected socket per worker (index r - 1 for rank r);
    // rank r > 0: client_fd = -1, side_fds = {the socket to rank 0}
    ShardedBatchedFHEPSIServer(int rank, int nranks, int device, int client_fd, const std::vector<int> &side_fds,
                               const std::vector<uint64_t> &serverSet, const HashTableParameter &htParams, uint64_t hashSeed = 987654321)
        : rank(rank), G(nranks), device(device), fd(client_fd), side(side_fds), serverSet(serverSet), ht(htParams), hashSeed(hashSeed)
    {
        if (G < 1 || rank < 0 || rank >= G) throw std::invalid_argument("rank outside the server group");
        if ((rank == 0 && side.size() != (size_t)G - 1) || (rank > 0 && side.size() != 1))
            throw std::invalid_argument("side channels: one per worker on rank 0, one to rank 0 on a worker");
        if (ht.serverStashSize != 0) throw std::invalid_argument("Error, batched FHE PIE does not support a stash (yet).");
        if (rank == 0) {  // secret per session, as in the reference (CuckooHashTable.cpp:51-52, BatchedFHEHIPPIE.cpp:25-26)
            std::random_device rd;
            auto u64 = [&rd] { return ((uint64_t)rd() << 32) ^ (uint64_t)rd(); };
            seeds[0] = u64(), seeds[1] = u64(), seeds[2] = u64();
        }
    }
    void setSecretSeedsForTesting(uint64_t evict, uint64_t shuffle, uint64_t mask) { seeds[0] = evict, seeds[1] = shuffle, seeds[2] = mask; }

    void run()  // PSIServer.hpp:66-87
    {
        try {
            runSetUpPhase();
            if (rank == 0) wire::signalPhaseOver(fd);
            runOfflinePhase();
            if (rank == 0) wire::signalPhaseOver(fd);
            runOnlinePhase();
        } catch (...) {
            if (cc) (void)piehip_rccl_abort(cc->handle());   // the peers' waits end now, not at their time-out
            throw;
        }
    }

    bool failOfflineForTesting = false;     // this rank's database build "fails": the group must end the session, not hang
    uint32_t collectiveTimeoutMs = 30000;   // bound on every wait for the other ranks (piehip_rccl_wait / piehip_rccl_agree)

    long long offlineComputation = 0, onlineComputation = 0;  // microseconds (rank 0: PSIServer.hpp:89-103)

    void runSetUpPhase()
    {
        std::vector<uint8_t> m;
        ContextMessage c;
        std::vector<uint64_t> evk;
        uint8_t id[PIEHIP_RCCL_ID_BYTES];
        if (rank == 0) {
            wire::readWithSizeIntoVector(fd, m);
            if (m.size() != sizeof(ContextMessage)) throw std::runtime_error("context message size");
            std::memcpy(&c, m.data(), sizeof(c));
            if (c.L < 1 || c.L > 7) throw std::invalid_argument("context: L out of range");
            wire::readWithSizeIntoVector(fd, m);  // public key: unused by the operator
            wire::readWithSizeIntoVector(fd, m);  // EvalMult key [L][2][L][N]
            const size_t words = (size_t)c.L * 2 * c.L * c.N;
            if (m.size() != words * sizeof(uint64_t)) throw std::runtime_error("EvalMult key message size");
            evk.resize(words);
            std::memcpy(evk.data(), m.data(), m.size());
            wire::checkCanonical(evk.data(), (size_t)c.L * 2 * c.L, c.L, c.N, c.moduli, "EvalMult key");
            PieContext::check(piehip_rccl_unique_id(id));
            for (int s : side) {  // session set-up for the workers: id, context, seeds, key
                wire::writeWithSize(s, id, sizeof(id));
                wire::writeWithSize(s, &c, sizeof(c));
                wire::writeWithSize(s, seeds, sizeof(seeds));
                wire::writeWithSize(s, evk.data(), evk.size() * sizeof(uint64_t));
            }
        } else {
            const int s = side[0];
            wire::readWithSizeIntoVector(s, m);
            if (m.size() != sizeof(id)) throw std::runtime_error("unique id message size");
            std::memcpy(id, m.data(), sizeof(id));
            wire::readWithSizeIntoVector(s, m);
            if (m.size() != sizeof(ContextMessage)) throw std::runtime_error("context message size");
            std::memcpy(&c, m.data(), sizeof(c));
            if (c.L < 1 || c.L > 7) throw std::invalid_argument("context: L out of range");
            wire::readWithSizeIntoVector(s, m);
            if (m.size() != sizeof(seeds)) throw std::runtime_error("seed message size");
            std::memcpy(seeds, m.data(), sizeof(seeds));
            wire::readWithSizeIntoVector(s, m);
            const size_t words = (size_t)c.L * 2 * c.L * c.N;
            if (m.size() != words * sizeof(uint64_t)) throw std::runtime_error("EvalMult key message size");
            evk.resize(words);
            std::memcpy(evk.data(), m.data(), m.size());
        }
        cc.reset(new PieContext(c.N, c.L, c.t, c.moduli, c.moduli + c.L, device));
        qMod.assign(c.moduli, c.moduli + c.L);
        cc->setEvalMultKey(evk.data());
        PieContext::check(piehip_rccl_init(cc->handle(), id, G, rank));
        PieContext::check(piehip_rccl_bin_slice(ht.maxItemsPerPosition, G, rank, &lo, &hi));
        if (hi > lo)
            PieContext::check(piehip_reserve(cc->handle(), serverSet.size(), ht.numberOfSimpleHashFunctions, ht.eachSimpleTableSize,
                                             ht.numberOfCuckooHashFunctions, ht.maxItemsPerPosition, ht.eachCuckooTableSize, lo, hi));
    }

    void runOfflinePhase()
    {
        const auto begin = std::chrono::steady_clock::now();
        std::exception_ptr failed;
        try {
            if (hi <= lo) throw std::runtime_error("more ranks than bin layers: start at most eachBinSize server processes");
            if (failOfflineForTesting) throw std::runtime_error("offline phase failed on this rank (test)");
            PieContext::check(piehip_build_db_bins(cc->handle(), serverSet.data(), serverSet.size(), ht.numberOfSimpleHashFunctions,
                                                   ht.eachSimpleTableSize, ht.numberOfCuckooHashFunctions, ht.maxItemsPerPosition,
                                                   ht.eachCuckooTableSize, hashSeed, seeds[0], seeds[1], seeds[2], lo, hi));
            PieContext::check(piehip_sync(cc->handle()));
        } catch (...) {
            failed = std::current_exception();
        }
        // every rank says whether its slice is ready; a no anywhere ends the session everywhere (a Cuckoo insertion that failed
        // -- CuckooHashTable.cpp:113 -- fails on every rank alike; out of memory on one GPU does not)
        int allBuilt = 0;
        PieContext::check(piehip_rccl_agree(cc->handle(), failed ? 0 : 1, &allBuilt, collectiveTimeoutMs));
        if (failed) std::rethrow_exception(failed);
        if (!allBuilt) throw std::runtime_error("another rank of the server group could not build its slice of the database");
        // one evaluation of an all-zero query through the whole online path (code objects, queues, the communicator's first
        // collective) while nobody waits for it.  Only rank 0 stages queries from host memory: the workers get their device-side
        // input buffers and run queues, no page-locked index matrix (29 MiB at C3 that nothing would ever write)
        const size_t ct = ctWords();
        uint64_t *pinIdx = nullptr, *pinMinus = nullptr, *pinRes = nullptr;
        if (rank == 0)
            PieContext::check(piehip_host_buffers(cc->handle(), &pinIdx, &pinMinus, &pinRes));
        else
            PieContext::check(piehip_host_buffers(cc->handle(), nullptr, nullptr, nullptr));
        if (rank == 0) {
            const uint32_t K = ht.numberOfCuckooHashFunctions, E = ht.eachCuckooTableSize;
            std::memset(pinMinus, 0, ct * sizeof(uint64_t));
            std::memset(pinIdx, 0, (size_t)K * E * ct * sizeof(uint64_t));
            PieContext::check(piehip_stage_minus(cc->handle(), pinMinus));
            for (uint32_t h = 0; h < K; h++) PieContext::check(piehip_stage_index_row(cc->handle(), h, pinIdx + (size_t)h * E * ct));
        }
        evaluateStagedQuery();
        offlineComputation = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
    }

    void runOnlinePhase()
    {
        const uint32_t L = cc->towers(), N = cc->ringDimension(), K = ht.numberOfCuckooHashFunctions, E = ht.eachCuckooTableSize,
                       b = ht.maxItemsPerPosition;
        const size_t ct = ctWords();
        if (rank == 0) {
            uint64_t *pinIdx = nullptr, *pinMinus = nullptr, *pinRes = nullptr;
            PieContext::check(piehip_host_buffers(cc->handle(), &pinIdx, &pinMinus, &pinRes));
            std::vector<uint8_t> m;
            try {
                wire::readWithSizeIntoVector(fd, m);  // receiveEncryptedMinusElements, .cpp:114-122
                wire::unpackCiphertextsInto(m, L, N, pinMinus, 1, qMod.data());
                PieContext::check(piehip_stage_minus(cc->handle(), pinMinus));
                for (uint32_t h = 0; h < K; h++)  // receiveIndexMatrix, .cpp:124-141: every message staged as it lands
                    for (uint32_t j = 0; j < E; j++) {
                        wire::readWithSizeIntoVector(fd, m);
                        wire::unpackCiphertextsInto(m, L, N, pinIdx + ((size_t)h * E + j) * ct, 1, qMod.data());
                        PieContext::check(piehip_stage_index_ct_q(cc->handle(), 0, h, j, pinIdx + ((size_t)h * E + j) * ct));
                    }
            } catch (...) {
                piehip_stage_reset(cc->handle());
                throw;
            }
        }
        const auto begin = std::chrono::steady_clock::now();
        const uint64_t *results = evaluateStagedQuery();  // .cpp:101-103 across the ranks
        onlineComputation = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
        if (rank == 0)
            for (uint32_t i = 0; i < b; i++) {  // sendResult, .cpp:143-152
                const auto out = wire::packCiphertexts(results + (size_t)i * ct, 1, L, N);
                wire::writeWithSize(fd, out.data(), out.size());
            }
    }

private:
    size_t ctWords() const { return 2 * (size_t)cc->towers() * cc->ringDimension(); }
    // the query staged on rank 0 -> every rank; run(); results -> rank 0's host memory.  Returns the b result ciphertexts (rank 0).
    const uint64_t *evaluateStagedQuery()
    {
        const uint32_t b = ht.maxItemsPerPosition;
        PieContext::check(piehip_rccl_broadcast_query(cc->handle(), 0));
        PieContext::check(piehip_run(cc->handle()));
        uint64_t *gathered = nullptr;  // rank 0: page-locked, owned by the library, [b][2][L][N] in bin order
        PieContext::check(piehip_gather_results_host(cc->handle(), b, 0, &gathered));
        PieContext::check(piehip_rccl_wait(cc->handle(), collectiveTimeoutMs));   // piehip_sync with a bound
        return gathered;
    }

    int rank, G, device, fd;
    std::vector<int> side;
    std::vector<uint64_t> serverSet;
    HashTableParameter ht;
    uint64_t hashSeed;
    uint64_t seeds[3] = {0, 0, 0};  // evict, shuffle, mask: drawn on rank 0, the same on every rank (slices of ONE table)
    uint32_t lo = 0, hi = 0;
    std::vector<uint64_t> qMod;
    std::unique_ptr<PieContext> cc;
};

}  // namespace piehip
