// ShardedBatchedFHEHIPPIE.hpp -- the reference operator's shape over several GPUs in ONE process.
//
// The reference server is one process holding one BatchedFHEHIPPIE (src/Server/FHE/BatchedFHEPSIServer.hpp:23,
// constructed at .cpp:86).  The outer loop of run() over bin layers (BatchedFHEHIPPIE.cpp:91) has independent iterations
// (SURVEY.md 8e), so this class gives that one process every GPU of the node: one PieContext (= one libpiehip handle) per
// device, each keeping a contiguous slice of the bin layers of the same shuffled table (piehip_load_db_table_bins with the
// same seeds), the per-query inputs replicated, run() enqueued on all devices before any is waited for, and the result list
// assembled in bin order -- each device copies its slice straight to host memory over its own PCIe link (the list leaves
// through sendResult, .cpp:143-152, so no peer copy to a "device 0" is needed).
// Host path: the query is copied once into the first shard's page-locked staging arrays; every piece (the minus element, each
// row of the index matrix) is then queued for upload on ALL shards at once (piehip_stage_*: asynchronous, one PCIe link per
// device), run() is piehip_run_staged on every shard followed by one wait per shard, and each shard's slice of the result
// list lands in that shard's page-locked result array.
// Same methods, call order and exceptions as BatchedFHEHIPPIE.hpp; one host thread drives all handles (the handles' own HIP
// streams run concurrently).  torchrun / one process per GPU with the RCCL gather is the other way to shard (bench.py).
#pragma once
#include <memory>

#include "BatchedFHEHIPPIE.hpp"

namespace piehip {

class ShardedBatchedFHEHIPPIE {
public:
    using Seeds = BatchedFHEHIPPIE::Seeds;

    // contexts: one per device (same ring, moduli and plaintext modulus; each must hold the EvalMult key)
    ShardedBatchedFHEHIPPIE(const std::vector<PieContext *> &contexts, const HashTableView &hct)
        : ShardedBatchedFHEHIPPIE(contexts, hct, Seeds::fromRandomDevice())
    {
    }
    // test-only: reproducible shuffle and masks
    ShardedBatchedFHEHIPPIE(const std::vector<PieContext *> &contexts, const HashTableView &hct, const Seeds &seeds) : ccs(contexts)
    {
        if (ccs.empty()) throw std::invalid_argument("at least one context");
        if (hct.serverStashSize != 0) throw std::invalid_argument("Error, batched FHE PIE does not support a stash (yet).");
        if (!hct.simpleMultiTables || !hct.cuckooMultiTables)
            throw std::invalid_argument("Error, batched FHE PIE currently does not support combined tables.");
        for (PieContext *c : ccs)
            if (c->ringDimension() != ccs[0]->ringDimension() || c->towers() != ccs[0]->towers() ||
                c->GetPlaintextModulus() != ccs[0]->GetPlaintextModulus())
                throw std::invalid_argument("contexts of a sharded operator must share their parameters");
        K = hct.numberOfCuckooTables;
        b = hct.eachBinSize;
        E = hct.eachCuckooTableSize;
        const uint32_t k = hct.numberOfSimpleTables, e = hct.eachSimpleTableSize;
        if ((size_t)k * e > ccs[0]->ringDimension()) throw std::invalid_argument("batch size exceeds the ring dimension");
        // devices beyond the bin count stay idle (b = 14 on 16 GPUs); slices differ by at most one layer
        const uint32_t G = (uint32_t)std::min<size_t>(ccs.size(), b);
        for (uint32_t g = 0; g < G; g++) {
            const uint32_t lo = (uint32_t)((uint64_t)b * g / G), hi = (uint32_t)((uint64_t)b * (g + 1) / G);
            PieContext::check(piehip_load_db_table_bins(ccs[g]->handle(), hct.table, k, e, K, b, E, seeds.shuffle, seeds.mask, lo, hi));
            slices.push_back({lo, hi});
        }
        resultList.resize(b);
        pinRes.resize(G);
        // the query is staged once, in the first shard's page-locked arrays (portable: every device uploads from them); the other
        // shards pin a result array only
        PieContext::check(piehip_host_buffers(ccs[0]->handle(), &pinIdx, &pinMinus, &pinRes[0]));
        for (uint32_t g = 1; g < G; g++) PieContext::check(piehip_host_buffers(ccs[g]->handle(), nullptr, nullptr, &pinRes[g]));
    }

    void run()  // BatchedFHEHIPPIE.cpp:88-129 on every shard
    {
        struct Reset {  // a refused or failed run() leaves no half-staged query behind on any shard
            ShardedBatchedFHEHIPPIE &o;
            bool ok = false;
            ~Reset()
            {
                o.minusStaged = false;
                o.rowsStaged = 0;
                if (!ok)
                    for (size_t g = 0; g < o.slices.size(); g++) piehip_stage_reset(o.ccs[g]->handle());
            }
        } reset{*this};
        if (minusStaged && rowsStaged == K) {
            for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_run_staged(ccs[g]->handle(), pinRes[g]));  // all devices busy
            for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_run_host_wait(ccs[g]->handle()));
        } else if (minusStaged || rowsStaged) {
            throw std::runtime_error("run: setMinusCompareElement and setIndex must both precede run()");
        } else {
            // nothing set since the last run(): the previous query again, as the reference operator (and BatchedFHEHIPPIE) would
            for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_run(ccs[g]->handle()));
            for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_get_results(ccs[g]->handle(), pinRes[g]));
        }
        reset.ok = true;
        listStale = true;
    }

    std::vector<LimbCt> &getResultList()  // .hpp:35-38; materialised from the shards' result arrays on the first call after run()
    {
        if (listStale) {
            const size_t ct = ctWords();
            for (size_t g = 0; g < slices.size(); g++)
                for (uint32_t i = slices[g].lo; i < slices[g].hi; i++) {
                    const uint64_t *src = pinRes[g] + (size_t)(i - slices[g].lo) * ct;
                    resultList[i].limbs.assign(src, src + ct);
                }
            listStale = false;
        }
        return resultList;
    }

    void setIndex(std::vector<std::vector<LimbCt>> &&indexMatrix)  // .hpp:40-43, [K][E] ciphertexts, replicated to every shard
    {
        const size_t ct = ctWords();
        if (indexMatrix.size() != K) throw std::invalid_argument("index matrix must have one row per inner hash function");
        for (uint32_t h = 0; h < K; h++) {
            if (indexMatrix[h].size() != E) throw std::invalid_argument("index matrix row length must be eachCuckooTableSize");
            for (uint32_t j = 0; j < E; j++)
                if (indexMatrix[h][j].limbs.size() != ct) throw std::invalid_argument("ciphertext does not match the context");
        }
        if (rowsStaged) drainUploads();  // a second matrix before run() replaces the first: its uploads read these very arrays
        for (uint32_t h = 0; h < K; h++) {
            uint64_t *row = pinIdx + (size_t)h * E * ct;
            for (uint32_t j = 0; j < E; j++) std::memcpy(row + (size_t)j * ct, indexMatrix[h][j].limbs.data(), ct * sizeof(uint64_t));
            // the row leaves for every device at once, while the next row is being copied
            for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_stage_index_row(ccs[g]->handle(), h, row));
        }
        rowsStaged = K;
    }

    void setMinusCompareElement(LimbCt minusCompareElement)  // .hpp:45-48
    {
        if (minusCompareElement.limbs.size() != ctWords()) throw std::invalid_argument("ciphertext does not match the context");
        if (minusStaged) drainUploads();
        std::memcpy(pinMinus, minusCompareElement.limbs.data(), ctWords() * sizeof(uint64_t));
        for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_stage_minus(ccs[g]->handle(), pinMinus));
        minusStaged = true;
    }

    struct Slice {
        uint32_t lo, hi;
    };
    const std::vector<Slice> &binSlices() const { return slices; }

private:
    size_t ctWords() const { return 2 * (size_t)ccs[0]->towers() * ccs[0]->ringDimension(); }
    void drainUploads()
    {
        for (size_t g = 0; g < slices.size(); g++) PieContext::check(piehip_run_host_wait(ccs[g]->handle()));
    }
    std::vector<PieContext *> ccs;
    std::vector<Slice> slices;
    uint32_t K = 0, b = 0, E = 0;
    std::vector<LimbCt> resultList;
    uint64_t *pinIdx = nullptr, *pinMinus = nullptr;  // the first shard's page-locked staging arrays: the source of every upload
    std::vector<uint64_t *> pinRes;                  // per shard: its slice of the result list, page-locked
    uint32_t rowsStaged = 0;
    bool minusStaged = false, listStale = false;
};

}  // namespace piehip
