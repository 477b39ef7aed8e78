// Compile-and-link check of the C++ host facade against libpiehip.so (run by tests/test_abi.py).
// With a GPU it also runs the reference's call order on a tiny table; without one it verifies that
// construction fails loudly (no CPU fallback).
#include <cstdio>
#include <vector>

#include "../nested_hashing_psi_amd/host/BatchedFHEHIPPIE.hpp"
#include "../nested_hashing_psi_amd/host/FHEHIPPIE.hpp"

int main()
{
    using namespace piehip;
    // argument checks of the reference constructor need no device
    HashTableView v;
    v.numberOfSimpleTables = 2, v.eachSimpleTableSize = 2, v.numberOfCuckooTables = 2, v.eachBinSize = 2, v.eachCuckooTableSize = 3;
    std::vector<uint64_t> tbl(2 * 2 * 2 * 2 * 3, 0);
    for (size_t i = 0; i < tbl.size(); i++) tbl[i] = (i * 7919u) % 65536u + 1;
    v.table = tbl.data();
    try {
        PieContext cc(1024, 2, 65537);
        bool threw = false;
        try {
            HashTableView bad = v;
            bad.serverStashSize = 1;
            BatchedFHEHIPPIE pie(cc, bad);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        if (!threw) return 2;
        BatchedFHEHIPPIE pie(cc, v);
        const size_t ct = 2 * 2 * 1024;
        std::vector<uint64_t> evk(2 * 2 * 2 * 1024, 1);
        cc.setEvalMultKey(evk.data());
        LimbCt minus;
        minus.limbs.assign(ct, 3);
        std::vector<std::vector<LimbCt>> idx(2, std::vector<LimbCt>(3));
        for (auto &row : idx)
            for (auto &c : row) c.limbs.assign(ct, 5);
        pie.setMinusCompareElement(minus);
        pie.setIndex(std::move(idx));
        pie.run();
        std::printf("facade ok: %zu result ciphertexts\n", pie.getResultList().size());
        {
            // the reference's setters overwrite (BatchedFHEHIPPIE.hpp:40-48): setIndex(A); setIndex(B); run() evaluates B, and a
            // run() that was refused (only half a query set) leaves the operator usable
            const std::vector<LimbCt> first = pie.getResultList();
            auto matrix = [&](uint64_t v) {
                std::vector<std::vector<LimbCt>> m(2, std::vector<LimbCt>(3));
                for (auto &row : m)
                    for (auto &c : row) c.limbs.assign(ct, v);
                return m;
            };
            LimbCt junk;
            junk.limbs.assign(ct, 11);
            pie.setMinusCompareElement(junk);
            pie.setMinusCompareElement(minus);   // replaces junk
            pie.setIndex(matrix(9));
            pie.setIndex(matrix(5));             // replaces the matrix of nines
            pie.run();
            for (size_t i = 0; i < first.size(); i++)
                if (pie.getResultList()[i].limbs != first[i].limbs) return 8;
            pie.setIndex(matrix(9));
            bool refused = false;
            try {
                pie.run();  // no minus element for this query
            } catch (const std::runtime_error &) {
                refused = true;
            }
            if (!refused) return 9;
            pie.setMinusCompareElement(minus);
            pie.setIndex(matrix(5));
            pie.run();
            for (size_t i = 0; i < first.size(); i++)
                if (pie.getResultList()[i].limbs != first[i].limbs) return 10;
            pie.run();  // nothing set since: the same query again
            for (size_t i = 0; i < first.size(); i++)
                if (pie.getResultList()[i].limbs != first[i].limbs) return 11;
            std::printf("setter overwrite / refused run ok\n");
        }
        {
            // a second query slot on the same database: the same query gives the same result list, evaluated concurrently
            PieContext cc2(1024, 2, 65537);
            BatchedFHEHIPPIE slot(cc2, pie);
            std::vector<std::vector<LimbCt>> idx2(2, std::vector<LimbCt>(3));
            for (auto &row : idx2)
                for (auto &c : row) c.limbs.assign(ct, 5);
            slot.setMinusCompareElement(minus);
            slot.setIndex(std::move(idx2));
            pie.enqueue();
            slot.enqueue();
            pie.collect();
            slot.collect();
            for (size_t i = 0; i < pie.getResultList().size(); i++)
                if (pie.getResultList()[i].limbs != slot.getResultList()[i].limbs) return 4;
            std::printf("query slot ok\n");
        }
        {
            // three queries per run() on the same database, every one from a client with its own EvalMult key: each result list
            // equals the single-query operator's under that key
            std::vector<std::vector<LimbCt>> want;
            std::vector<std::vector<uint64_t>> keys;
            std::vector<LimbCt> minusOf(3);
            auto matrixOf = [&](uint32_t q) {
                std::vector<std::vector<LimbCt>> m(2, std::vector<LimbCt>(3));
                for (uint32_t h = 0; h < 2; h++)
                    for (uint32_t j = 0; j < 3; j++) m[h][j].limbs.assign(ct, 5 + 7 * q + h + 2 * j);
                return m;
            };
            for (uint32_t q = 0; q < 3; q++) {   // what each client would get from an operator of its own
                keys.emplace_back(2 * 2 * 2 * 1024, 1 + 5 * q);
                cc.setEvalMultKey(keys[q].data());
                minusOf[q].limbs.assign(ct, 3 + q);
                pie.setMinusCompareElement(minusOf[q]);
                pie.setIndex(matrixOf(q));
                pie.run();
                want.push_back(pie.getResultList());
            }
            cc.setEvalMultKey(evk.data());
            PieContext cc3(1024, 2, 65537);
            BatchedFHEHIPPIEQueryBatch batch(cc3, pie, 3);
            for (uint32_t q = 0; q < 3; q++) {
                batch.setEvalMultKey(q, keys[q].data());
                batch.setMinusCompareElement(q, minusOf[q]);
                batch.setIndex(q, matrixOf(q));
            }
            batch.run();
            for (uint32_t q = 0; q < 3; q++)
                for (size_t i = 0; i < want[q].size(); i++)
                    if (batch.getResultList(q)[i].limbs != want[q][i].limbs) return 6;
            batch.run();  // the same batch again
            for (uint32_t q = 0; q < 3; q++)
                for (size_t i = 0; i < want[q].size(); i++)
                    if (batch.getResultList(q)[i].limbs != want[q][i].limbs) return 12;
            bool refused = false;
            try {
                batch.setMinusCompareElement(3, minus);
            } catch (const std::invalid_argument &) {
                refused = true;
            }
            if (!refused) return 7;
            refused = false;
            batch.setMinusCompareElement(1, minus);
            try {
                batch.run();  // one query of the batch half set
            } catch (const std::runtime_error &) {
                refused = true;
            }
            if (!refused) return 13;
            std::printf("query batch ok\n");
        }
        // the rotation-based sibling (FHEHIPPIE.hpp): argument checks, then the reference call order
        CuckooTableView cv;
        cv.numberOfHashFunctions = 2, cv.binSize = 4, cv.eachTableSize = 4;
        std::vector<uint64_t> ctab(2 * 4 * 4);
        for (size_t i = 0; i < ctab.size(); i++) ctab[i] = (i * 104729u) % 65536u + 1;
        cv.table = ctab.data();
        threw = false;
        try {
            CuckooTableView bad = cv;
            bad.binSize = 3;
            FHEHIPPIE p2(cc, {bad});
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        if (!threw) return 3;
        std::vector<int32_t> rots = {1, 2, -1, -2, -3};
        std::vector<uint64_t> rk(rots.size() * 2 * 2 * 2 * 1024, 1);
        setRotationKeys(cc, rots, rk.data());
        FHEHIPPIE rot(cc, {cv, cv});
        std::vector<LimbCt> ridx(2 * 2);
        for (auto &c : ridx) c.limbs.assign(ct, 7);
        rot.setIndex(std::move(ridx));
        rot.run();
        std::printf("rotation facade ok: %zu result ciphertexts\n", rot.getResultList().size());
        return 0;
    } catch (const std::runtime_error &e) {
        std::printf("no device: %s\n", e.what());
        return 77;  // skipped: no GPU
    }
}
