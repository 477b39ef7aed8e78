// FHEHIPPIE.hpp -- C++ host facade of the rotation-based sibling operator over the C ABI (include/piehip.h).
//
// Same class shape as the reference
//     src/Common/Crypto/PrivateIndexedEqualityCheck/FHEHIPPIE.hpp:18-49  (constructor, run(), getResultList(), setIndex(&&))
// and the same error behaviour (std::invalid_argument when bin size != bins per hash function or a stash is
// present, FHEHIPPIE.cpp:13-20).  The reference holds one operator per client slot in an FHEHIPPIECollection
// (PIECollection.hpp); here one object can hold a whole collection (`npie` tables) so the device evaluates them
// as one batch.  Rotation keys (EvalSum + EvalAtIndex maps of the crypto context) go to PieContext::setRotationKeys.
#pragma once
#include <algorithm>
#include <numeric>

#include "BatchedFHEHIPPIE.hpp"

namespace piehip {

// What the reference reads from its CuckooHashTable (FHEHIPPIE.cpp:9-59): sizes, the stash check and the table
struct CuckooTableView {
    uint32_t numberOfHashFunctions = 0;  // K
    uint32_t binSize = 0;                // b
    uint32_t eachTableSize = 0;          // E
    uint64_t stashSize = 0;
    const uint64_t *table = nullptr;     // [K][b][E]
};

inline void setRotationKeys(PieContext &cc, const std::vector<int32_t> &indices, const uint64_t *keys)
{
    PieContext::check(piehip_load_rotation_keys(cc.handle(), (uint32_t)indices.size(), indices.data(), keys));
}

class FHEHIPPIE {
public:
    // FHEHIPPIE(cryptor, pK, ct), FHEHIPPIE.cpp:9-59; `tables` = one view per operator of the collection.
    // permVec2 / permutationVector / masks come from a seeded generator instead of std::random_device.
    FHEHIPPIE(PieContext &cryptor, const std::vector<CuckooTableView> &tables, uint64_t seed = 0x9E3779B97F4A7C15ULL) : cc(cryptor)
    {
        if (tables.empty()) throw std::invalid_argument("empty collection");
        K = tables[0].numberOfHashFunctions;
        b = tables[0].binSize;
        E = tables[0].eachTableSize;
        for (const auto &ct : tables) {
            if (ct.binSize != ct.eachTableSize)
                throw std::invalid_argument(
                    "Error, for FHE PIE the size of a cuckoo bin has to be equal than the number of bins per hash function.");
            if (ct.stashSize != 0) throw std::invalid_argument("Error, FHE PIE does not support a stash (yet).");
            if (ct.numberOfHashFunctions != K || ct.binSize != b) throw std::invalid_argument("tables of one collection must share their shape");
        }
        npie = (uint32_t)tables.size();
        const uint64_t t = cc.GetPlaintextModulus();
        std::vector<int64_t> slots((size_t)npie * K * b * (E + 1), 1), masks((size_t)npie * K * b);
        permutationVector.resize(npie);
        uint64_t s = seed;
        for (uint32_t i = 0; i < npie; i++) {
            std::vector<uint32_t> permVec2 = permutation(b, s);  // hides the correct bin index, FHEHIPPIE.cpp:28
            permutationVector[i] = permutation(K, s);            // initPermutationVector, FHEHIPPIE.hpp:30-34
            for (uint32_t hf = 0; hf < K; hf++)
                for (uint32_t bin = 0; bin < b; bin++) {
                    int64_t *row = &slots[(((size_t)i * K + hf) * b + permVec2[bin]) * (E + 1)];
                    for (uint32_t pos = 0; pos < E; pos++) row[pos] = (int64_t)tables[i].table[((size_t)hf * b + bin) * E + pos];
                    row[E] = 1;  // exponent of the "minus client" element, FHEHIPPIE.cpp:48
                    masks[((size_t)i * K + hf) * b + bin] = (int64_t)(next(s) % (t - 1) + 1);  // without 0, FHEHIPPIE.cpp:52
                }
        }
        PieContext::check(piehip_fhepie_load_table(cc.handle(), npie, K, b, E, slots.data(), masks.data()));
        shuffledResultList.resize((size_t)npie * K);
    }

    void run()  // FHEHIPPIE.cpp:61-77
    {
        PieContext::check(piehip_fhepie_run(cc.handle()));
        const size_t ct = 2 * (size_t)cc.towers() * cc.ringDimension();
        std::vector<uint64_t> flat(ct * npie * K);
        PieContext::check(piehip_fhepie_get_results(cc.handle(), flat.data()));
        for (uint32_t i = 0; i < npie; i++)
            for (uint32_t hf = 0; hf < K; hf++) {
                const uint64_t *src = &flat[((size_t)i * K + hf) * ct];
                shuffledResultList[(size_t)i * K + permutationVector[i][hf]].limbs.assign(src, src + ct);  // FHEHIPPIE.cpp:76
            }
    }

    std::vector<LimbCt> &getResultList() { return shuffledResultList; }  // [npie][K], FHEHIPPIE.hpp:40-43

    void setIndex(std::vector<LimbCt> &&indexMatrix)  // [npie][K] ciphertexts, FHEHIPPIE.hpp:45-48
    {
        const size_t ct = 2 * (size_t)cc.towers() * cc.ringDimension();
        if (indexMatrix.size() != (size_t)npie * K) throw std::invalid_argument("index matrix must hold one ciphertext per hash function");
        std::vector<uint64_t> flat(indexMatrix.size() * ct);
        for (size_t i = 0; i < indexMatrix.size(); i++) {
            if (indexMatrix[i].limbs.size() != ct) throw std::invalid_argument("ciphertext does not match the context");
            std::memcpy(&flat[i * ct], indexMatrix[i].limbs.data(), ct * sizeof(uint64_t));
        }
        PieContext::check(piehip_fhepie_set_index(cc.handle(), flat.data()));
    }

protected:
    PieContext &cc;
    uint32_t npie = 0, K = 0, b = 0, E = 0;
    std::vector<LimbCt> shuffledResultList;
    std::vector<std::vector<uint32_t>> permutationVector;

    static uint64_t next(uint64_t &s)  // splitmix64
    {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    static std::vector<uint32_t> permutation(uint32_t n, uint64_t &s)  // createPermutationVector (Fisher-Yates)
    {
        std::vector<uint32_t> p(n);
        std::iota(p.begin(), p.end(), 0u);
        for (uint32_t i = n; i > 1; i--) std::swap(p[i - 1], p[next(s) % i]);
        return p;
    }
};

}  // namespace piehip
