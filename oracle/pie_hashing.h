/*
 * pie_hashing.h -- CPU restatement of the nested-hashing layer and of the packing / client
 * vector construction either side of the hot path.  TEST INFRASTRUCTURE ONLY (see pie_oracle.h).
 *
 * Follows (reference, read as text):
 *   src/Common/Hashing/TabulationHashing.cpp:16-54      tabulation hash, 16 byte-tables x 256 x u64
 *   src/Common/Hashing/HashUtils.cpp:29-59              hash index = hash mod tableSize, simple table
 *   src/Common/Hashing/CuckooHashTable.cpp:72-158       blocked Cuckoo insert / lookUp
 *   src/Common/Hashing/HierarchicalCuckooHashTable.cpp:55-72  outer simple hashing, k tables
 *   src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.cpp:23-82  bin shuffle, DB gather, masks
 *   src/Client/FHE/BatchedFHEPSIClient.cpp:107-151,178-192   index matrix, minus vector, result scan
 * Items are uint64 (the reference's biginteger holds < 2^64 on the FHE path: cast to int64 at
 * BatchedFHEHIPPIE.cpp:62).  0 is the empty-slot sentinel on both sides.
 * The reference seeds evictions / shuffles / masks from std::random_device (not reproducible);
 * here every one of them takes an explicit seed.
 */
#ifndef PIE_HASHING_H
#define PIE_HASHING_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ph_tab ph_tab;
/* std::mt19937(seed) + std::uniform_int_distribution<uint64_t> as libstdc++ evaluates them */
ph_tab *ph_tab_create(uint64_t seed, uint32_t nfun);
void ph_tab_destroy(ph_tab *h);
uint64_t ph_tab_hash(const ph_tab *h, uint64_t x, uint32_t hf);

/* server: hierarchical table, flat tbl[k][e][K][b][E]; returns 0, or -1 on Cuckoo failure
 * (the reference throws runtime_error, CuckooHashTable.cpp:113) */
int ph_hct_build(const ph_tab *h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b,
                 uint32_t E, uint64_t evict_seed, uint64_t *tbl);
/* BatchedFHEHIPPIE.cpp:23-35: shuffle the b bin layers of every (sub-table, inner hash) row */
void ph_hct_shuffle_bins(uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, uint64_t seed);
/* BatchedFHEHIPPIE.cpp:45-70: slots[K][b][E][B], B = k*e, slot s = outer*e + pos */
void ph_pack_db(const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, int64_t *slots);
/* BatchedFHEHIPPIE.cpp:72-82: masks[b][B] uniform in [1, t-1] */
void ph_masks(uint64_t t, uint32_t b, uint32_t B, uint64_t seed, int64_t *masks);

/* client: Cuckoo table ctab[k][e] (one layer), BatchedFHEPSIClient.cpp:97-99,109 */
int ph_client_build(const ph_tab *h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint64_t evict_seed,
                    uint64_t *ctab);
/* BatchedFHEPSIClient.cpp:113-151: index[K][E][B] one-hot, minus[B] (dummy slot: +1, all-zero column) */
void ph_client_vectors(const ph_tab *h, const uint64_t *ctab, uint32_t k, uint32_t e, uint32_t K, uint32_t E,
                       int64_t *index, int64_t *minus);
/* BatchedFHEPSIClient.cpp:178-192: decrypted[b][B] -> intersection; returns count */
size_t ph_client_scan(const uint64_t *ctab, uint32_t k, uint32_t e, uint32_t b, const int64_t *decrypted,
                      uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
