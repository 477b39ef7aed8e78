/*
 * pie_hashing.c -- CPU restatement of the nested-hashing layer, the DB packing and the client
 * vector construction.  TEST INFRASTRUCTURE ONLY; reference citations in pie_hashing.h.
 */
#include "pie_hashing.h"

#include <stdlib.h>
#include <string.h>

#include "pie_oracle.h"

/* ---- std::mt19937 ------------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937;

static void mt_seed(mt19937 *g, uint64_t seed)
{
    g->mt[0] = (uint32_t)seed; /* seed mod 2^32, as mersenne_twister_engine::seed does */
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1) v ^= 0x9908b0dfu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* libstdc++ uniform_int_distribution<uint64_t>{} over a 32-bit engine: high word first */
static uint64_t mt_u64(mt19937 *g)
{
    uint64_t hi = mt_next(g);
    uint64_t lo = mt_next(g);
    return (hi << 32) + lo;
}

/* ---- TabulationHashing (TabulationHashing.cpp:16-54) --------------------------------------- */
struct ph_tab {
    uint32_t nfun;
    uint64_t *tbl; /* [nfun][16][256] */
};

ph_tab *ph_tab_create(uint64_t seed, uint32_t nfun)
{
    ph_tab *h = (ph_tab *)malloc(sizeof(ph_tab));
    h->nfun = nfun;
    h->tbl = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nfun * 16 * 256);
    mt19937 g;
    mt_seed(&g, seed);
    for (size_t i = 0; i < (size_t)nfun * 16 * 256; i++) h->tbl[i] = mt_u64(&g);
    return h;
}
void ph_tab_destroy(ph_tab *h)
{
    if (!h) return;
    free(h->tbl);
    free(h);
}
uint64_t ph_tab_hash(const ph_tab *h, uint64_t x, uint32_t hf)
{
    const uint64_t *t = h->tbl + (size_t)hf * 16 * 256;
    uint64_t res = 0;
    for (int i = 0; i < 16; i++) { /* tParam = 16 bytes; bytes 8..15 of a 64-bit item are zero */
        res ^= t[i * 256 + (x & 0xff)];
        x >>= 8;
    }
    return res;
}

/* ---- blocked Cuckoo table T[K][b][E] (CuckooHashTable.cpp:72-158) ------------------------ */
static int cuckoo_lookup(const ph_tab *h, const uint64_t *T, uint32_t K, uint32_t b, uint32_t E, uint32_t start,
                         uint64_t x)
{
    for (uint32_t hf = 0; hf < K; hf++) {
        uint64_t idx = ph_tab_hash(h, x, start + hf) % E;
        for (uint32_t bin = 0; bin < b; bin++) {
            uint64_t cur = T[((size_t)hf * b + bin) * E + idx];
            if (cur == x) return 1;
            if (cur == 0) break;
        }
    }
    return 0;
}
static int cuckoo_insert(const ph_tab *h, uint64_t *T, uint32_t K, uint32_t b, uint32_t E, uint32_t start, uint64_t x,
                         po_rng *rng)
{
    if (cuckoo_lookup(h, T, K, b, E, start, x)) return 0;
    for (uint32_t run = 0; run < 1000; run++) { /* numberOfRetries, CuckooHashTable.hpp:30 */
        for (uint32_t hf = 0; hf < K; hf++) {
            uint64_t idx = ph_tab_hash(h, x, start + hf) % E;
            for (uint32_t bin = 0; bin < b; bin++) {
                uint64_t *cell = &T[((size_t)hf * b + bin) * E + idx];
                if (*cell == 0) {
                    *cell = x;
                    return 0;
                }
            }
            uint32_t ri = (uint32_t)po_rng_below(rng, b);
            uint64_t *cell = &T[((size_t)hf * b + ri) * E + idx];
            uint64_t tmp = *cell;
            *cell = x;
            x = tmp;
        }
    }
    return -1; /* stash not supported on the batched path (BatchedFHEHIPPIE.cpp:13-16) */
}

int ph_hct_build(const ph_tab *h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint32_t K, uint32_t b,
                 uint32_t E, uint64_t evict_seed, uint64_t *tbl)
{
    const size_t sub = (size_t)K * b * E;
    memset(tbl, 0, sizeof(uint64_t) * (size_t)k * e * sub);
    uint32_t *pos = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    int rc = 0;
    for (uint32_t i = 0; i < k && rc == 0; i++) {
        /* generateSimpleHashTable (HashUtils.cpp:48-59): bucket = hash_i(x) mod e, item order kept */
        for (size_t a = 0; a < n; a++) pos[a] = (uint32_t)(ph_tab_hash(h, items[a], i) % e);
        /* inner tables use hash ids k..k+K-1 (HierarchicalCuckooHashTable.cpp:49) */
        po_rng *rngs = (po_rng *)malloc(sizeof(po_rng) * e);
        for (uint32_t p = 0; p < e; p++) po_rng_seed(&rngs[p], evict_seed * 0x100000001B3ULL + (uint64_t)i * e + p);
        for (size_t a = 0; a < n && rc == 0; a++) {
            uint64_t *T = tbl + ((size_t)i * e + pos[a]) * sub;
            if (cuckoo_insert(h, T, K, b, E, k, items[a], &rngs[pos[a]])) rc = -1;
        }
        free(rngs);
    }
    free(pos);
    return rc;
}

/* one independent generator per (sub-table, inner hash function) row -- rows can then be shuffled in parallel
 * (the GPU offline phase does); the reference draws from one std::random_device-seeded stream, so any
 * permutation is equally faithful (BatchedFHEHIPPIE.cpp:25-35) */
void ph_hct_shuffle_bins(uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, uint64_t seed)
{
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * E);
    for (size_t s = 0; s < (size_t)k * e; s++)
        for (uint32_t hf = 0; hf < K; hf++) {
            po_rng r;
            po_rng_seed(&r, seed * 0x100000001B3ULL + (s * K + hf));
            uint64_t *row = tbl + (s * K + hf) * (size_t)b * E; /* b layers of E */
            for (uint32_t i = b - 1; i > 0; i--) {
                uint32_t j = (uint32_t)po_rng_below(&r, i + 1);
                if (j != i) {
                    memcpy(tmp, row + (size_t)i * E, sizeof(uint64_t) * E);
                    memcpy(row + (size_t)i * E, row + (size_t)j * E, sizeof(uint64_t) * E);
                    memcpy(row + (size_t)j * E, tmp, sizeof(uint64_t) * E);
                }
            }
        }
    free(tmp);
}

void ph_pack_db(const uint64_t *tbl, uint32_t k, uint32_t e, uint32_t K, uint32_t b, uint32_t E, int64_t *slots)
{
    const size_t B = (size_t)k * e, sub = (size_t)K * b * E;
    for (uint32_t hf = 0; hf < K; hf++)
        for (uint32_t bin = 0; bin < b; bin++)
            for (uint32_t j = 0; j < E; j++) {
                int64_t *dst = slots + (((size_t)hf * b + bin) * E + j) * B;
                for (size_t s = 0; s < B; s++) dst[s] = (int64_t)tbl[s * sub + ((size_t)hf * b + bin) * E + j];
            }
}

/* counter-based: mask(bin, slot) = 1 + floor(mix(seed, bin, slot) * (t-1) / 2^64), mix = splitmix64 finaliser.
 * Uniform over [1, t-1] up to a 2^-31 bias (BatchedFHEHIPPIE.cpp:79 uses randGen(mt) % (t-1) + 1, biased too). */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void ph_masks(uint64_t t, uint32_t b, uint32_t B, uint64_t seed, int64_t *masks)
{
    for (uint32_t bin = 0; bin < b; bin++)
        for (uint32_t s = 0; s < B; s++) {
            uint64_t x = mix64(mix64(seed) ^ (((uint64_t)bin << 32) | s));
            uint64_t v = (uint64_t)(((unsigned __int128)x * (t - 1)) >> 64) + 1;
            masks[(size_t)bin * B + s] = (int64_t)v;
        }
}

int ph_client_build(const ph_tab *h, const uint64_t *items, size_t n, uint32_t k, uint32_t e, uint64_t evict_seed,
                    uint64_t *ctab)
{
    /* CuckooHashTable(hash, e, k, startingHashId 0, stash 0, multi, 1 layer): T[k][1][e] */
    memset(ctab, 0, sizeof(uint64_t) * (size_t)k * e);
    po_rng r;
    po_rng_seed(&r, evict_seed);
    for (size_t a = 0; a < n; a++)
        if (cuckoo_insert(h, ctab, k, 1, e, 0, items[a], &r)) return -1;
    return 0;
}

void ph_client_vectors(const ph_tab *h, const uint64_t *ctab, uint32_t k, uint32_t e, uint32_t K, uint32_t E,
                       int64_t *index, int64_t *minus)
{
    const size_t B = (size_t)k * e;
    memset(index, 0, sizeof(int64_t) * (size_t)K * E * B);
    for (size_t s = 0; s < B; s++) {
        uint64_t x = ctab[s];
        if (x == 0) {
            minus[s] = 1; /* dummy slot: every factor equals 1, never 0 (BatchedFHEPSIClient.cpp:128-131) */
        } else {
            minus[s] = -(int64_t)x;
            for (uint32_t hf = 0; hf < K; hf++) {
                uint64_t idx = ph_tab_hash(h, x, k + hf) % E;
                index[((size_t)hf * E + idx) * B + s] = 1;
            }
        }
    }
}

size_t ph_client_scan(const uint64_t *ctab, uint32_t k, uint32_t e, uint32_t b, const int64_t *decrypted, uint64_t *out)
{
    const size_t B = (size_t)k * e;
    size_t cnt = 0;
    for (size_t s = 0; s < B; s++)
        for (uint32_t bin = 0; bin < b; bin++)
            if (decrypted[(size_t)bin * B + s] == 0) {
                out[cnt++] = ctab[s];
                break;
            }
    return cnt;
}
