// kernels_hash.hip -- the server's offline phase on the device: nested hashing of the server set and
// gathering of the packed database.
//
// Replaces, with identical results for identical seeds (checked against oracle/pie_hashing.c):
//   TabulationHashing::hashWithIndicator            src/Common/Hashing/TabulationHashing.cpp:45-54
//   generateSimpleHashTable                         src/Common/Hashing/HashUtils.cpp:48-59
//   CuckooHashTable::insert / lookUp                src/Common/Hashing/CuckooHashTable.cpp:72-158
//   HierarchicalCuckooHashTable::insertAll          src/Common/Hashing/HierarchicalCuckooHashTable.cpp:55-72
//   BatchedFHEHIPPIE constructor: bin shuffle, gather, masks   BatchedFHEHIPPIE.cpp:23-82
//
// Structure: per outer hash function, items are keyed by their bucket and stably radix-sorted (rocPRIM via
// hipCUB: the only library primitive in this code base), which reproduces the reference's per-bucket item
// order; the k*e blocked Cuckoo tables are then filled independently, one wave per table with the table in LDS
// (one thread per table when a table does not fit), each with its own eviction generator; the bin-layer shuffle runs
// one thread per (table, inner hash) row.
#include <algorithm>
#include <hipcub/hipcub.hpp>

#include "kernels.hpp"

namespace piehip {

static const u32 HTPB = 256;

// xoshiro256** seeded by splitmix64: the generator of oracle/pie_oracle.c (po_rng)
struct Rng {
    u64 s[4];
};
__device__ __forceinline__ u64 rotl64(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ void rng_seed(Rng &r, u64 seed)
{
    for (int i = 0; i < 4; i++) {
        seed += 0x9E3779B97F4A7C15ULL;
        u64 z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        r.s[i] = z ^ (z >> 31);
    }
}
__device__ __forceinline__ u64 rng_next(Rng &r)
{
    const u64 result = rotl64(r.s[1] * 5, 7) * 9;
    const u64 t = r.s[1] << 17;
    r.s[2] ^= r.s[0];
    r.s[3] ^= r.s[1];
    r.s[1] ^= r.s[2];
    r.s[0] ^= r.s[3];
    r.s[2] ^= t;
    r.s[3] = rotl64(r.s[3], 45);
    return result;
}
__device__ __forceinline__ u64 rng_below(Rng &r, u64 bound)
{
    u64 mask = bound - 1;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    for (;;) {
        const u64 v = rng_next(r) & mask;
        if (v < bound) return v;
    }
}

// tab: [nfun][16][256] u64.  Items are 64-bit: bytes 8..15 of the reference's 128-bit input are zero.
__device__ __forceinline__ u64 tab_hash(const u64 *__restrict__ tab, u64 x, u32 hf)
{
    const u64 *t = tab + (size_t)hf * 16 * 256;
    u64 res = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        res ^= t[i * 256 + (x & 0xff)];
        x >>= 8;
    }
#pragma unroll
    for (int i = 8; i < 16; i++) res ^= t[i * 256];
    return res;
}

__global__ void __launch_bounds__(HTPB) hash_keys_kernel(const u64 *__restrict__ tab, const u64 *__restrict__ items, u32 n,
                                                         u32 hf, u32 e, u32 *__restrict__ keys, u32 *__restrict__ vals)
{
    const u32 a = blockIdx.x * HTPB + threadIdx.x;
    if (a >= n) return;
    keys[a] = (u32)(tab_hash(tab, items[a], hf) % e);
    vals[a] = a;
}

// start[p] = first position of bucket p in the sorted key array (start[e] = n)
__global__ void __launch_bounds__(HTPB) bucket_bounds_kernel(const u32 *__restrict__ keys, u32 n, u32 e, u32 *__restrict__ start)
{
    const u32 p = blockIdx.x * HTPB + threadIdx.x;
    if (p > e) return;
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (keys[mid] < p) lo = mid + 1; else hi = mid;
    }
    start[p] = lo;
}

__device__ __forceinline__ bool cuckoo_lookup(const u64 *__restrict__ tab, const u64 *T, u32 K, u32 b, u32 E, u32 first_hf, u64 x)
{
    for (u32 hf = 0; hf < K; hf++) {
        const u64 idx = tab_hash(tab, x, first_hf + hf) % E;
        for (u32 bin = 0; bin < b; bin++) {
            const u64 cur = T[((size_t)hf * b + bin) * E + idx];
            if (cur == x) return true;
            if (cur == 0) break;
        }
    }
    return false;
}

// one thread per inner table (outer function `of`, outer position p): sequential insertion in item order
__global__ void __launch_bounds__(64) cuckoo_build_kernel(const u64 *__restrict__ tab, const u64 *__restrict__ items,
                                                          const u32 *__restrict__ order, const u32 *__restrict__ start, u32 of,
                                                          u32 k, u32 e, u32 K, u32 b, u32 E, u64 evict_seed,
                                                          u64 *__restrict__ tbl, u32 *__restrict__ fail)
{
    const u32 p = blockIdx.x * 64 + threadIdx.x;
    if (p >= e) return;
    u64 *T = tbl + ((size_t)of * e + p) * K * b * E;
    Rng rng;
    rng_seed(rng, evict_seed * 0x100000001B3ULL + (u64)of * e + p);
    for (u32 a = start[p]; a < start[p + 1]; a++) {
        u64 x = items[order[a]];
        if (cuckoo_lookup(tab, T, K, b, E, k, x)) continue;
        bool placed = false;
        for (u32 run = 0; run < 1000 && !placed; run++) {  // numberOfRetries, CuckooHashTable.hpp:30
            for (u32 hf = 0; hf < K && !placed; hf++) {
                const u64 idx = tab_hash(tab, x, k + hf) % E;
                for (u32 bin = 0; bin < b; bin++) {
                    u64 *cell = &T[((size_t)hf * b + bin) * E + idx];
                    if (*cell == 0) {
                        *cell = x;
                        placed = true;
                        break;
                    }
                }
                if (!placed) {
                    const u32 ri = (u32)rng_below(rng, b);
                    u64 *cell = &T[((size_t)hf * b + ri) * E + idx];
                    const u64 tmp = *cell;
                    *cell = x;
                    x = tmp;
                }
            }
        }
        if (!placed) {
            atomicOr(fail, 1u);  // "(Blocked) Cuckoo hashing error", CuckooHashTable.cpp:113
            return;
        }
    }
}

// The same insertion walk with one WAVE per inner table and the table in LDS (K b E words: 3 KiB at the headline shape).
// The walk is sequential in the items (the reference's insertion order decides which item is evicted), but each step is a
// column scan: the lanes read the b cells of the item's column at once and a ballot finds the first empty one; the 16
// table look-ups of a tabulation hash are one load per lane and an XOR reduction.  One thread per table spent 2.1 ms per
// outer hash function at |S| = 2^20 (78 waves on 1024 SIMDs, every lane in its own table: no two loads coalesce).
// Same generator, same draws (one per eviction), same result as cuckoo_build_kernel.
__device__ __forceinline__ u64 wave_xor16(u64 v)  // XOR over each aligned group of 16 lanes, result in all of them
{
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}
__global__ void __launch_bounds__(256) cuckoo_build_wave_kernel(const u64 *__restrict__ tab, const u64 *__restrict__ items,
                                                                const u32 *__restrict__ order, const u32 *__restrict__ start,
                                                                u32 of, u32 k, u32 e, u32 K, u32 b, u32 E, u64 evict_seed,
                                                                u64 *__restrict__ tbl, u32 *__restrict__ fail, u32 waves_per_block)
{
    extern __shared__ u64 lds_tables[];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 p = blockIdx.x * waves_per_block + wave;
    if (wave >= waves_per_block || p >= e) return;  // no workgroup barrier below: waves are independent
    const u32 cells = K * b * E;
    // per wave: the table, then the columns of the current batch of 64 items under every inner function [K][64]
    u64 *T = lds_tables + (size_t)wave * (cells + (K * 64 + 1) / 2);
    u32 *S = reinterpret_cast<u32 *>(T + cells);
    for (u32 c = lane; c < cells; c += 64) T[c] = 0;
    Rng rng;  // wave-uniform: every lane carries the same state
    rng_seed(rng, evict_seed * 0x100000001B3ULL + (u64)of * e + p);
    // column of an evicted item (wave-uniform x, rare): lanes 0..15 fetch the 16 table entries of the hash
    auto column = [&](u64 x, u32 hf) -> u32 {
        const u32 i = lane & 15;
        const u64 part = tab[((size_t)(k + hf) * 16 + i) * 256 + (i < 8 ? (u32)((x >> (8 * i)) & 0xff) : 0u)];
        return (u32)(wave_xor16(part) % E);
    };
    // first empty bin of column (hf, idx), or b if it is full; hit: some bin of the column holds x
    auto scan = [&](u32 hf, u32 idx, u64 x, bool &hit) -> u32 {
        u32 first = b;
        hit = false;
        for (u32 b0 = 0; b0 < b; b0 += 64) {
            const u32 bin = b0 + lane;
            const u64 cur = bin < b ? T[((size_t)hf * b + bin) * E + idx] : ~(u64)0;
            const u64 zero = __ballot(cur == 0), same = __ballot(bin < b && cur == x);
            hit = hit || same != 0;
            if (zero != 0 && first == b) first = b0 + (u32)__ffsll((unsigned long long)zero) - 1;
        }
        return first;
    };
    const u32 a_end = start[p + 1];
    bool failed = false;
    for (u32 a0 = start[p]; a0 < a_end && !failed; a0 += 64) {
        // this batch: lane t holds item t and hashes it under every inner function (all its table look-ups in flight at once)
        const u64 mine = a0 + lane < a_end ? items[order[a0 + lane]] : 0;
        for (u32 hf = 0; hf < K; hf++) S[hf * 64 + lane] = (u32)(tab_hash(tab, mine, k + hf) % E);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32 cnt = a_end - a0 < 64 ? a_end - a0 : 64;
        for (u32 t = 0; t < cnt && !failed; t++) {
            u64 x = __shfl(mine, (int)t, 64);
            // lookUp: the item is already in one of its columns
            bool dup = false;
            for (u32 hf = 0; hf < K && !dup; hf++) {
                bool hit;
                (void)scan(hf, S[hf * 64 + t], x, hit);
                dup = hit;
            }
            if (dup) continue;
            bool placed = false, original = true;  // original: x is still item t (its columns are in S)
            for (u32 run = 0; run < 1000 && !placed; run++) {  // numberOfRetries, CuckooHashTable.hpp:30
                for (u32 hf = 0; hf < K && !placed; hf++) {
                    const u32 idx = original ? S[hf * 64 + t] : column(x, hf);
                    bool hit;
                    const u32 first = scan(hf, idx, x, hit);
                    if (first < b) {
                        if (lane == 0) T[((size_t)hf * b + first) * E + idx] = x;
                        placed = true;
                    } else {
                        const u32 ri = (u32)rng_below(rng, b);
                        u64 *cell = &T[((size_t)hf * b + ri) * E + idx];
                        const u64 tmp = *cell;  // every lane reads the same word
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) *cell = x;
                        x = tmp;
                        original = false;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (!placed) {
                if (lane == 0) atomicOr(fail, 1u);  // "(Blocked) Cuckoo hashing error", CuckooHashTable.cpp:113
                failed = true;
            }
        }
        __builtin_amdgcn_wave_barrier();  // S is rewritten by the next batch
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u64 *G = tbl + ((size_t)of * e + p) * cells;
    for (u32 c = lane; c < cells; c += 64) G[c] = T[c];
}

// Fisher-Yates over the b bin layers of one (table, inner hash) row, generator seeded per row
__global__ void __launch_bounds__(64) shuffle_rows_kernel(u64 *__restrict__ tbl, u32 rows, u32 b, u32 E, u64 seed)
{
    const u32 row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    Rng rng;
    rng_seed(rng, seed * 0x100000001B3ULL + row);
    u64 *base = tbl + (size_t)row * b * E;
    for (u32 i = b - 1; i > 0; i--) {
        const u32 j = (u32)rng_below(rng, i + 1);
        if (j != i)
            for (u32 c = 0; c < E; c++) {
                const u64 tmp = base[(size_t)i * E + c];
                base[(size_t)i * E + c] = base[(size_t)j * E + c];
                base[(size_t)j * E + c] = tmp;
            }
    }
}

// slots[h][bin][j][s] = tbl[s][h][bin][j] as a centred int64 (BatchedFHEHIPPIE.cpp:48-66); flags items >= t
__global__ void __launch_bounds__(HTPB) gather_slots_kernel(const u64 *__restrict__ tbl, u32 B, u32 K, u32 b, u32 E, u64 t,
                                                            int64_t *__restrict__ slots, u32 *__restrict__ fail)
{
    const u32 s = blockIdx.x * HTPB + threadIdx.x;
    if (s >= B) return;
    const u32 pt = blockIdx.y;  // (h * b + bin) * E + j
    const u64 v = tbl[(size_t)s * K * b * E + pt];
    if (v >= t) atomicOr(fail, 2u);
    slots[(size_t)pt * B + s] = v > t / 2 ? (int64_t)v - (int64_t)t : (int64_t)v;
}

__device__ __forceinline__ u64 mix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// counter-based masks in [1, t-1] (same rule as oracle ph_masks)
__global__ void __launch_bounds__(HTPB) mask_slots_kernel(u64 t, u32 B, u64 seed, int64_t *__restrict__ out)
{
    const u32 s = blockIdx.x * HTPB + threadIdx.x;
    if (s >= B) return;
    const u32 bin = blockIdx.y;
    const u64 x = mix64(mix64(seed) ^ (((u64)bin << 32) | s));
    const u64 v = __umul64hi(x, t - 1) + 1;
    out[(size_t)bin * B + s] = v > t / 2 ? (int64_t)v - (int64_t)t : (int64_t)v;
}

// ---- host-side driver -------------------------------------------------------------------------------------
size_t hash_sort_temp_bytes(u32 n, u32 e)
{
    size_t bytes = 0;
    int bits = 1;
    while ((1u << bits) < e) bits++;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                                             (u32 *)nullptr, (int)n, 0, bits, (hipStream_t) nullptr);
    return bytes;
}

// tbl [k][e][K][b][E] must be zeroed by the caller; scratch: keys/vals (2 x 2 x n u32), start (e+1 u32), temp
hipError_t launch_hash_build(const u64 *d_tab, const u64 *d_items, u32 n, u32 k, u32 e, u32 K, u32 b, u32 E, u64 evict_seed,
                             u64 shuffle_seed, u64 *d_tbl, u32 *d_keys, u32 *d_vals, u32 *d_start, void *d_temp, size_t temp_bytes,
                             u32 *d_fail, hipStream_t st)
{
    int bits = 1;
    while ((1u << bits) < e) bits++;
    u32 *keys_in = d_keys, *keys_out = d_keys + n, *vals_in = d_vals, *vals_out = d_vals + n;
    for (u32 of = 0; of < k; of++) {
        hipLaunchKernelGGL(hash_keys_kernel, dim3((n + HTPB - 1) / HTPB), dim3(HTPB), 0, st, d_tab, d_items, n, of, e, keys_in, vals_in);
        size_t tb = temp_bytes;
        hipError_t err = hipcub::DeviceRadixSort::SortPairs(d_temp, tb, (const u32 *)keys_in, keys_out, (const u32 *)vals_in, vals_out,
                                                            (int)n, 0, bits, st);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(bucket_bounds_kernel, dim3((e + 1 + HTPB - 1) / HTPB), dim3(HTPB), 0, st, keys_out, n, e, d_start);
        const size_t table_bytes = ((size_t)K * b * E + (K * 64 + 1) / 2) * sizeof(u64);  // + the batch's columns
        if (table_bytes <= 64 * 1024) {  // one wave per table, table in LDS
            const u32 wpb = (u32)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / table_bytes));
            hipLaunchKernelGGL(cuckoo_build_wave_kernel, dim3((e + wpb - 1) / wpb), dim3(64 * wpb), wpb * table_bytes, st, d_tab, d_items,
                               vals_out, d_start, of, k, e, K, b, E, evict_seed, d_tbl, d_fail, wpb);
        } else {
            hipLaunchKernelGGL(cuckoo_build_kernel, dim3((e + 63) / 64), dim3(64), 0, st, d_tab, d_items, vals_out, d_start, of, k, e, K, b,
                               E, evict_seed, d_tbl, d_fail);
        }
    }
    const u32 rows = k * e * K;
    hipLaunchKernelGGL(shuffle_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, d_tbl, rows, b, E, shuffle_seed);
    return hipGetLastError();
}

void launch_shuffle_rows(u64 *d_tbl, u32 rows, u32 b, u32 E, u64 seed, hipStream_t st)
{
    hipLaunchKernelGGL(shuffle_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, d_tbl, rows, b, E, seed);
}
void launch_gather_slots(const u64 *d_tbl, u32 B, u32 K, u32 b, u32 E, u64 t, int64_t *d_slots, u32 *d_fail, hipStream_t st)
{
    dim3 grid((B + HTPB - 1) / HTPB, K * b * E);
    hipLaunchKernelGGL(gather_slots_kernel, grid, dim3(HTPB), 0, st, d_tbl, B, K, b, E, t, d_slots, d_fail);
}
void launch_mask_slots(u64 t, u32 b, u32 B, u64 seed, int64_t *d_out, hipStream_t st)
{
    dim3 grid((B + HTPB - 1) / HTPB, b);
    hipLaunchKernelGGL(mask_slots_kernel, grid, dim3(HTPB), 0, st, t, B, seed, d_out);
}

}  // namespace piehip
