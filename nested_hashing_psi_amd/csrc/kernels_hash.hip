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
// order; the k*e blocked Cuckoo tables are then filled independently, one thread per table, each with its
// own eviction generator; the bin-layer shuffle runs one thread per (table, inner hash) row.
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
        hipLaunchKernelGGL(cuckoo_build_kernel, dim3((e + 63) / 64), dim3(64), 0, st, d_tab, d_items, vals_out, d_start, of, k, e, K, b,
                           E, evict_seed, d_tbl, d_fail);
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
