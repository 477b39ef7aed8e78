// Several libpiehip handles alive in one process (run by tests/test_gpu_parity.py on the GPU box):
//  1. two contexts with different ring dimensions and different moduli evaluate interleaved queries -- per-device and
//     per-instantiation kernel attributes, tables and queues must not leak between handles;
//  2. ShardedBatchedFHEHIPPIE over three contexts (all on the one visible device, as three devices of a node would be)
//     returns, bit for bit, the result list of the unsharded BatchedFHEHIPPIE given the same seeds.
// Exit code 0 = ok, 77 = no GPU.
#include <chrono>
#include <cstdio>
#include <vector>

#include "../nested_hashing_psi_amd/host/ShardedBatchedFHEHIPPIE.hpp"

using namespace piehip;

static uint64_t mix(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static std::vector<uint64_t> moduli_of(PieContext &cc)
{
    std::vector<uint64_t> m(2 * cc.towers() + 2);
    PieContext::check(piehip_get_moduli(cc.handle(), m.data()));
    return m;
}

static void fill_ct(std::vector<uint64_t> &v, const std::vector<uint64_t> &mod, uint32_t L, uint32_t N, uint64_t &seed)
{
    v.resize(2 * (size_t)L * N);
    for (uint32_t c = 0; c < 2; c++)
        for (uint32_t i = 0; i < L; i++)
            for (uint32_t j = 0; j < N; j++) v[((size_t)c * L + i) * N + j] = mix(seed) % mod[i];
}

static long long g_last_query_us = 0;  // setMinusCompareElement + setIndex + run of the last query()
template <class Op>
static std::vector<std::vector<uint64_t>> query(Op &op, PieContext &cc, uint32_t K, uint32_t E, uint64_t seed)
{
    const auto mod = moduli_of(cc);
    LimbCt minus;
    fill_ct(minus.limbs, mod, cc.towers(), cc.ringDimension(), seed);
    std::vector<std::vector<LimbCt>> idx(K, std::vector<LimbCt>(E));
    for (auto &row : idx)
        for (auto &c : row) fill_ct(c.limbs, mod, cc.towers(), cc.ringDimension(), seed);
    const auto t0 = std::chrono::steady_clock::now();
    op.setMinusCompareElement(minus);
    op.setIndex(std::move(idx));
    op.run();
    g_last_query_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    std::vector<std::vector<uint64_t>> out;
    for (auto &c : op.getResultList()) out.push_back(c.limbs);
    return out;
}

static void load_key(PieContext &cc, uint64_t seed)
{
    const auto mod = moduli_of(cc);
    const uint32_t L = cc.towers(), N = cc.ringDimension();
    std::vector<uint64_t> evk((size_t)L * 2 * L * N);
    for (size_t i = 0; i < evk.size(); i++) evk[i] = mix(seed) % mod[(i / N) % L];
    cc.setEvalMultKey(evk.data());
}

int main()
{
    try {
        HashTableView v;
        v.numberOfSimpleTables = 2, v.eachSimpleTableSize = 5, v.numberOfCuckooTables = 2, v.eachBinSize = 7, v.eachCuckooTableSize = 3;
        std::vector<uint64_t> tbl((size_t)2 * 5 * 2 * 7 * 3);
        for (size_t i = 0; i < tbl.size(); i++) tbl[i] = (i * 7919u) % 65000u + 1;
        v.table = tbl.data();
        const BatchedFHEHIPPIE::Seeds seeds{11, 22};

        // 1. different rings / moduli side by side: A = 2^14 with the default 60-bit chain (folded register-blocked NTT, 68 KiB
        //    LDS per workgroup), B = 2^12 with caller-supplied 50-bit primes (generic paths)
        PieContext ccA(16384, 2, 65537);
        uint64_t q50[2], p50[3];
        {
            // the largest primes below 2^50 that are 1 mod 2N (trial division: ~16 M steps per prime)
            auto is_prime = [](uint64_t n) {
                if (n % 2 == 0) return false;
                for (uint64_t d = 3; d * d <= n; d += 2)
                    if (n % d == 0) return false;
                return true;
            };
            uint64_t c = (1ULL << 50) + 1;  // candidates stay 1 mod 8192
            int got = 0;
            uint64_t all[5];
            while (got < 5) {
                c -= 8192;
                if (is_prime(c)) all[got++] = c;
            }
            q50[0] = all[0], q50[1] = all[1], p50[0] = all[2], p50[1] = all[3], p50[2] = all[4];
        }
        PieContext ccB(4096, 2, 65537, q50, p50);
        load_key(ccA, 1);
        load_key(ccB, 2);
        BatchedFHEHIPPIE opA(ccA, v, seeds), opB(ccB, v, seeds);
        const auto a1 = query(opA, ccA, 2, 3, 100);
        const auto b1 = query(opB, ccB, 2, 3, 200);
        const auto a2 = query(opA, ccA, 2, 3, 100);   // the same query again, after the other handle ran
        const auto b2 = query(opB, ccB, 2, 3, 200);
        if (a1 != a2 || b1 != b2) {
            std::printf("interleaved handles disturbed each other\n");
            return 1;
        }
        // a fresh handle reproduces what the long-lived one computed
        {
            PieContext ccA2(16384, 2, 65537);
            load_key(ccA2, 1);
            BatchedFHEHIPPIE opA2(ccA2, v, seeds);
            if (query(opA2, ccA2, 2, 3, 100) != a1) {
                std::printf("second handle of the same shape differs\n");
                return 2;
            }
        }

        // 2. sharded operator == unsharded operator
        PieContext c0(16384, 2, 65537), c1(16384, 2, 65537), c2(16384, 2, 65537);
        load_key(c0, 1), load_key(c1, 1), load_key(c2, 1);
        ShardedBatchedFHEHIPPIE sh({&c0, &c1, &c2}, v, seeds);
        if (sh.binSlices().size() != 3 || sh.binSlices()[0].hi - sh.binSlices()[0].lo != 2 || sh.binSlices()[2].hi != 7) {
            std::printf("unexpected bin slices\n");
            return 3;
        }
        for (uint64_t s : {100ull, 300ull}) {
            const auto want = query(opA, ccA, 2, 3, s);
            const long long one_us = g_last_query_us;
            const auto got = query(sh, c0, 2, 3, s);
            // three handles: the uploads and runs of all shards are queued before any is waited for
            std::printf("query %llu: one handle %lld us, three shards (uploads and runs overlapped) %lld us\n", (unsigned long long)s, one_us,
                        g_last_query_us);
            if (got != want) {
                std::printf("sharded result differs from the unsharded one (query %llu)\n", (unsigned long long)s);
                return 4;
            }
        }
        std::printf("sharded check ok: 2 + 1 + 3 handles, %zu result ciphertexts\n", sh.getResultList().size());
        return 0;
    } catch (const std::runtime_error &e) {
        std::printf("no device: %s\n", e.what());
        return 77;
    }
}
