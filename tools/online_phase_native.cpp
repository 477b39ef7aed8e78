// online_phase_native.cpp -- tools/online_phase_probe.py without Python: the server's online phase over the C ABI in a process
// that holds only the system's HIP runtime (a Python process with torch holds the wheel's).
//   g++ -std=c++17 -O1 -Iinclude -o online_phase_native tools/online_phase_native.cpp -Lnested_hashing_psi_amd -lpiehip \
//       -Wl,-rpath,$PWD/nested_hashing_psi_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
//   ./online_phase_native [reps]        (rocprofv3 --kernel-trace --memory-copy-trace -- ./online_phase_native)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "piehip.h"

#define CK(x)                                                                   \
    do {                                                                        \
        if ((x) != 0) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, piehip_last_error());               \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t N = 16384, L = 4, K = 2, E = 14, b = 14, B = 9898;
    const uint64_t t = 4296540161ULL;
    uint64_t q[8], p[9];
    CK(piehip_default_moduli(N, L, q, p));
    piehip_handle h;
    CK(piehip_create(&h, N, L, t, q, p, 0, nullptr));
    uint64_t s = 88172645463325252ULL;
    auto rnd = [&s] { s ^= s << 13, s ^= s >> 7, s ^= s << 17; return s; };
    {
        std::vector<uint64_t> evk((size_t)L * 2 * L * N);
        for (size_t i = 0; i < evk.size(); i++) evk[i] = rnd() % q[(i / N) % L];
        CK(piehip_load_relin_key(h, evk.data()));
        std::vector<int64_t> slots((size_t)K * b * E * B), mask((size_t)b * B);
        for (auto &v : slots) v = (int64_t)(rnd() % 1000);
        for (auto &v : mask) v = 1 + (int64_t)(rnd() % 999);
        CK(piehip_load_db_slots(h, K, b, E, B, slots.data(), mask.data()));
    }
    uint64_t *idx, *minus, *res;
    CK(piehip_host_buffers(h, &idx, &minus, &res));
    const size_t ct = 2 * (size_t)L * N;
    for (size_t i = 0; i < (size_t)K * E * ct; i++) idx[i] = rnd() % q[(i / N) % L];
    for (size_t i = 0; i < ct; i++) minus[i] = rnd() % q[(i / N) % L];
    for (int rep = 0; rep < reps; rep++) {
        CK(piehip_stage_minus(h, minus));
        for (uint32_t r = 0; r < K; r++)
            for (uint32_t j = 0; j < E; j++) {
                CK(piehip_stage_index_ct_q(h, 0, r, j, idx + ((size_t)r * E + j) * ct));
                std::this_thread::sleep_for(std::chrono::microseconds(200));  // the next message arrives
            }
        const auto t0 = std::chrono::steady_clock::now();
        CK(piehip_run_staged(h, res));
        const auto t1 = std::chrono::steady_clock::now();
        CK(piehip_run_host_wait(h));
        const auto t2 = std::chrono::steady_clock::now();
        printf("run_staged returns after %lld us, results in host memory after %lld us\n",
               (long long)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count(),
               (long long)std::chrono::duration_cast<std::chrono::microseconds>(t2 - t0).count());
    }
    CK(piehip_destroy(h));
    return 0;
}
