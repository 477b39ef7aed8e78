// One rank of a native multi-rank run of the sharded path behind the C ABI (csrc/piehip_rccl.cpp), for tests/test_rccl_ranks.py:
//   rccl_ranks_main <rank> <nranks> <root> <N> <L> <t> <K> <E> <b> <nq> <dir> [mode]
// Every rank loads ITS contiguous slice of the bin layers (piehip_rccl_bin_slice) of the database in <dir>, joins the
// communicator (the unique id travels through a file), and then, twice: the root stages the nq queries from page-locked memory,
// piehip_rccl_broadcast_query on every rank, piehip_run, piehip_gather_results_host to the root, piehip_rccl_wait; the root
// writes the gathered list [b][nq][2][L][N] to <dir>/out<round>.bin.  The test compares it with the ORACLE's run() of every query.
// mode (error paths): "skip"  = the last rank leaves before the gather of round 1 without a word (exit 7): the root's wait must end
//                               by TIME-OUT, not hang;  "abort" = it calls piehip_rccl_abort first: the root's wait ends at once;
//                     "agree" = after round 0 every rank calls piehip_rccl_agree, the last rank says no: everybody exits 5.
// No torch, no Python: what a C++ server process does.  On the one-GPU test box the ranks share the GPU and the process preloads
// the test-only RCCL stand-in (tests/fake_rccl); with one rank per GPU and the real RCCL the same binary runs unchanged.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/piehip.h"

static std::vector<uint64_t> slurp(const std::string &path, size_t words)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<uint64_t> v(words);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(words * 8));
    if ((size_t)f.gcount() != words * 8) {
        std::fprintf(stderr, "%s: short file\n", path.c_str());
        std::exit(2);
    }
    return v;
}
#define CHECK(expr)                                                                                   \
    do {                                                                                              \
        int rc_ = (expr);                                                                             \
        if (rc_ != PIEHIP_OK) {                                                                       \
            std::fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #expr, rc_, piehip_last_error());   \
            return 3;                                                                                 \
        }                                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 12) return 2;
    const int rank = std::atoi(argv[1]), G = std::atoi(argv[2]), root = std::atoi(argv[3]);
    const uint32_t N = (uint32_t)std::atoi(argv[4]), L = (uint32_t)std::atoi(argv[5]);
    const uint64_t t = std::strtoull(argv[6], nullptr, 10);
    const uint32_t K = (uint32_t)std::atoi(argv[7]), E = (uint32_t)std::atoi(argv[8]), b = (uint32_t)std::atoi(argv[9]),
                   nq = (uint32_t)std::atoi(argv[10]);
    const std::string dir = argv[11], mode = argc > 12 ? argv[12] : "";
    const size_t LN = (size_t)L * N, ct = 2 * LN;
    piehip_handle h = nullptr;
    CHECK(piehip_create(&h, N, L, t, nullptr, nullptr, 0, nullptr));
    const std::vector<uint64_t> evk = slurp(dir + "/evk.bin", (size_t)L * 2 * LN);
    CHECK(piehip_load_relin_key(h, evk.data()));
    uint32_t lo = 0, hi = 0;
    CHECK(piehip_rccl_bin_slice(b, G, rank, &lo, &hi));
    {
        const std::vector<uint64_t> db = slurp(dir + "/db.bin", (size_t)K * b * E * LN), masks = slurp(dir + "/masks.bin", (size_t)b * LN);
        const uint32_t nb = hi - lo;
        std::vector<uint64_t> mine((size_t)K * nb * E * LN);
        for (uint32_t hf = 0; hf < K; hf++)
            std::memcpy(&mine[(size_t)hf * nb * E * LN], &db[((size_t)hf * b + lo) * E * LN], (size_t)nb * E * LN * 8);
        CHECK(piehip_load_db(h, K, nb, E, mine.data(), masks.data() + (size_t)lo * LN));
    }
    if (nq > 1) CHECK(piehip_set_query_batch(h, nq));
    // the unique id: made on the root, handed over through the file system (a server uses its side sockets)
    unsigned char id[PIEHIP_RCCL_ID_BYTES];
    const std::string idfile = dir + "/unique_id";
    if (rank == root) {
        CHECK(piehip_rccl_unique_id(id));
        std::ofstream f(idfile + ".tmp", std::ios::binary);
        f.write(reinterpret_cast<const char *>(id), sizeof(id));
        f.close();
        std::rename((idfile + ".tmp").c_str(), idfile.c_str());
    } else {
        for (int i = 0; i < 6000; i++) {
            std::ifstream f(idfile, std::ios::binary);
            if (f && f.read(reinterpret_cast<char *>(id), sizeof(id))) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (i == 5999) return 4;
        }
    }
    CHECK(piehip_rccl_init(h, id, G, rank));
    // only the root stages queries from host memory; the others need their device-side input buffers
    std::vector<uint64_t *> pinIdx(nq, nullptr), pinMinus(nq, nullptr);
    uint64_t *pinRes = nullptr;
    for (uint32_t q = 0; q < nq; q++) {
        if (rank == root)
            CHECK(piehip_host_buffers_q(h, q, &pinIdx[q], &pinMinus[q], &pinRes));
        else
            CHECK(piehip_host_buffers_q(h, q, nullptr, nullptr, nullptr));
    }
    std::vector<std::vector<uint64_t>> idx(nq), minus(nq);
    if (rank == root)
        for (uint32_t q = 0; q < nq; q++) {
            idx[q] = slurp(dir + "/idx" + std::to_string(q) + ".bin", (size_t)K * E * ct);
            minus[q] = slurp(dir + "/minus" + std::to_string(q) + ".bin", ct);
        }
    const uint32_t timeout_ms = 4000;
    for (int round = 0; round < 2; round++) {
        if (rank == root)
            for (uint32_t q = 0; q < nq; q++) {  // round 1: the queries change places
                const uint32_t src = (q + (uint32_t)round) % nq;
                std::memcpy(pinIdx[q], idx[src].data(), idx[src].size() * 8);
                std::memcpy(pinMinus[q], minus[src].data(), minus[src].size() * 8);
                CHECK(piehip_stage_minus_q(h, q, pinMinus[q]));
                for (uint32_t hf = 0; hf < K; hf++) CHECK(piehip_stage_index_row_q(h, q, hf, pinIdx[q] + (size_t)hf * E * ct));
            }
        CHECK(piehip_rccl_broadcast_query(h, root));
        CHECK(piehip_run(h));
        if (round == 1 && rank == G - 1 && rank != root && (mode == "skip" || mode == "abort")) {
            if (mode == "abort") (void)piehip_rccl_abort(h);
            // "skip": alive but silent for longer than the peers' bound (a process that exits closes its sockets, which the stand-in
            // transport notices at once; a rank stuck somewhere else is what the time-out is for)
            else std::this_thread::sleep_for(std::chrono::milliseconds(7000));
            return 7;   // leaves without joining the gather
        }
        uint64_t *gathered = nullptr;
        CHECK(piehip_gather_results_host(h, b, root, &gathered));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = piehip_rccl_wait(h, timeout_ms);
        if (rc != PIEHIP_OK) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("rank %d: wait ended after %.0f ms: %s\n", rank, ms, piehip_last_error());
            // the handle stays usable without its communicator: a collective call is refused, not hung
            uint64_t *again = nullptr;
            const int rc2 = piehip_gather_results_host(h, b, root, &again);
            std::printf("rank %d: gather after the abort -> %d\n", rank, rc2);
            piehip_destroy(h);
            return 6;
        }
        if (rank == root) {
            std::ofstream f(dir + "/out" + std::to_string(round) + ".bin", std::ios::binary);
            f.write(reinterpret_cast<const char *>(gathered), (std::streamsize)((size_t)b * nq * ct * 8));
        }
        if (round == 0 && mode == "agree") {
            int all = -1;
            CHECK(piehip_rccl_agree(h, rank == G - 1 ? 0 : 1, &all, timeout_ms));
            std::printf("rank %d: agree -> %d\n", rank, all);
            CHECK(piehip_rccl_destroy(h));
            piehip_destroy(h);
            return all ? 0 : 5;
        }
    }
    CHECK(piehip_rccl_destroy(h));
    CHECK(piehip_destroy(h));
    return 0;
}
