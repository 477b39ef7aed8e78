// One rank of the multi-process C++ server (host/ShardedBatchedFHEPSIServer.hpp): one process per GPU, RCCL behind the C ABI.
//   sharded_server_main <rank> <nranks> <device> <client fd | -1> <side fd[,side fd ...] | -> <server set file> k e K E b
// Rank 0 holds the client's socket and one side socket per worker; a worker holds one side socket to rank 0.
// tests/test_sharding_gpu.py::test_cpp_server_over_rccl_one_rank runs it with one rank over the real RCCL (the test box has one GPU
// and RCCL refuses two ranks on one device); tests/test_rccl_ranks.py runs 2, 4 and 5 ranks of it on that one GPU over the test-only
// stand-in of tests/fake_rccl; tests/test_abi.py compiles and links it on the CPU box.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../nested_hashing_psi_amd/host/ShardedBatchedFHEPSIServer.hpp"

int main(int argc, char **argv)
{
    if (argc != 12) return 2;
    try {
        const int rank = std::atoi(argv[1]), nranks = std::atoi(argv[2]), device = std::atoi(argv[3]), client = std::atoi(argv[4]);
        std::vector<int> side;
        if (std::strcmp(argv[5], "-"))
            for (const char *p = argv[5]; *p;) {
                char *end = nullptr;
                side.push_back((int)std::strtol(p, &end, 10));
                p = (*end == ',') ? end + 1 : end;
            }
        std::ifstream f(argv[6], std::ios::binary | std::ios::ate);
        const size_t bytes = (size_t)f.tellg();
        f.seekg(0);
        std::vector<uint64_t> set(bytes / 8);
        f.read(reinterpret_cast<char *>(set.data()), (std::streamsize)(set.size() * 8));
        piehip::HashTableParameter ht;
        ht.numberOfSimpleHashFunctions = (uint32_t)std::atoi(argv[7]);
        ht.eachSimpleTableSize = (uint32_t)std::atoi(argv[8]);
        ht.numberOfCuckooHashFunctions = (uint32_t)std::atoi(argv[9]);
        ht.eachCuckooTableSize = (uint32_t)std::atoi(argv[10]);
        ht.maxItemsPerPosition = (uint32_t)std::atoi(argv[11]);
        piehip::ShardedBatchedFHEPSIServer server(rank, nranks, device, client, side, set, ht);
        // tests only: fixed table secrets (so that the result ciphertexts can be compared with the oracle's bit for bit), a shorter
        // bound on the waits for the other ranks, and a rank that fails its offline phase
        if (const char *sd = std::getenv("PIEHIP_TEST_SEEDS")) {
            unsigned long long a = 0, b = 0, c = 0;
            if (std::sscanf(sd, "%llu,%llu,%llu", &a, &b, &c) == 3) server.setSecretSeedsForTesting(a, b, c);
        }
        if (const char *tm = std::getenv("PIEHIP_TEST_TIMEOUT_MS")) server.collectiveTimeoutMs = (uint32_t)std::atoi(tm);
        if (const char *fr = std::getenv("PIEHIP_TEST_FAIL_OFFLINE_RANK"))
            if (std::atoi(fr) == rank) server.failOfflineForTesting = true;
        server.run();
        if (rank == 0) std::printf("OfflineComputation,%lld\nOnlineComputation,%lld\n", server.offlineComputation, server.onlineComputation);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "server rank %s: %s\n", argv[1], e.what());
        return 1;
    }
}
