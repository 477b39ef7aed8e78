// The server process of the two-party test (tests/test_gpu_parity.py::test_two_process_psi_over_the_wire):
//   server_main <socket fd[,socket fd ...]> <server set file (raw uint64)> k e K E b
// (several descriptors: one client each, their queries evaluated as one batch)
// It runs host/BatchedFHEPSIServer.hpp's three phases over the inherited socket and prints the reference's timing keys.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "../nested_hashing_psi_amd/host/BatchedFHEPSIServer.hpp"

int main(int argc, char **argv)
{
    if (argc != 8) return 2;
    try {
        std::vector<int> fds;
        for (const char *p = argv[1]; *p;) {
            char *end = nullptr;
            fds.push_back((int)std::strtol(p, &end, 10));
            p = (*end == ',') ? end + 1 : end;
        }
        std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
        const size_t bytes = (size_t)f.tellg();
        f.seekg(0);
        std::vector<uint64_t> set(bytes / 8);
        f.read(reinterpret_cast<char *>(set.data()), (std::streamsize)(set.size() * 8));
        piehip::HashTableParameter ht;
        ht.numberOfSimpleHashFunctions = (uint32_t)std::atoi(argv[3]);
        ht.eachSimpleTableSize = (uint32_t)std::atoi(argv[4]);
        ht.numberOfCuckooHashFunctions = (uint32_t)std::atoi(argv[5]);
        ht.eachCuckooTableSize = (uint32_t)std::atoi(argv[6]);
        ht.maxItemsPerPosition = (uint32_t)std::atoi(argv[7]);
        piehip::BatchedFHEPSIServer server(fds, set, ht);
        server.run();
        std::printf("OfflineComputation,%lld\nOnlineComputation,%lld\n", server.offlineComputation, server.onlineComputation);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "server: %s\n", e.what());
        return 1;
    }
}
