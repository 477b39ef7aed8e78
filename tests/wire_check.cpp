// Framing round trip over a socket pair (run by tests/test_abi.py; no GPU, no library needed): the message sequence of the
// reference's online phase -- minus element, K*E index ciphertexts one message each, phase barrier, b results back.
#include <sys/socket.h>
#include <cstdio>
#include <thread>

#include "../nested_hashing_psi_amd/host/WireFraming.hpp"

int main()
{
    using namespace piehip::wire;
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 1;
    const uint32_t L = 2, N = 1024, KE = 6, b = 3;
    const size_t ct = 2 * (size_t)L * N;
    std::vector<uint64_t> idx(KE * ct), res(b * ct);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i * 0x9E3779B97F4A7C15ULL;
    for (size_t i = 0; i < res.size(); i++) res[i] = ~i * 0xD1B54A32D192ED03ULL;
    int bad = 0;
    std::thread client([&] {
        try {
            writeWithSize(sv[0], packCiphertexts(idx.data(), 1, L, N).data(), sizeof(LimbHeader) + ct * 8);  // "minus" ciphertext
            for (uint32_t j = 0; j < KE; j++) {
                auto m = packCiphertexts(&idx[j * ct], 1, L, N);
                writeWithSize(sv[0], m.data(), m.size());
            }
            signalPhaseOver(sv[0]);
            std::vector<uint8_t> m;
            std::vector<uint64_t> got;
            readWithSizeIntoVector(sv[0], m);
            if (unpackCiphertexts(m, L, N, got) != b || got != res) bad |= 1;
        } catch (const std::exception &e) {
            std::printf("client: %s\n", e.what());
            bad |= 2;
        }
    });
    try {
        std::vector<uint8_t> m;
        std::vector<uint64_t> got, all;
        readWithSizeIntoVector(sv[1], m);
        if (unpackCiphertexts(m, L, N, got) != 1) bad |= 4;
        for (uint32_t j = 0; j < KE; j++) {
            readWithSizeIntoVector(sv[1], m);
            if (unpackCiphertexts(m, L, N, got) != 1) bad |= 4;
            all.insert(all.end(), got.begin(), got.end());
        }
        waitForPhaseOver(sv[1]);
        if (all != idx) bad |= 8;
        auto out = packCiphertexts(res.data(), b, L, N);
        writeWithSize(sv[1], out.data(), out.size());
        bool threw = false;
        try {
            std::vector<uint8_t> shortmsg(8, 0);
            unpackCiphertexts(shortmsg, L, N, got);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        if (!threw) bad |= 16;
        // a header that announces more ciphertexts than the message carries is refused before anything is sized by it
        threw = false;
        try {
            const LimbHeader lie = {0x48454950u, 0x7fffffffu, L, N, 0};
            std::vector<uint8_t> m24(sizeof(lie));
            std::memcpy(m24.data(), &lie, sizeof(lie));
            unpackCiphertexts(m24, L, N, got);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        if (!threw) bad |= 128;
        // residues must be canonical: one word at its modulus is rejected, all words below it are accepted
        {
            const uint64_t q[2] = {1000003, 1000033};
            std::vector<uint64_t> ok(ct);
            for (size_t i = 0; i < ok.size(); i++) ok[i] = i % 1000003;
            if (unpackCiphertexts(packCiphertexts(ok.data(), 1, L, N), L, N, got, q) != 1) bad |= 64;
            ok[3 * (size_t)N + 5] = q[1];  // limb 3 = component 1, tower 1
            threw = false;
            try {
                unpackCiphertexts(packCiphertexts(ok.data(), 1, L, N), L, N, got, q);
            } catch (const std::invalid_argument &) {
                threw = true;
            }
            if (!threw) bad |= 64;
        }
    } catch (const std::exception &e) {
        std::printf("server: %s\n", e.what());
        bad |= 32;
    }
    client.join();
    std::printf(bad ? "wire check FAILED (%d)\n" : "wire check ok\n", bad);
    return bad;
}
