// WireFraming.hpp -- the message framing of the reference's TCP channel, for a server that keeps libpiehip behind it.
//
// The reference moves every object as one size-prefixed message of libscapi's CommPartyTCPSynced:
//     channel->writeWithSize(string)            src/Server/FHE/BatchedFHEPSIServer.cpp:150, PSIServer.hpp:46-49
//     channel->readWithSizeIntoVector(vector)   BatchedFHEPSIServer.cpp:26,36,47,118,134
// i.e. a 4-byte native-endian int length followed by the payload [libscapi source is not part of the reference tree:
// UNVERIFIED], and marks the end of a protocol phase with an empty message (PSIServer.hpp:46-49, PSIClient.hpp:50-54).
// The payloads of the reference are OpenFHE cereal BINARY blobs (BatchedFHEHIPPIE.hpp:14-16); restating that object graph
// needs OpenFHE, so a drop-in keeps OpenFHE for (de)serialisation and hands DCRTPoly towers to the C ABI.
// This header carries the framing over any connected stream socket / pipe, plus a flat limb-array payload
// ([count][2][L][N] uint64 with a 24-byte header) for deployments that do not need OpenFHE's format on the wire.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unistd.h>
#include <vector>

namespace piehip {
namespace wire {

inline void write_all(int fd, const void *buf, size_t len)
{
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    while (len) {
        const ssize_t n = ::write(fd, p, len);
        if (n < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error(std::string("write: ") + std::strerror(errno));
        }
        p += n;
        len -= (size_t)n;
    }
}
inline void read_all(int fd, void *buf, size_t len)
{
    uint8_t *p = static_cast<uint8_t *>(buf);
    while (len) {
        const ssize_t n = ::read(fd, p, len);
        if (n < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error(std::string("read: ") + std::strerror(errno));
        }
        if (n == 0) throw std::runtime_error("read: connection closed inside a message");
        p += n;
        len -= (size_t)n;
    }
}

// writeWithSize: 4-byte native-endian int length, then the bytes (messages are limited to INT32_MAX bytes, as there)
inline void writeWithSize(int fd, const void *data, size_t len)
{
    if (len > 0x7fffffffu) throw std::invalid_argument("message exceeds the 31-bit length field");
    const int32_t n = (int32_t)len;
    write_all(fd, &n, sizeof(n));
    if (len) write_all(fd, data, len);
}
inline void writeWithSize(int fd, const std::string &s) { writeWithSize(fd, s.data(), s.size()); }

// readWithSizeIntoVector: resizes the vector to the announced length
inline void readWithSizeIntoVector(int fd, std::vector<uint8_t> &out)
{
    int32_t n = 0;
    read_all(fd, &n, sizeof(n));
    if (n < 0) throw std::runtime_error("negative message length");
    out.resize((size_t)n);
    if (n) read_all(fd, out.data(), (size_t)n);
}

// phase barrier: an empty message (PSIServer::signalPhaseOver / PSIClient::waitForServer)
inline void signalPhaseOver(int fd) { writeWithSize(fd, nullptr, 0); }
inline void waitForPhaseOver(int fd)
{
    std::vector<uint8_t> m;
    readWithSizeIntoVector(fd, m);
    if (!m.empty()) throw std::runtime_error("expected the empty phase-barrier message");
}

// flat ciphertext payload: {magic "PIEH", count, L, N, reserved} then count x 2 x L x N uint64 (EVALUATION towers)
struct LimbHeader {
    uint32_t magic, count, L, N;
    uint64_t reserved;
};
inline std::vector<uint8_t> packCiphertexts(const uint64_t *limbs, uint32_t count, uint32_t L, uint32_t N)
{
    const size_t words = (size_t)count * 2 * L * N;
    std::vector<uint8_t> msg(sizeof(LimbHeader) + words * sizeof(uint64_t));
    const LimbHeader h = {0x48454950u, count, L, N, 0};
    std::memcpy(msg.data(), &h, sizeof(h));
    std::memcpy(msg.data() + sizeof(h), limbs, words * sizeof(uint64_t));
    return msg;
}
// every residue of limb i must be canonical, i.e. below moduli[i % L] -- the device kernels assume it (lazy accumulation), and
// what arrives here comes from the other party
inline void checkCanonical(const uint64_t *limbs, size_t nlimbs, uint32_t L, uint32_t N, const uint64_t *moduli, const char *what)
{
    for (size_t limb = 0; limb < nlimbs; limb++) {
        const uint64_t q = moduli[limb % L], *p = limbs + limb * N;
        uint64_t bad = 0;
        for (uint32_t j = 0; j < N; j++) bad |= (uint64_t)(p[j] >= q);
        if (bad) throw std::invalid_argument(std::string(what) + " residue not below its modulus");
    }
}
// Unpacks straight into caller memory (e.g. the page-locked staging array the upload starts from): dst[count][2][L][N] must
// hold `expect` ciphertexts.  Throws if the message does not describe [expect][2][L][N] or (moduli != null: q_0..q_{L-1}) a
// residue is not canonical.
// header of a ciphertext message, validated against the context and the message length BEFORE anything is sized by it (the
// count comes from the other party)
inline LimbHeader checkedHeader(const std::vector<uint8_t> &msg, uint32_t L, uint32_t N)
{
    if (msg.size() < sizeof(LimbHeader)) throw std::invalid_argument("short ciphertext message");
    LimbHeader h;
    std::memcpy(&h, msg.data(), sizeof(h));
    if (h.magic != 0x48454950u || h.L != L || h.N != N) throw std::invalid_argument("ciphertext message does not match the context");
    const size_t per_ct = (size_t)2 * L * N * sizeof(uint64_t), body = msg.size() - sizeof(LimbHeader);
    if (per_ct == 0 || body % per_ct != 0 || body / per_ct != h.count) throw std::invalid_argument("ciphertext message length mismatch");
    return h;
}
inline void unpackCiphertextsInto(const std::vector<uint8_t> &msg, uint32_t L, uint32_t N, uint64_t *dst, uint32_t expect,
                                  const uint64_t *moduli = nullptr)
{
    const LimbHeader h = checkedHeader(msg, L, N);
    if (h.count != expect) throw std::invalid_argument("ciphertext message does not match the context");
    const size_t words = (size_t)h.count * 2 * L * N;
    std::memcpy(dst, msg.data() + sizeof(h), words * sizeof(uint64_t));
    if (moduli) checkCanonical(dst, (size_t)h.count * 2 * L, L, N, moduli, "ciphertext");
}
// returns the ciphertext count; throws if the message does not describe [count][2][L][N]
inline uint32_t unpackCiphertexts(const std::vector<uint8_t> &msg, uint32_t L, uint32_t N, std::vector<uint64_t> &limbs,
                                  const uint64_t *moduli = nullptr)
{
    const LimbHeader h = checkedHeader(msg, L, N);   // the count is bounded by the message that actually arrived
    limbs.resize((size_t)h.count * 2 * L * N);
    unpackCiphertextsInto(msg, L, N, limbs.data(), h.count, moduli);
    return h.count;
}

}  // namespace wire
}  // namespace piehip
