// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so.1 that lets several ranks share ONE GPU.
//
// The test box has one MI355X and RCCL refuses two ranks on one device, so the multi-rank branches of csrc/piehip_rccl.cpp (the
// root's per-peer receives, a worker's send, grouped broadcasts, root != 0, uneven bin slices) and the worker side of
// host/ShardedBatchedFHEPSIServer.hpp could never execute there.  libpiehip binds RCCL by name at run time (dlopen of
// "librccl.so.1", the copy already in the process first): a test process that preloads THIS library -- built by the test itself
// into a temporary directory, never installed, never part of the package (tests/test_abi.py checks that) -- gets the twelve entry
// points below with RCCL's semantics over Unix-domain sockets:
//   * calls return at once; the transfer is ordered in the HIP stream it was given (device -> pinned bounce buffer, a host
//     function queued in the stream that moves the bytes through the socket, bounce buffer -> device), so a peer that never
//     shows up blocks the STREAM, not the caller -- as a RCCL kernel would -- and ncclCommAbort releases it;
//   * ncclGroupStart / ncclGroupEnd defer and fuse, ncclCommGetAsyncError reports a failed or aborted transfer.
// What it says nothing about: RCCL itself, xGMI, bandwidth, CU occupancy.  Results moved through it are compared with the oracle.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Kind { K_SEND, K_RECV, K_BCAST, K_ALLREDUCE };

struct Op {
    Kind kind;
    const void *src;
    void *dst;
    size_t bytes, count;
    int peer;  // peer (send / recv) or root (broadcast)
    ncclDataType_t dt;
    ncclRedOp_t red;
    void *bounce;
};

struct Batch;
}  // namespace

struct ncclComm {  // (rccl.h declares the tag; the real layout is RCCL's business)
    int nranks = 0, rank = 0;
    std::vector<int> fd;
    std::atomic<int> aborted{0};
    std::atomic<int> async_err{0};
    std::mutex m;
    std::vector<Batch *> retired;
};

namespace {

struct Batch {
    ncclComm *c;
    std::vector<Op> ops;
    hipEvent_t done = nullptr;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local ncclComm *g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;

size_t dt_size(ncclDataType_t dt)
{
    switch (dt) {
    case ncclInt8: case ncclUint8: case ncclFloat8e4m3: case ncclFloat8e5m2: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

bool io_all(ncclComm *c, int fd, void *buf, size_t n, bool wr)
{
    char *p = (char *)buf;
    while (n) {
        if (c->aborted.load()) return false;
        struct pollfd pf = {fd, (short)(wr ? POLLOUT : POLLIN), 0};
        const int pr = poll(&pf, 1, 50);
        if (pr < 0 && errno != EINTR) return false;
        if (pr <= 0) continue;
        const ssize_t k = wr ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
        if (k == 0 && !wr) return false;  // the peer is gone
        if (k < 0) {
            if (errno == EINTR || errno == EAGAIN) continue;
            return false;
        }
        p += k, n -= (size_t)k;
    }
    return true;
}

struct Hdr {
    uint64_t bytes;
    uint32_t kind, from;
};
bool put(ncclComm *c, int peer, Kind k, const void *buf, size_t n)
{
    Hdr h = {n, (uint32_t)k, (uint32_t)c->rank};
    return io_all(c, c->fd[peer], &h, sizeof(h), true) && io_all(c, c->fd[peer], (void *)buf, n, true);
}
bool get(ncclComm *c, int peer, Kind k, void *buf, size_t n)
{
    Hdr h;
    if (!io_all(c, c->fd[peer], &h, sizeof(h), false)) return false;
    if (h.bytes != n || h.kind != (uint32_t)k || h.from != (uint32_t)peer) {
        fprintf(stderr, "fake rccl rank %d: expected %zu bytes of kind %d from %d, the peer sent %llu of kind %u (from %u): the ranks' calls do not match\n",
                c->rank, n, (int)k, peer, (unsigned long long)h.bytes, h.kind, h.from);
        return false;
    }
    return io_all(c, c->fd[peer], buf, n, false);
}

template <typename T>
void reduce_t(T *acc, const T *x, size_t n, ncclRedOp_t op)
{
    for (size_t i = 0; i < n; i++) {
        if (op == ncclSum) acc[i] = (T)(acc[i] + x[i]);
        else if (op == ncclMax) acc[i] = acc[i] > x[i] ? acc[i] : x[i];
        else if (op == ncclMin) acc[i] = acc[i] < x[i] ? acc[i] : x[i];
        else if (op == ncclProd) acc[i] = (T)(acc[i] * x[i]);
    }
}
void reduce(void *acc, const void *x, size_t n, ncclDataType_t dt, ncclRedOp_t op)
{
    switch (dt) {
    case ncclInt32: reduce_t((int32_t *)acc, (const int32_t *)x, n, op); break;
    case ncclUint32: reduce_t((uint32_t *)acc, (const uint32_t *)x, n, op); break;
    case ncclInt64: reduce_t((int64_t *)acc, (const int64_t *)x, n, op); break;
    case ncclUint64: reduce_t((uint64_t *)acc, (const uint64_t *)x, n, op); break;
    default: break;
    }
}

// runs on the HIP runtime's callback thread, in stream order: no HIP calls in here
void host_run(void *arg)
{
    Batch *b = (Batch *)arg;
    ncclComm *c = b->c;
    bool ok = !c->aborted.load() && !c->async_err.load();
    for (Op &o : b->ops) {
        if (!ok) break;
        switch (o.kind) {
        case K_SEND: ok = put(c, o.peer, K_SEND, o.bounce, o.bytes); break;
        case K_RECV: ok = get(c, o.peer, K_SEND, o.bounce, o.bytes); break;
        case K_BCAST:
            if (c->rank == o.peer) {
                for (int r = 0; r < c->nranks && ok; r++)
                    if (r != c->rank) ok = put(c, r, K_BCAST, o.bounce, o.bytes);
            } else {
                ok = get(c, o.peer, K_BCAST, o.bounce, o.bytes);
            }
            break;
        case K_ALLREDUCE:
            if (c->rank == 0) {
                std::vector<char> tmp(o.bytes);
                for (int r = 1; r < c->nranks && ok; r++) {
                    ok = get(c, r, K_ALLREDUCE, tmp.data(), o.bytes);
                    if (ok) reduce(o.bounce, tmp.data(), o.count, o.dt, o.red);
                }
                for (int r = 1; r < c->nranks && ok; r++) ok = put(c, r, K_ALLREDUCE, o.bounce, o.bytes);
            } else {
                ok = put(c, 0, K_ALLREDUCE, o.bounce, o.bytes) && get(c, 0, K_ALLREDUCE, o.bounce, o.bytes);
            }
            break;
        }
    }
    if (!ok && !c->async_err.load()) c->async_err.store(c->aborted.load() ? (int)ncclInternalError : (int)ncclRemoteError);
}

void free_batch(Batch *b)
{
    for (Op &o : b->ops)
        if (o.bounce) (void)hipHostFree(o.bounce);
    if (b->done) (void)hipEventDestroy(b->done);
    delete b;
}

ncclResult_t submit(ncclComm *c, std::vector<Op> &ops, hipStream_t st)
{
    if (c->aborted.load()) return ncclInvalidUsage;
    {  // batches whose last copy has completed give their bounce buffers back
        std::lock_guard<std::mutex> lock(c->m);
        size_t keep = 0;
        for (Batch *b : c->retired) {
            if (hipEventQuery(b->done) == hipSuccess) free_batch(b);
            else c->retired[keep++] = b;
        }
        c->retired.resize(keep);
        (void)hipGetLastError();
    }
    Batch *b = new Batch;
    b->c = c;
    b->ops.swap(ops);
    for (Op &o : b->ops) {
        if (hipHostMalloc(&o.bounce, o.bytes ? o.bytes : 8, hipHostMallocPortable) != hipSuccess) return ncclUnhandledCudaError;
        const bool out = o.kind == K_SEND || o.kind == K_ALLREDUCE || (o.kind == K_BCAST && c->rank == o.peer);
        if (out && o.bytes && hipMemcpyAsync(o.bounce, o.src, o.bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipLaunchHostFunc(st, host_run, b) != hipSuccess) return ncclUnhandledCudaError;
    for (Op &o : b->ops) {
        const bool in = o.kind == K_RECV || o.kind == K_ALLREDUCE || (o.kind == K_BCAST && c->rank != o.peer);
        if (in && o.bytes && hipMemcpyAsync(o.dst, o.bounce, o.bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipEventCreateWithFlags(&b->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(b->done, st) != hipSuccess)
        return ncclUnhandledCudaError;
    std::lock_guard<std::mutex> lock(c->m);
    c->retired.push_back(b);
    return ncclSuccess;
}

ncclResult_t enqueue(ncclComm *c, const Op &o, hipStream_t st)
{
    if (!c) return ncclInvalidArgument;
    if (g_depth) {
        if (g_comm && (g_comm != c || g_stream != st)) return ncclInvalidUsage;  // (one communicator and stream per group is all the tests need)
        g_comm = c, g_stream = st;
        g_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one(1, o);
    return submit(c, one, st);
}

std::string sock_path(const ncclUniqueId &id, int rank)
{
    char hex[33];
    for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    return std::string("/tmp/fakerccl-") + hex + "." + std::to_string(rank);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    const int f = open("/dev/urandom", O_RDONLY);
    if (f < 0 || read(f, id->internal, 16) != 16) return ncclSystemError;
    close(f);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->nranks = nranks, c->rank = rank;
    c->fd.assign(nranks, -1);
    int lis = -1;
    const std::string mine = sock_path(id, rank);
    if (nranks > 1) {
        lis = socket(AF_UNIX, SOCK_STREAM, 0);
        struct sockaddr_un a;
        memset(&a, 0, sizeof(a));
        a.sun_family = AF_UNIX;
        snprintf(a.sun_path, sizeof(a.sun_path), "%s", mine.c_str());
        unlink(mine.c_str());
        if (lis < 0 || bind(lis, (struct sockaddr *)&a, sizeof(a)) || listen(lis, nranks)) return ncclSystemError;
    }
    for (int r = 0; r < rank; r++) {  // connect to every lower rank (its listener may not exist yet: retry for a minute)
        const std::string path = sock_path(id, r);
        struct sockaddr_un a;
        memset(&a, 0, sizeof(a));
        a.sun_family = AF_UNIX;
        snprintf(a.sun_path, sizeof(a.sun_path), "%s", path.c_str());
        int s = -1;
        for (int tries = 0; tries < 6000; tries++) {
            s = socket(AF_UNIX, SOCK_STREAM, 0);
            if (s >= 0 && connect(s, (struct sockaddr *)&a, sizeof(a)) == 0) break;
            if (s >= 0) close(s);
            s = -1;
            struct timespec ts = {0, 10 * 1000 * 1000};
            nanosleep(&ts, nullptr);
        }
        if (s < 0) return ncclSystemError;
        const uint32_t me = (uint32_t)rank;
        if (send(s, &me, 4, MSG_NOSIGNAL) != 4) return ncclSystemError;
        c->fd[r] = s;
    }
    for (int n = rank + 1; n < nranks; n++) {  // ... and take the connections of the higher ones
        struct pollfd pf = {lis, POLLIN, 0};
        if (poll(&pf, 1, 60000) <= 0) return ncclSystemError;
        const int s = accept(lis, nullptr, nullptr);
        uint32_t who = 0;
        if (s < 0 || recv(s, &who, 4, MSG_WAITALL) != 4 || who >= (uint32_t)nranks || (int)who <= rank || c->fd[who] >= 0) return ncclSystemError;
        c->fd[who] = s;
    }
    if (lis >= 0) {
        close(lis);
        unlink(mine.c_str());
    }
    for (int r = 0; r < nranks; r++)
        if (c->fd[r] >= 0) {
            int sz = 4 << 20;
            (void)setsockopt(c->fd[r], SOL_SOCKET, SO_SNDBUF, &sz, sizeof(sz));
            (void)setsockopt(c->fd[r], SOL_SOCKET, SO_RCVBUF, &sz, sizeof(sz));
        }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    c->aborted.store(1);
    for (int f : c->fd)
        if (f >= 0) (void)shutdown(f, SHUT_RDWR);  // wakes a transfer blocked on the peer, here and over there
    return ncclSuccess;  // (the object stays: a host function of an earlier call may still be looking at it)
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    (void)hipDeviceSynchronize();
    for (Batch *b : c->retired) free_batch(b);
    for (int f : c->fd)
        if (f >= 0) close(f);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *e)
{
    if (!c || !e) return ncclInvalidArgument;
    *e = (ncclResult_t)c->async_err.load();
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    if (!g_ops.empty()) r = submit(g_comm, g_ops, g_stream);
    g_ops.clear();
    g_comm = nullptr;
    g_stream = nullptr;
    return r;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    Op o = {K_SEND, buf, nullptr, count * dt_size(dt), count, peer, dt, ncclSum, nullptr};
    return enqueue(c, o, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    Op o = {K_RECV, nullptr, buf, count * dt_size(dt), count, peer, dt, ncclSum, nullptr};
    return enqueue(c, o, st);
}

ncclResult_t ncclBroadcast(const void *src, void *dst, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t st)
{
    if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (c->nranks == 1) {
        if (src != dst && hipMemcpyAsync(dst, src, count * dt_size(dt), hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    Op o = {K_BCAST, src, dst, count * dt_size(dt), count, root, dt, ncclSum, nullptr};
    if (c->rank == root && src != dst && hipMemcpyAsync(dst, src, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    return enqueue(c, o, st);
}

ncclResult_t ncclAllReduce(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    if (!c) return ncclInvalidArgument;
    if (dt != ncclInt32 && dt != ncclUint32 && dt != ncclInt64 && dt != ncclUint64) return ncclInvalidArgument;
    Op o = {K_ALLREDUCE, src, dst, count * dt_size(dt), count, 0, dt, op, nullptr};
    return enqueue(c, o, st);
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
    case ncclSystemError: return "unhandled system error (fake rccl)";
    case ncclInternalError: return "internal error / aborted (fake rccl)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    case ncclRemoteError: return "remote process exited or the calls of the ranks do not match (fake rccl)";
    case ncclInProgress: return "in progress (fake rccl)";
    default: return "unknown result (fake rccl)";
    }
}

}  // extern "C"
