// Where do page-locked allocations land (NUMA node) and what does that do to the upload rate?  The GPU box has two sockets; a
// staging array on the far socket crosses the inter-socket fabric on its way to the GPU.
//   hipcc -O2 --offload-arch=gfx950 -o tools/.numa_probe_bin tools/numa_pinned_probe.hip
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int node_of(void *p)
{
    int status = -1;
    void *pages[1] = {p};
    if (syscall(SYS_move_pages, 0, 1UL, pages, nullptr, &status, 0) != 0) return -2;
    return status;
}
int main()
{
    const size_t bytes = 87u << 20;
    void *d;
    hipMalloc(&d, bytes);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned flags[3] = {hipHostMallocDefault, hipHostMallocPortable, hipHostMallocNumaUser};
    const char *names[3] = {"default", "portable", "numa-user"};
    for (int round = 0; round < 3; round++)
        for (int f = 0; f < 3; f++) {
            // move this thread to another CPU before allocating, as a long-running host thread drifts
            cpu_set_t set;
            sched_getaffinity(0, sizeof(set), &set);
            std::vector<int> cpus;
            for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set)) cpus.push_back(c);
            const int target = cpus[(round * 7 + f * 3) % cpus.size()];
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(target, &one);
            sched_setaffinity(0, sizeof(one), &one);
            void *h = nullptr;
            if (hipHostMalloc(&h, bytes, flags[f]) != hipSuccess) { printf("alloc failed\n"); continue; }
            memset(h, 1, bytes);
            sched_setaffinity(0, sizeof(set), &set);
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipStreamSynchronize(s);
                const double t0 = now();
                hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
                hipStreamSynchronize(s);
                best = std::min(best, now() - t0);
            }
            printf("%-10s allocated on cpu %3d: pages on node %d (first) / %d (last)   H2D 87 MiB %.3f ms = %.1f GB/s\n", names[f], target,
                   node_of(h), node_of((char *)h + bytes - 4096), best * 1e3, bytes / best / 1e9);
            hipHostFree(h);
        }
    return 0;
}
