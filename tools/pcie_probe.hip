// What the PCIe link gives the host-memory query path: H2D of a query's pieces (rows of 14 MiB), D2H of a result list, both at
// once, for the page-locked allocation flavours HIP offers, and for a copy KERNEL reading the host array instead of the DMA engine.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/.pcie_probe_bin tools/pcie_probe.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void copy_kernel(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t MiB = 1 << 20, row = 14 * MiB, up_bytes = 6 * row + 3 * MiB, dn_bytes = 42 * MiB;  // a batch of three C3 queries
    struct Flavour {
        const char *name;
        unsigned flags;
    } fl[] = {{"default", hipHostMallocDefault},
              {"portable", hipHostMallocPortable},
              {"non-coherent", hipHostMallocNonCoherent},
              {"coherent", hipHostMallocCoherent},
              {"write-combined", hipHostMallocWriteCombined},
              {"numa-user", hipHostMallocNumaUser}};
    void *d_up = nullptr, *d_dn = nullptr;
    CK(hipMalloc(&d_up, up_bytes));
    CK(hipMalloc(&d_dn, dn_bytes));
    CK(hipMemset(d_dn, 1, dn_bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (const Flavour &f : fl) {
        void *h_up = nullptr, *h_dn = nullptr;
        if (hipHostMalloc(&h_up, up_bytes, f.flags) != hipSuccess || hipHostMalloc(&h_dn, dn_bytes, f.flags) != hipSuccess) {
            std::printf("%-15s allocation refused\n", f.name);
            (void)hipGetLastError();
            continue;
        }
        std::memset(h_up, 3, up_bytes);
        std::memset(h_dn, 0, dn_bytes);
        auto up = [&] {
            size_t off = 0;
            for (int i = 0; i < 3; i++, off += MiB) (void)hipMemcpyAsync((char *)d_up + off, (char *)h_up + off, MiB, hipMemcpyHostToDevice, s1);
            for (int i = 0; i < 6; i++, off += row) (void)hipMemcpyAsync((char *)d_up + off, (char *)h_up + off, row, hipMemcpyHostToDevice, s1);
        };
        auto down = [&] {
            (void)hipMemcpyAsync(h_dn, d_dn, 24 * MiB, hipMemcpyDeviceToHost, s2);
            (void)hipMemcpyAsync((char *)h_dn + 24 * MiB, (char *)d_dn + 24 * MiB, 18 * MiB, hipMemcpyDeviceToHost, s2);
        };
        auto timed = [&](int what) {
            const int n = 20;
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                if (what & 1) up();
                if (what & 2) down();
                (void)hipDeviceSynchronize();
                const double t0 = now();
                for (int i = 0; i < n; i++) {
                    if (what & 1) up();
                    if (what & 2) down();
                }
                (void)hipDeviceSynchronize();
                best = std::min(best, (now() - t0) / n);
            }
            return best;
        };
        const double tu = timed(1), td = timed(2), tb = timed(3);
        // the copy kernel: the GPU reads the page-locked array itself
        void *h_dev = nullptr;
        double tk = 0;
        if (hipHostGetDevicePointer(&h_dev, h_up, 0) == hipSuccess) {
            for (int rep = 0; rep < 2; rep++) {
                (void)hipDeviceSynchronize();
                const double t0 = now();
                for (int i = 0; i < 10; i++)
                    hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, s1, (const ulonglong2 *)h_dev, (ulonglong2 *)d_up, up_bytes / 16);
                (void)hipDeviceSynchronize();
                tk = (now() - t0) / 10;
            }
        }
        std::printf("%-15s H2D 87 MiB %.3f ms (%.1f GB/s)  D2H 42 MiB %.3f ms (%.1f GB/s)  both %.3f ms (up+down %.1f GB/s; per C3 query %.3f ms)  "
                    "copy kernel H2D %.3f ms (%.1f GB/s)\n",
                    f.name, tu * 1e3, up_bytes / tu / 1e9, td * 1e3, dn_bytes / td / 1e9, tb * 1e3, (up_bytes + dn_bytes) / tb / 1e9, tb * 1e3 / 3,
                    tk * 1e3, tk > 0 ? up_bytes / tk / 1e9 : 0.0);
        (void)hipHostFree(h_up);
        (void)hipHostFree(h_dn);
    }
    return 0;
}
