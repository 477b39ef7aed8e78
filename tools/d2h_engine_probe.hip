// Which engine does hipMemcpyAsync(device -> page-locked host) use: the SDMA engine or a blit kernel (__amd_rocclr_copyBuffer)?
// A blit kernel occupies wave slots on every CU for the whole (PCIe-bound) transfer; the persistent transform kernels need the CUs'
// whole register files, so their workgroups cannot be placed beside it.   rocprofv3 --kernel-trace -- ./d2h_engine_probe <mode>
//   mode 0: copy on the same stream, straight behind a kernel          mode 1: copy on a second stream behind an event wait
//   mode 2: copy on the same stream after the stream has been synchronised
//   mode 3: two streams, each kernel -> copy, at the same time        mode 4: uploads on a third stream first, then as mode 3 behind a fork
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void busy(unsigned long long *p, int n)
{
    unsigned long long v = p[threadIdx.x];
    for (int i = 0; i < n; i++) v = v * 6364136223846793005ULL + 1442695040888963407ULL;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = 8 << 20;
    unsigned long long *d, *h;
    hipMalloc((void **)&d, bytes);
    hipHostMalloc((void **)&h, bytes, hipHostMallocPortable);
    unsigned long long *d2, *h2;
    hipMalloc((void **)&d2, bytes);
    hipHostMalloc((void **)&h2, bytes, hipHostMallocPortable);
    hipStream_t s0, s1, s2;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, s1, d, 20000);
        if (mode == 0) {
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s1);
        } else if (mode == 1) {
            hipEventRecord(e, s1);
            hipStreamWaitEvent(s2, e, 0);
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s2);
        } else if (mode == 2) {
            hipStreamSynchronize(s1);
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s1);
        } else {
            if (mode == 4) {
                for (int i = 0; i < 8; i++) hipMemcpyAsync((char *)d2 + i * (1 << 20), (char *)h2 + i * (1 << 20), 1 << 20, hipMemcpyHostToDevice, s0);
                hipEventRecord(e, s0);
                hipStreamWaitEvent(s1, e, 0);
                hipStreamWaitEvent(s2, e, 0);
            }
            hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, s2, d2, 30000);
            hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s1);
            hipMemcpyAsync(h2, d2, bytes / 2, hipMemcpyDeviceToHost, s2);
        }
        hipDeviceSynchronize();
    }
    printf("mode %d done\n", mode);
    return 0;
}
