// Do two streams that have each copied in BOTH directions still get the link's two directions at once?
// (the in-order host path puts a query's uploads and its result download on the handle's own stream)
//   hipcc -O2 --offload-arch=gfx950 -o tools/.sdma_probe_bin tools/sdma_engine_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t MiB = 1 << 20, up = 87 * MiB, dn = 42 * MiB;
    const int NS = 4;
    hipStream_t s[NS];
    void *hu[NS], *hd[NS], *du[NS], *dd[NS];
    for (int i = 0; i < NS; i++) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        hipHostMalloc(&hu[i], up, hipHostMallocPortable);
        hipHostMalloc(&hd[i], dn, hipHostMallocPortable);
        hipMalloc(&du[i], up);
        hipMalloc(&dd[i], dn);
        memset(hu[i], 1, up);
    }
    auto H2D = [&](int i) { for (int k = 0; k < 6; k++) hipMemcpyAsync((char *)du[i] + k * 14 * MiB, (char *)hu[i] + k * 14 * MiB, 14 * MiB, hipMemcpyHostToDevice, s[i]); };
    auto D2H = [&](int i) { hipMemcpyAsync(hd[i], dd[i], dn, hipMemcpyDeviceToHost, s[i]); };
    auto timed = [&](const char *what, auto f) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipDeviceSynchronize();
            const double t0 = now();
            f();
            hipDeviceSynchronize();
            best = std::min(best, now() - t0);
        }
        printf("%-70s %.3f ms\n", what, best * 1e3);
    };
    timed("fresh streams: s0 uploads 87 MiB, s1 downloads 42 MiB, at once", [&] { H2D(0); D2H(1); });
    for (int i = 0; i < NS; i++) { H2D(i); D2H(i); }   // every stream has now copied both ways
    hipDeviceSynchronize();
    timed("both streams have copied both ways: s0 uploads, s1 downloads", [&] { H2D(0); D2H(1); });
    timed("                                     s1 uploads, s0 downloads", [&] { H2D(1); D2H(0); });
    timed("                                     s2 uploads, s3 downloads", [&] { H2D(2); D2H(3); });
    timed("                                     s0 uploads, s2 downloads", [&] { H2D(0); D2H(2); });
    timed("one stream: s0 uploads, then s0 downloads (in order)", [&] { H2D(0); D2H(0); });
    timed("s0 uploads alone", [&] { H2D(0); });
    timed("s1 downloads alone", [&] { D2H(1); });
    return 0;
}
