#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32; typedef unsigned __int128 u128;
__device__ __forceinline__ u64 mad_u(u32 a, u32 b, u64 c) { u64 d, carry; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ u64 mul_u(u32 a, u32 b) { u64 d, carry; asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=&v"(d), "=s"(carry) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u64 shoup4(u64 b, u64 w, u64 ws, u64 nq)
{
    const u32 bl = (u32)b, bh = (u32)(b >> 32), wl = (u32)w, wh = (u32)(w >> 32);
    const u32 sl = (u32)ws, sh = (u32)(ws >> 32), nql = (u32)nq, nqh = (u32)(nq >> 32);
    const u64 m1 = mul_u(bl, sh);          // < 2^63  (sh < 2^31: ws is the 63-bit Shoup constant)
    const u64 cr = mad_u(bh, sl, m1);      // both cross terms, < 2^64 for b < 2^63
    const u64 top = mul_u(bh, sh);
    const u64 qe = (top << 1) + (cr >> 31);
    u64 acc = mul_u((u32)qe, nql);
    acc = mad_u(bl, wl, acc);
    u64 c = mul_u((u32)qe, nqh);
    c = mad_u((u32)(qe >> 32), nql, c);
    c = mad_u(bl, wh, c);
    c = mad_u(bh, wl, c);
    return acc + ((u64)(u32)c << 32);
}
__global__ void k(const u64* b, const u64* w, const u64* ws, u64 nq, u64* out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = shoup4(b[i], w[i], ws[i], nq);
}
int main() {
    const int n = 1 << 16; u64 q = 1152921504606830593ULL;
    std::vector<u64> b(n), w(n), ws(n), out(n);
    u64 s = 88172645463325252ULL;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; b[i] = s % (8 * q);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = s % q;
        ws[i] = (u64)(((u128)w[i] << 64) / q) >> 1;
    }
    b[0] = 0; b[1] = 4 * q; b[2] = 8 * q - 1; b[3] = q; b[4] = 5;
    u64 *db, *dw, *dws, *dout;
    hipMalloc(&db, n * 8); hipMalloc(&dw, n * 8); hipMalloc(&dws, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dws, ws.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, db, dw, dws, 0 - q, dout, n);
    hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0, maxk = 0;
    for (int i = 0; i < n; i++) {
        u64 want = (u64)(((u128)b[i] * w[i]) % q);
        u64 got = out[i];
        if (got % q != want || got >= 4 * q) { if (bad < 5) printf("i=%d b=%llu w=%llu got=%llu want=%llu\n", i, (unsigned long long)b[i], (unsigned long long)w[i], (unsigned long long)got, (unsigned long long)want); bad++; }
        int kk = (int)(got / q); if (kk > maxk) maxk = kk;
    }
    printf("shoup4: %d bad of %d, max lazy multiple %d\n", bad, n, maxk);
    return bad != 0;
}
