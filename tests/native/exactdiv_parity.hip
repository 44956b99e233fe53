// Test-only program for rtow::exact_div3 (raytracing-in-one-weekend_amd/csrc/rtow_exactmath.hip.h).
//
// Part 1 - is a / b = RN(q + (a - b q) y) with y = RN(1 / b), q = RN(a y) - one residual step on a correctly rounded reciprocal
// (rtow::exact_div_step) - the IEEE quotient?  The two-operand space cannot be enumerated (2^64), but the question does not depend on the
// exponents as long as no intermediate leaves the normal range: every step scales exactly by powers of two.  So it is enumerated over
// MANTISSAS: all 2^23 x 2^23 pairs a = 1.ma, b = 1.mb (quotients in (1/2, 2)), on the device, against the compiler's `a / b`; about half a
// minute of an MI355X.  argv[1] = log2 of the number of b mantissas to cover (default 23 = all).
// Part 2 - the function itself, range guard and fallback included, on 2^32 pseudo-random operand quadruples of arbitrary bit patterns
// (every exponent, zeros, subnormals, infinities, NaNs) and on 2^32 with exponents near the edges of the fast range.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../raytracing-in-one-weekend_amd/csrc/rtow_exactmath.hip.h"

__device__ __forceinline__ bool same(float a, float b)
{
    if (a != a && b != b) return true;                              // both NaN (payloads are not part of the contract)
    return __float_as_uint(a) == __float_as_uint(b);
}

__global__ void sweep_mantissas(unsigned bCount, unsigned long long* bad, unsigned* firstBad)
{
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned long long b1 = 0;
    for (unsigned mb = blockIdx.x * blockDim.x + threadIdx.x; mb < bCount; mb += stride) {
        // spread the covered b mantissas over the whole range when only a part is asked for
        const unsigned mbits = bCount == (1u << 23) ? mb : (unsigned)(((unsigned long long)mb << 23) / bCount) | (mb & 1u);
        const float b = __uint_as_float(0x3f800000u | (mbits & 0x7fffffu));
        const float y = rtow::exact_rcp(b);
        volatile float vb = b;
        for (unsigned ma = 0; ma < (1u << 23); ma++) {
            const float a = __uint_as_float(0x3f800000u | ma);
            const float want = a / vb;
            if (__float_as_uint(rtow::exact_div_step(a, b, y)) != __float_as_uint(want)) { if (!b1) { atomicCAS(&firstBad[0], 0u, ma); atomicCAS(&firstBad[1], 0u, mbits); } b1++; }
        }
    }
    if (b1) atomicAdd(&bad[0], b1);
}

__device__ __forceinline__ unsigned mix(unsigned long long& s)     // splitmix64, upper half
{
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}

__global__ void sweep_random(int edges, unsigned perThread, unsigned long long* bad, unsigned* firstBad)
{
    unsigned long long s = 0x1234567ull + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x100000001B3ull + (edges ? 77u : 0u);
    unsigned long long n = 0;
    for (unsigned k = 0; k < perThread; k++) {
        unsigned u[4];
        for (int i = 0; i < 4; i++) {
            u[i] = mix(s);
            if (edges) {
                // exponents within +-3 of the ends of the two fast ranges (64, 191 for numerators; 96, 159 for the divisor), and a few zeros
                const unsigned r = mix(s);
                const unsigned ends[4] = {64u, 191u, 96u, 159u};
                const unsigned e = ends[(i == 3 ? 2u : 0u) + (r & 1u)] + ((r >> 1) % 7u) - 3u;
                u[i] = (u[i] & 0x807fffffu) | (e << 23);
                if ((r >> 8) % 61u == 0u) u[i] &= 0x80000000u;
            }
        }
        const float ax = __uint_as_float(u[0]), ay = __uint_as_float(u[1]), az = __uint_as_float(u[2]), b = __uint_as_float(u[3]);
        float qx, qy, qz;
        rtow::exact_div3(ax, ay, az, b, qx, qy, qz);
        volatile float vb = b;
        if (!same(qx, ax / vb) || !same(qy, ay / vb) || !same(qz, az / vb)) { if (!n) { atomicCAS(&firstBad[0], 0u, u[0]); atomicCAS(&firstBad[1], 0u, u[3]); } n++; }
    }
    if (n) atomicAdd(&bad[0], n);
}

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 23;
    const unsigned bCount = 1u << (lg < 1 ? 1 : lg > 23 ? 23 : lg);
    unsigned long long* bad; unsigned* first;
    if (hipMalloc(&bad, 16) != hipSuccess || hipMalloc(&first, 16) != hipSuccess) return 2;
    unsigned long long h[2]; unsigned f[4];
    (void)hipMemset(bad, 0, 16); (void)hipMemset(first, 0, 16);
    sweep_mantissas<<<4096, 256>>>(bCount, bad, first);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
    printf("mantissa pairs %llu: %llu mismatches (first ma %u mb %u)\n", (unsigned long long)bCount << 23, h[0], f[0], f[1]);
    for (int edges = 0; edges < 2; edges++) {
        (void)hipMemset(bad, 0, 16); (void)hipMemset(first, 0, 16);
        sweep_random<<<4096, 256>>>(edges, 4096u, bad, first);                       // 2^20 threads x 2^12 quadruples
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
        (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
        printf("%s quadruples %llu: %llu mismatches (first a %08x b %08x)\n", edges ? "edge-exponent" : "random", 1ull << 32, h[0], f[0], f[1]);
    }
    fflush(stdout);
    return 0;
}
