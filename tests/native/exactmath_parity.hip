// Test-only program: rtow::exact_rcp / exact_sqrt (raytracing-in-one-weekend_amd/csrc/rtow_exactmath.hip.h) against the compiler's IEEE
// `1.0f / x` and `__builtin_sqrtf(x)` for EVERY one of the 2^32 float operands, on the device; plus the composition the path uses for
// normalize (1 / sqrt(d)).  Built by tests/test_gpu_detmath.py with the product's own flags.  Prints the mismatch counts.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../raytracing-in-one-weekend_amd/csrc/rtow_exactmath.hip.h"

__device__ __forceinline__ bool same(float a, float b)
{
    if (a != a && b != b) return true;                              // both NaN (payloads are not part of the contract)
    return __float_as_uint(a) == __float_as_uint(b);
}

__global__ void sweep(unsigned long long* bad, unsigned* firstBad)
{
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (unsigned k = 0; k < (unsigned)((1ull << 32) / stride); k++, i += stride) {     // the grid size (2^20 threads) divides 2^32
        const float x = __uint_as_float(i);
        volatile float vx = x;                                      // keep the reference expressions from being folded with the candidates
        const float r0 = 1.0f / vx, r1 = __builtin_sqrtf(vx);
        if (!same(rtow::exact_rcp(x), r0)) { if (!b0) atomicCAS(&firstBad[0], 0u, i); b0++; }
        if (!same(rtow::exact_sqrt(x), r1)) { if (!b1) atomicCAS(&firstBad[1], 0u, i); b1++; }
        const float r2 = 1.0f / r1;
        if (!same(rtow::exact_rcp(rtow::exact_sqrt(x)), r2)) { if (!b2) atomicCAS(&firstBad[2], 0u, i); b2++; }
        const float r3 = r0 != r0 ? __builtin_inff() : r0;         // rayInvDirection: rcp, NaN -> +INF
        if (__float_as_uint(rtow::exact_rcp_nan_to_inf(x)) != __float_as_uint(r3)) { if (!b3) atomicCAS(&firstBad[3], 0u, i); b3++; }
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    if (b1) atomicAdd(&bad[1], (unsigned long long)b1);
    if (b2) atomicAdd(&bad[2], (unsigned long long)b2);
    if (b3) atomicAdd(&bad[3], (unsigned long long)b3);
}

int main()
{
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 32); hipMalloc(&first, 16);
    hipMemset(bad, 0, 32); hipMemset(first, 0, 16);
    // 2^32 operands over 2^20 threads: 4096 operands each (the grid size divides 2^32)
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long h[4]; unsigned f[4];
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost);
    printf("exactmath parity over 2^32 operands: rcp %llu mismatches (first 0x%08x), sqrt %llu (first 0x%08x), rcp(sqrt) %llu (first 0x%08x), rcp with NaN -> inf %llu (first 0x%08x)\n",
           h[0], f[0], h[1], f[1], h[2], f[2], h[3], f[3]);
    return (h[0] || h[1] || h[2] || h[3]) ? 1 : 0;
}
