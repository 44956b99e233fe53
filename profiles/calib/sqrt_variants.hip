// sqrt_variants.hip - which short instruction sequence IS the correctly rounded square root on gfx950?  Every candidate against
// __builtin_sqrtf for all operands in the fast-path range of rtow::exact_sqrt (positive, biased exponent 2..253), on the device.
// Development probe (profiles/calib): the winner goes into csrc/rtow_exactmath.hip.h, whose own exhaustive test is tests/native/exactmath_parity.hip.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float v1(float x) { const float y = __builtin_amdgcn_rsqf(x); float s = x * y; const float h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s); r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
__device__ __forceinline__ float v2(float x) { const float y = __builtin_amdgcn_rsqf(x); float g = x * y, h = 0.5f * y; const float r = __builtin_fmaf(-h, g, 0.5f); g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h); const float d = __builtin_fmaf(-g, g, x); return __builtin_fmaf(d, h, g); }
__device__ __forceinline__ float v3(float x) { const float s = __builtin_amdgcn_sqrtf(x); const float h = 0.5f * __builtin_amdgcn_rsqf(x); const float r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
__device__ __forceinline__ float v4(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float o = rm <= 0.0f ? sm : s;
    o = rp > 0.0f ? sp : o;
    return o;
}
__device__ __forceinline__ float v5(float x) { const float y = __builtin_amdgcn_rsqf(x); const float s = x * y; const float h = 0.5f * y; const float r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }   // rsq + ONE residual step
__device__ __forceinline__ float v6(float x) { const float y = __builtin_amdgcn_rsqf(x); float s = x * y; float h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s); const float e = __builtin_fmaf(-h, s, 0.5f); h = __builtin_fmaf(h, e, h); r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }

__global__ void sweep(unsigned long long* bad, unsigned* expMask)
{
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b[6] = {0, 0, 0, 0, 0, 0};
    // per variant: bit e of word (e >> 5) set = some operand with biased exponent e mismatched
    for (unsigned k = 0; k < (unsigned)((1ull << 32) / stride); k++, i += stride) {
        if (!((i - 0x01000000u) < 0x7e000000u)) continue;
        const float x = __uint_as_float(i);
        volatile float vx = x;
        const unsigned ref = __float_as_uint(__builtin_sqrtf(vx));
        const unsigned e = i >> 23;
        const float c[6] = {v1(x), v2(x), v3(x), v4(x), v5(x), v6(x)};
        for (int k = 0; k < 6; k++) if (__float_as_uint(c[k]) != ref) { b[k]++; atomicOr(&expMask[k * 8 + (e >> 5)], 1u << (e & 31)); }
    }
    for (int k = 0; k < 6; k++) if (b[k]) atomicAdd(&bad[k], (unsigned long long)b[k]);
}

int main()
{
    unsigned long long* bad;
    hipMalloc(&bad, 64); hipMemset(bad, 0, 64);
    unsigned* em; hipMalloc(&em, 6 * 8 * 4); hipMemset(em, 0, 6 * 8 * 4);
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, bad, em);
    unsigned long long h[6];
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"operands\": \"positive floats, biased exponent 2..253\", \"mismatches\": {\"v1 rsq + two residual steps\": %llu, \"v2 rsq + coupled (g,h) step + residual\": %llu, "
           "\"v3 sqrt + residual with 0.5*rsq\": %llu, \"v4 sqrt + one-ulp neighbour test\": %llu, \"v5 rsq + one residual step\": %llu, \"v6 rsq + residual, refined h, residual\": %llu}}\n",
           h[0], h[1], h[2], h[3], h[4], h[5]);
    unsigned hm[48];
    hipMemcpy(hm, em, sizeof(hm), hipMemcpyDeviceToHost);
    for (int k = 0; k < 6; k++) {
        int lo = -1, hi = -1, n = 0;
        for (int e = 0; e < 256; e++) if (hm[k * 8 + (e >> 5)] >> (e & 31) & 1u) { if (lo < 0) lo = e; hi = e; n++; }
        printf("v%d: biased exponents with a mismatch: %d of them, from %d to %d\n", k + 1, n, lo, hi);
    }
    return 0;
}
