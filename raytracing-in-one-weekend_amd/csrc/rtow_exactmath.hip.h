// rtow_exactmath.hip.h - correctly rounded 1/x and sqrt(x) in a handful of instructions.
//
// The path's float program uses IEEE division and square root, and hipcc's correctly rounded expansions of them cost
// about 15 and 20 full-rate-instruction equivalents each (profiles/calib: v_div_scale x 2, v_rcp, five FMAs, v_div_fmas, v_div_fixup;
// v_sqrt plus a scaled +-1 ulp search) - a third of the shading arithmetic.  For the two UNARY cases the hardware approximation
// (1 ulp) plus one fused residual step is already the correctly rounded result for every operand whose exponent is far from the ends of
// the range; everything else (zero, subnormal, huge, infinite, NaN, negative for sqrt) takes the compiler's expansion, so the function is
// the IEEE operation for all 2^32 inputs.  That is not argued, it is checked: tests/native/exactmath_parity.hip evaluates every one of
// the 2^32 operands on the device against `1.0f / x` and `__builtin_sqrtf(x)` (tests/test_gpu_detmath.py, every -m gpu run).
// Binary division (a / b) stays the compiler's: its operand space cannot be enumerated.
#pragma once
#include <hip/hip_runtime.h>

namespace rtow {

// RN(1 / x)
__device__ __forceinline__ float exact_rcp(float x)
{
    // biased exponent of x in [2, 252]: x and 1 / x are normal with room to spare, the residual below is exact
    if (__builtin_expect(((__float_as_uint(x) & 0x7f800000u) - 0x01000000u) <= 0x7d000000u, 1)) {
        const float y = __builtin_amdgcn_rcpf(x);                  // v_rcp_f32: 1 ulp
        const float e = __builtin_fmaf(-x, y, 1.0f);               // exact residual 1 - x * y (fused)
        return __builtin_fmaf(e, y, y);
    }
    return 1.0f / x;
}

// rcp(x) with NaN -> +INF, the reference's rayInvDirection (JOBS/SampleBatchJob.cs:406-412: `select(rcp, INFINITY, isnan(rcp))`).  The fast path
// cannot produce a NaN (finite, normal operand), so only the fallback looks for one.
__device__ __forceinline__ float exact_rcp_nan_to_inf(float x)
{
    if (__builtin_expect(((__float_as_uint(x) & 0x7f800000u) - 0x01000000u) <= 0x7d000000u, 1)) {
        const float y = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(e, y, y);
    }
    const float r = 1.0f / x;
    return r != r ? __builtin_inff() : r;
}

// RN(sqrt(x))
__device__ __forceinline__ float exact_sqrt(float x)
{
    // positive, biased exponent in [32, 253]; negative numbers, zeros, subnormals, infinities and NaNs fail the unsigned compare.  Below
    // 2^-102 the residual x - s * s underflows and the step below rounds wrongly (profiles/calib/sqrt_variants.hip: every mismatch of this
    // form has a biased exponent <= 24); the compiler's own expansion rescales from 2^-96 down, and so does this one by falling back to it
    if (__builtin_expect((__float_as_uint(x) - 0x10000000u) < 0x6f000000u, 1)) {
        const float y = __builtin_amdgcn_rsqf(x);                  // v_rsq_f32: 1 ulp
        const float s = x * y;
        const float h = 0.5f * y;
        const float r = __builtin_fmaf(-s, s, x);                  // exact residual x - s * s (fused)
        return __builtin_fmaf(r, h, s);
    }
    return __builtin_sqrtf(x);
}

} // namespace rtow
