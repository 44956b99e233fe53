// rtow_exactmath.hip.h - correctly rounded 1/x and sqrt(x) in a handful of instructions.
//
// The path's float program uses IEEE division and square root, and hipcc's correctly rounded expansions of them cost
// about 15 and 20 full-rate-instruction equivalents each (profiles/calib: v_div_scale x 2, v_rcp, five FMAs, v_div_fmas, v_div_fixup;
// v_sqrt plus a scaled +-1 ulp search) - a third of the shading arithmetic.  For the two UNARY cases the hardware approximation
// (1 ulp) plus one fused residual step is already the correctly rounded result for every operand whose exponent is far from the ends of
// the range; everything else (zero, subnormal, huge, infinite, NaN, negative for sqrt) takes the compiler's expansion, so the function is
// the IEEE operation for all 2^32 inputs.  That is not argued, it is checked: tests/native/exactmath_parity.hip evaluates every one of
// the 2^32 operands on the device against `1.0f / x` and `__builtin_sqrtf(x)` (tests/test_gpu_detmath.py, every -m gpu run).
// Binary division: the operand space (2^64) cannot be enumerated, but the question "is one residual step on the correctly rounded
// reciprocal the IEEE quotient?" does not depend on the exponents while no intermediate leaves the normal range (every step scales exactly
// by powers of two), so it is enumerated over MANTISSAS: all 2^23 x 2^23 pairs, 34 s of an MI355X, no mismatch
// (tests/native/exactdiv_parity.hip).  exact_div3 uses it where three numerators share one divisor (a sphere's normal, RT/HitTests.cs:56);
// a single division gains nothing once the range guard is paid for and stays the compiler's.
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

// RN(a / b) given y = RN(1 / b): q = RN(a y), r = a - b q (exact, fused), RN(q + r y).  Valid while a, b, the quotient and the residual
// are normal; exact_div3 guards that.  This very function is what tests/native/exactdiv_parity.hip enumerates.
__device__ __forceinline__ float exact_div_step(float a, float b, float y)
{
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// (ax / b, ay / b, az / b), each correctly rounded.  Fast path: the numerators' biased exponents are in [64, 191] (magnitudes in
// [2^-63, 2^65)) and the divisor's in [96, 159] ([2^-31, 2^33)): then every quotient has an exponent in [-96, 96], the reciprocal one in
// [-33, 31] and the residuals are at least 2^-110 - all normal, which is what the enumeration over mantissas needs to carry over.
// Anything else (a zero component, infinities, NaNs, tiny or huge values) takes three IEEE divisions.  The numerators' range test:
// (bits << 1) + 0x40000000 has its top bit set exactly when the biased exponent is in [64, 191].
__device__ __forceinline__ void exact_div3(float ax, float ay, float az, float b, float& qx, float& qy, float& qz)
{
    const unsigned g = ((__float_as_uint(ax) << 1) + 0x40000000u) & ((__float_as_uint(ay) << 1) + 0x40000000u) & ((__float_as_uint(az) << 1) + 0x40000000u);
    const bool bOk = ((__float_as_uint(b) & 0x7f800000u) - 0x30000000u) < 0x20000000u;
    if (__builtin_expect((int)g < 0 && bOk, 1)) {
        const float y0 = __builtin_amdgcn_rcpf(b);                 // exact_rcp's fast path: b is well inside its range
        const float e = __builtin_fmaf(-b, y0, 1.0f);
        const float y = __builtin_fmaf(e, y0, y0);
        qx = exact_div_step(ax, b, y);
        qy = exact_div_step(ay, b, y);
        qz = exact_div_step(az, b, y);
    } else {
        qx = ax / b; qy = ay / b; qz = az / b;
    }
}

// Host overloads (the host pass of hipcc; rtow_probe.hip walks one ray on the CPU with the sample kernel's own hit tests): the IEEE operations themselves -
// which is what the device forms above are for every operand (enumerated, see the head of this file).
__host__ inline float exact_rcp(float x) { return 1.0f / x; }
__host__ inline float exact_rcp_nan_to_inf(float x) { const float r = 1.0f / x; return r != r ? __builtin_inff() : r; }
__host__ inline float exact_sqrt(float x) { return __builtin_sqrtf(x); }
__host__ inline void exact_div3(float ax, float ay, float az, float b, float& qx, float& qy, float& qz) { qx = ax / b; qy = ay / b; qz = az / b; }

} // namespace rtow
