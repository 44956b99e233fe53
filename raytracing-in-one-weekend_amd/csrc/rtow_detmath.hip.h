// rtow_detmath.hip.h - deterministic fp32 sin/cos/log/exp2/pow for the gfx950 sample kernel.
//
// The reference calls math.sincos / math.log / math.pow (RT/RandomSource.cs:58,80,126, RT/Microfacet.cs:72,
// RT/Material.cs:80,216, JOBS/SampleBatchJob.cs:332); Burst's lowering of those is unpublished.  DESIGN.md
// section "Numeric specification" fixes one float program for each of them, built only from binary32
// mul / add / fma (v_fma_f32), v_rndne_f32 and bit operations, all exactly rounded on CDNA4, so the kernel's
// random decisions can be compared bit for bit with the CPU checker under tests/.  This file is the device
// implementation of that specification; it is compiled with -ffp-contract=off so that nothing outside the
// explicit __builtin_fmaf calls is fused.
#pragma once
#include <hip/hip_runtime.h>

namespace rtow {

__device__ __forceinline__ float bits_to_float(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned float_to_bits(float f) { return __float_as_uint(f); }

// sin(x), cos(x): quadrant reduction with a three-term pi/2 (Cody-Waite), degree-7 / degree-8 minimax on [-pi/4, pi/4].
__device__ __forceinline__ void det_sincos(float x, float& s, float& c)
{
    const float q = __builtin_rintf(x * 0.636619772367581343f);
    float r = __builtin_fmaf(q, -1.5703125f, x);
    r = __builtin_fmaf(q, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(q, -7.54978995489188216e-8f, r);
    const float r2 = r * r;

    float ps = __builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    ps = ps * r2;
    const float sin_r = __builtin_fmaf(ps, r, r);

    float pc = __builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    pc = pc * r2;
    const float cos_r = __builtin_fmaf(pc, r2, __builtin_fmaf(-0.5f, r2, 1.0f));

    const int quadrant = ((int)q) & 3;
    const bool swap = (quadrant & 1) != 0;
    const float s0 = swap ? cos_r : sin_r;
    const float c0 = swap ? sin_r : cos_r;
    // quadrant 0: ( s,  c)   1: ( c, -s)   2: (-s, -c)   3: (-c,  s)
    s = (quadrant & 2) ? -s0 : s0;
    c = (quadrant == 1 || quadrant == 2) ? -c0 : c0;
}

__device__ __forceinline__ float det_log(float x)
{
    if (x == 0.0f) return -__builtin_inff();
    if (!(x > 0.0f)) return __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    const unsigned u = float_to_bits(x);
    int e = (int)((u >> 23) & 0xffu) - 126;
    float m = bits_to_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float m2 = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    p = p * m;
    p = p * m2;
    const float ef = (float)e;
    p = __builtin_fmaf(-2.12194440e-4f, ef, p);
    p = __builtin_fmaf(-0.5f, m2, p);
    float out = m + p;
    out = __builtin_fmaf(0.693359375f, ef, out);
    return out;
}

__device__ __forceinline__ float det_exp2(float t)
{
    t = t > 126.0f ? 126.0f : t;
    t = t < -126.0f ? -126.0f : t;
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 1.535336188319500e-4f;
    p = __builtin_fmaf(p, f, 1.339887440266574e-3f);
    p = __builtin_fmaf(p, f, 9.618437357674640e-3f);
    p = __builtin_fmaf(p, f, 5.550332471162809e-2f);
    p = __builtin_fmaf(p, f, 2.402264791363012e-1f);
    p = __builtin_fmaf(p, f, 6.931472028550421e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return p * bits_to_float((unsigned)((int)n + 127) << 23);
}

// x^k for a small non-negative integer k: binary exponentiation, low bit first.
__device__ __forceinline__ float det_powi(float x, unsigned k)
{
    float acc = 1.0f, b = x;
    while (k) {
        if (k & 1u) acc = acc * b;
        k >>= 1;
        if (k) b = b * b;
    }
    return acc;
}

__device__ __forceinline__ float det_pow(float x, float y)
{
    if (y >= 0.0f && y <= 1024.0f && y == __builtin_rintf(y)) return det_powi(x, (unsigned)y);
    if (x < 0.0f || x != x) return __builtin_nanf("");
    if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
    return det_exp2(y * (det_log(x) * 1.44269504088896341f));
}

// the two fixed-exponent powers on the hot path
__device__ __forceinline__ float det_sq(float x) { return x * x; }                                   // pow(x, 2)
__device__ __forceinline__ float det_pow5(float x) { const float x2 = x * x; return x * (x2 * x2); } // pow(x, 5)

} // namespace rtow
