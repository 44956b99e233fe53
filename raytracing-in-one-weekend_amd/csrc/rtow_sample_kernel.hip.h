// rtow_sample_kernel.hip.h - the sample-batch megakernel (hand-written gfx950 / CDNA4) and the device helpers shared with the small
// kernels of rtow_kernels.hip.  Included by one translation unit per scene kind (rtow_sample_*.hip): the kernel has 338
// template instantiations (240 + 90 with wide codes + 4 pinhole twins + 4 twins with the lanes in a hurry) and compiling them side by side keeps the build at a minute and a half.
//
// sample_batch_kernel replaces SampleBatchJob.Execute + Sample + FindHitCandidates + FindHits + Entity.Hit +
// Sphere.Hit + Material.Scatter + View.GetRay + RandomSource (JOBS/SampleBatchJob.cs:59-475, RT/*.cs).
//
// Shape of the kernel:
//  * persistent: one 1024-lane workgroup per CU; the whole scene image (BVH nodes, primitives, materials) is staged once into LDS with
//    coalesced 16-byte loads, behind an 8-entry candidate list per lane and a [level][lane] 16-bit traversal stack with one row per level of the
//    scene's own tree (LdsPlan, rtow_kernels.h: every LDS size is decided per launch);
//  * one lane = one PIXEL under the reference RNG policy (the reference seeds its generator once per pixel and runs it through all of
//    that pixel's samples, JOBS/SampleBatchJob.cs:91,132-157), one lane = one (pixel, 16-sample group) unit under RTOW_RNG_PER_SAMPLE;
//    pixel tickets are handed to waves in 64-pixel chunks, most expensive chunks first;
//  * per-lane state machine REGEN -> TRAV -> TEST -> (VOL ->) HIT | SKY and a wave-level stage scheduler: a trip of the main loop runs the
//    stages in pipeline order, each only if enough live lanes wait in it (ballot + popcount against thresholds); the box walk is
//    resumable and sliced; depth-0 rays take their candidates from a per-pixel list of leaf-parent nodes instead of walking;
//  * leaf boxes are the reference's own entity boxes with the reference's own slab test (a primitive is only tested when the ray passes
//    its box, and that guard is part of the result); inner boxes are padded unions; exact primitive tests are deferred to TEST;
//  * the per-depth emission/attenuation stacks (JOBS/SampleBatchJob.cs:103-104,311,330) are kept as 16-bit material codes packed in
//    VGPRs and re-expanded when the path is folded tail -> head (:384-396), which keeps the colour bit-identical to the reference's
//    fold order without 2 x TraceDepth float3 of per-lane storage (textured scenes keep per-depth colours in scratch instead);
//  * several successive batches of a frame can run in ONE launch (chained batches, rtowSampleBatchChainDevice): batch b of a 64-pixel chunk is
//    handed out once batch b - 1 of that chunk is stored (per-chunk counters, device-coherent accumulator accesses), so lanes that run out
//    of one batch's pixels take the next batch's instead of idling through the batch's tail;
//  * template parameters: ALL_LDS (scene fully LDS resident), KIND (spheres / moving spheres / general entities / volumes / textured /
//    both / triangles / textured triangles, + exact-tie bit), HW (history words: 4 / 8 = every code in registers, trace depth <= 8 / 16; 32 = eight codes in
//    registers, the rest in LDS rows, trace depth <= 64), DIAG (0 RayCount only / 1 the FULL_DIAGNOSTICS counters of this library's walk / 2 those or the
//    reference tree's), NOISE (white / blue / STBN), PER_SAMPLE, GEO (bit 2 wide codes, bit 3 pinhole twin, bit 4 lanes in a hurry); launchByDiagGeo (end of this file) says which
//    instantiation serves which batch.
//
// Numerics: compiled with -ffp-contract=off; every expression below is written in the evaluation order of the C#
// source (left to right, no fusion), with IEEE division and square root (1 / x and sqrt(x) through the exhaustively checked short forms
// of rtow_exactmath.hip.h), and the deterministic transcendental functions of rtow_detmath.hip.h.  No MFMA: there is no dense contraction anywhere on this path.
#pragma once
#include "rtow_kernels.h"

#include <type_traits>

#include "rtow_detmath.hip.h"
#include "rtow_exactmath.hip.h"

// IEEE 1 / x and sqrt(x) of the path's float program: the exhaustively checked short forms of rtow_exactmath.hip.h (same result for every
// operand; RTOW_EXACT_MATH=0 builds the compiler's expansions instead, for A/B timing).
#ifndef RTOW_EXACT_MATH
#define RTOW_EXACT_MATH 1
#endif
#ifndef RTOW_SPLIT_NODE_LOADS
#define RTOW_SPLIT_NODE_LOADS 1   // 0: A/B build with one flat load per node quad (the base chosen per lane) in the kernels whose tree does not fit LDS
#endif
#ifndef RTOW_WHOLE_MATERIAL
#define RTOW_WHOLE_MATERIAL 1     // 0: A/B build in which the kernels beyond LDS load a hit's material record piece by piece where it is used (HIT)
#endif
#ifndef RTOW_LDS_VIEW
#define RTOW_LDS_VIEW 2           // bit 0 / bit 1 = the kernels with / without the scene in LDS read the view's and the sky's launch constants from an LDS copy (see REGEN)
#endif
#ifndef RTOW_COLD_VIEW
#define RTOW_COLD_VIEW 1          // 0: A/B build in which every kernel holds the view's and the sky's launch constants in scalar registers through every stage
#endif
#ifndef RTOW_TRI_HOT
#define RTOW_TRI_HOT 1        // 0: A/B build in which the all-triangle kinds test and shade from the 128-byte GpuPrim records (the compact GpuTriHot / GpuTriCold records are still uploaded)
#endif
#ifndef RTOW_PREFETCH
#define RTOW_PREFETCH 0       // A/B builds, wide-code kernels: bit 0 = the far child's node is requested when it is pushed, bit 1 = a triangle's compact record when it is listed.
                              // Measured on the 250 882-triangle mesh, same box (profiles/r06a_mesh_layout_prefetch_watch.json): 2 178 Msamples/s without, 2 090 with bit 0, 1 990 with
                              // bit 1, 1 910 with both - every extra vector-memory request costs: the kernel is bound by its request path (one request per lane and node quad), not by latency alone
#endif
#ifndef RTOW_PINHOLE
#define RTOW_PINHOLE 1        // 0: A/B build without the pinhole twins (kGeoPinhole)
#endif
#ifndef RTOW_TIE_WATCH
#define RTOW_TIE_WATCH 1      // 0: A/B build without the nearest-hit tie watch of the sphere kinds
#endif
#if RTOW_EXACT_MATH
#define RTOW_RCP(x) rtow::exact_rcp(x)
#define RTOW_RCP_NAN_TO_INF(x) rtow::exact_rcp_nan_to_inf(x)
#define RTOW_SQRT(x) rtow::exact_sqrt(x)
#ifndef RTOW_EXACT_DIV3
#define RTOW_EXACT_DIV3 1
#endif
#else
#define RTOW_RCP(x) (1.0f / (x))
#define RTOW_RCP_NAN_TO_INF(x) ([](float r_) { return r_ != r_ ? __builtin_inff() : r_; }(1.0f / (x)))
#define RTOW_SQRT(x) __builtin_sqrtf(x)
#define RTOW_EXACT_DIV3 0
#endif

namespace rtow {

namespace {

// ------------------------------------------------------------------------------------------------------------
// small float3 helpers; each spells out the reference's evaluation order
// ------------------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
// (the helpers marked __host__ __device__ below - vectors, um_min / um_max, scene access, sphere_at, sphere_hit, general_hit - are also what rtowProbeNearestHit walks its one
// ray with on the host: rtow_probe.hip; the host pass evaluates the same expressions with the IEEE operations the device's short forms stand for)

__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ V3 v3(const RtowFloat3& a) { return v3(a.x, a.y, a.z); }
__host__ __device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__host__ __device__ __forceinline__ V3 scale(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__host__ __device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// math.normalize(v) = rsqrt(dot(v, v)) * v with rsqrt(x) = 1 / sqrt(x)
__device__ __forceinline__ V3 normalize(V3 v) { const float r = RTOW_RCP(RTOW_SQRT(dot(v, v))); return scale(r, v); }
// math.reflect(i, n) = i - 2f * n * dot(i, n)
__device__ __forceinline__ V3 reflect(V3 i, V3 n)
{
    const float d = dot(i, n);
    return v3(i.x - (2.0f * n.x) * d, i.y - (2.0f * n.y) * d, i.z - (2.0f * n.z) * d);
}
// math.min / math.max return the FIRST operand when the second is NaN
__host__ __device__ __forceinline__ float um_min(float x, float y) { return (y != y || x < y) ? x : y; }
__host__ __device__ __forceinline__ float um_max(float x, float y) { return (y != y || x > y) ? x : y; }
__device__ __forceinline__ float um_saturate(float x) { return um_max(0.0f, um_min(1.0f, x)); }

constexpr float kPi = 3.14159265f; // math.PI

// x / pow(2, depth) == x * 2^-depth exactly (scaling by a power of two), including the subnormal end of the range;
// pow(2, depth) overflows to +inf from depth 128 on, where the quotient is 0.
__device__ __forceinline__ float inv_pow2(int depth)
{
    if (depth <= 126) return __uint_as_float((unsigned)(127 - depth) << 23);
    if (depth == 127) return __uint_as_float(0x00400000u);
    return 0.0f;
}

// Unity.Mathematics.Random: NextState returns the pre-update state; NextFloat = asfloat(0x3f800000 | (s >> 9)) - 1
__device__ __forceinline__ float rng_next(unsigned& state)
{
    const unsigned t = state;
    unsigned s = t;
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    state = s;
    return __uint_as_float(0x3f800000u | (t >> 9)) - 1.0f;
}

// Unity.Mathematics.half -> float (exact)
__device__ __forceinline__ float half_bits_to_float(unsigned h)
{
    const unsigned sign = (h & 0x8000u) << 16;
    const unsigned exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0) return __uint_as_float(__float_as_uint((float)man * 5.9604644775390625e-8f) | sign);   // zero / subnormal: man * 2^-24, exact
    if (exp == 31) return __uint_as_float(sign | 0x7f800000u | (man << 13));
    return __uint_as_float(sign | ((exp + 112u) << 23) | (man << 13));
}
// Tools.TangentToWorldSpace (UTIL/Tools.cs:19-37): corrected Frisvad / Pixar basis, float3x3(tangent, normal, bitangent) * v, normalized
__device__ __forceinline__ V3 tangent_to_world(float tx, float ty, float tz, V3 n)
{
    const float s = n.z >= 0 ? 1.0f : -1.0f;
    const float a = -RTOW_RCP(s + n.z);      // -1 / x == -(1 / x): rounding to nearest is symmetric
    const float b = n.x * n.y * a;
    const V3 tangent = v3(1 + s * n.x * n.x * a, s * b, -s * n.x);
    const V3 bitangent = v3(b, s + n.y * n.y * a, -n.y);
    const V3 r = v3(tangent.x * tx + n.x * ty + bitangent.x * tz,
                    tangent.y * tx + n.y * ty + bitangent.y * tz,
                    tangent.z * tx + n.z * ty + bitangent.z * tz);
    return normalize(r);
}
// RandomSource.OnCosineWeightedHemisphere (RT/RandomSource.cs:63-89) from its two uniform numbers
__device__ __forceinline__ V3 cosine_hemisphere_uv(float u, float v, V3 n)
{
    const float radius = RTOW_SQRT(u);
    const float theta = v * 2 * kPi;
    float sinT, cosT;
    det_sincos(theta, sinT, cosT);
    const float tx = radius * cosT, tz = radius * sinT;
    const float ty = RTOW_SQRT(1 - u);
    return tangent_to_world(tx, ty, tz, n);
}
// RandomSource.NextFloat3Direction (RT/RandomSource.cs:113-128) from its two uniform numbers
__device__ __forceinline__ V3 direction_uv(float r0, float r1)
{
    const float z = r0 * 2.0f - 1.0f;
    const float rr = RTOW_SQRT(um_max(1.0f - z * z, 0.0f));
    const float angle = r1 * kPi * 2.0f;
    float sn, cs;
    det_sincos(angle, sn, cs);
    return v3(cs * rr, sn * rr, z);
}

// ------------------------------------------------------------------------------------------------------------
// RandomSource (RT/RandomSource.cs:15-150), one specialisation per NoiseColor.
//   White              Unity.Mathematics.Random, seeded per pixel (JOBS/SampleBatchJob.cs:91)
//   Blue               RT/BlueNoise.cs: one half4 texture walked by PerPixelNoise (RT/PerPixelNoise.cs) - the R2 sequence (RT/R2.cs) gives
//                      the same offsets to every pixel, the pixel's coordinates shift them; NextFloat2 is .xy of ONE texel
//   SpatioTemporalBlue RT/SpatioTemporalBlueNoise.cs: five byte textures (scalar, vector2, cosine-weighted unit vector3, unit vector2,
//                      unit vector3), each with its own PerPixelNoise walk
// PerPixelNoise keeps (offset, n) with offset = floor(R2(n - 1) * rowStride): a function of n, so n alone is the state.
// ------------------------------------------------------------------------------------------------------------
struct NoiseSite { const SampleKernelArgs* A; unsigned cx, cy; };   // where the texture-driven sources read: the batch's textures, the pixel

// index of the texel PerPixelNoise.Next() reads for counter value n (n = seed + 1 + number of earlier Next() calls), RT/PerPixelNoise.cs:27-38
__device__ __forceinline__ unsigned noise_texel(unsigned n, unsigned rowStride, unsigned cx, unsigned cy)
{
    constexpr float g = 1.32471795724474602596f;        // RT/R2.cs:11-13, folded one binary32 operation at a time
    constexpr float a1 = 1.0f / g;
    constexpr float a2 = 1.0f / (g * g);
    const float fn = (float)(n - 1u);
    const float x = 0.5f + a1 * fn, y = 0.5f + a2 * fn;
    const float rx = x - __builtin_floorf(x), ry = y - __builtin_floorf(y);          // % 1 of a non-negative float: exact
    const unsigned ox = (unsigned)__builtin_floorf(rx * (float)rowStride), oy = (unsigned)__builtin_floorf(ry * (float)rowStride);
    return ((cy + oy) % rowStride) * rowStride + ((cx + ox) % rowStride);
}

template <int NOISE>
struct Rng;

template <>
struct Rng<RTOW_NOISE_WHITE> {
    unsigned s;
    __device__ __forceinline__ void begin_pixel(const NoiseSite&, unsigned pix, unsigned seed)
    {
        s = (seed * 0x8C4CA03Fu) ^ (pix * 0x7383ED49u);   // :91; the Random ctor then discards one NextState()
        (void)rng_next(s);
    }
    // RTOW_RNG_PER_SAMPLE: sample `smp` of the pixel gets its own generator (include/rtow.h)
    __device__ __forceinline__ void begin_sample(const NoiseSite& at, unsigned pix, unsigned smp)
    {
        s = ((at.A->seed * 0x8C4CA03Fu) ^ (pix * 0x7383ED49u)) ^ ((smp + 1u) * 0x9E3779B9u);
        if (s == 0u) s = 0x9E3779B9u;                            // Random needs a non-zero state
        (void)rng_next(s);
    }
    __device__ __forceinline__ float next(const NoiseSite&) { return rng_next(s); }
    __device__ __forceinline__ void next2(const NoiseSite&, float& a, float& b) { a = rng_next(s); b = rng_next(s); }
    __device__ __forceinline__ void in_unit_disk(const NoiseSite&, float& x, float& y)       // RT/RandomSource.cs:40-61
    {
        const float theta = rng_next(s) * (2.0f * kPi - 0.0f) + 0.0f;                        // NextFloat(0, 2 * PI)
        const float radius = RTOW_SQRT(rng_next(s));
        float sinT, cosT;
        det_sincos(theta, sinT, cosT);
        x = radius * cosT; y = radius * sinT;
    }
    __device__ __forceinline__ V3 cosine_hemisphere(const NoiseSite&, V3 n) { const float u = rng_next(s), v = rng_next(s); return cosine_hemisphere_uv(u, v, n); }
    __device__ __forceinline__ void skip_cosine_hemisphere(const NoiseSite&) { (void)rng_next(s); (void)rng_next(s); }
    __device__ __forceinline__ V3 direction(const NoiseSite&) { const float r0 = rng_next(s), r1 = rng_next(s); return direction_uv(r0, r1); }
    __device__ __forceinline__ unsigned trace_value() const { return s; }
};

// The per-sample policies (include/rtow.h RtowRngPolicy; not the reference's stream): every sample has its own white-noise generator -
// Unity's xorshift32 reseeded per sample (RTOW_RNG_PER_SAMPLE) or xoroshiro64** (RTOW_RNG_PER_SAMPLE_XOROSHIRO, the generator north_star names:
// output rotl(s0 * 0x9E3779BB, 5) * 5; s1 ^= s0; s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9); s1 = rotl(s1, 13)).  One kernel variant serves both: the
// choice is a launch constant, and a second state word costs these variants nothing they notice.
struct RngPerSample {
    unsigned s, s1;
    __device__ __forceinline__ static unsigned rotl(unsigned x, unsigned k) { return __builtin_amdgcn_alignbit(x, x, 32u - k); }
    __device__ __forceinline__ float draw(const NoiseSite& at)
    {
        if (at.A->xoroshiro) {
            const unsigned s0 = s;
            unsigned t1 = s1;
            const unsigned result = rotl(s0 * 0x9E3779BBu, 5u) * 5u;
            t1 ^= s0;
            s = rotl(s0, 26u) ^ t1 ^ (t1 << 9);
            s1 = rotl(t1, 13u);
            return __uint_as_float(0x3f800000u | (result >> 9)) - 1.0f;
        }
        return rng_next(s);
    }
    __device__ __forceinline__ void begin_pixel(const NoiseSite&, unsigned, unsigned) { s = 1u; s1 = 0u; }     // every sample reseeds (begin_sample)
    __device__ __forceinline__ void begin_sample(const NoiseSite& at, unsigned pix, unsigned smp)
    {
        s = ((at.A->seed * 0x8C4CA03Fu) ^ (pix * 0x7383ED49u)) ^ ((smp + 1u) * 0x9E3779B9u);
        if (at.A->xoroshiro) {
            s1 = (s * 0x85EBCA6Bu) ^ 0xC2B2AE35u;
            if ((s | s1) == 0u) s1 = 0x9E3779B9u;
        } else if (s == 0u) {
            s = 0x9E3779B9u;                                      // Random needs a non-zero state
        }
        (void)draw(at);                                           // like the Random ctor: one output discarded
    }
    __device__ __forceinline__ float next(const NoiseSite& at) { return draw(at); }
    __device__ __forceinline__ void next2(const NoiseSite& at, float& a, float& b) { a = draw(at); b = draw(at); }
    __device__ __forceinline__ void in_unit_disk(const NoiseSite& at, float& x, float& y)
    {
        const float theta = draw(at) * (2.0f * kPi - 0.0f) + 0.0f;
        const float radius = RTOW_SQRT(draw(at));
        float sinT, cosT;
        det_sincos(theta, sinT, cosT);
        x = radius * cosT; y = radius * sinT;
    }
    __device__ __forceinline__ V3 cosine_hemisphere(const NoiseSite& at, V3 n) { const float u = draw(at), v = draw(at); return cosine_hemisphere_uv(u, v, n); }
    __device__ __forceinline__ void skip_cosine_hemisphere(const NoiseSite& at) { (void)draw(at); (void)draw(at); }
    __device__ __forceinline__ V3 direction(const NoiseSite& at) { const float r0 = draw(at), r1 = draw(at); return direction_uv(r0, r1); }
    __device__ __forceinline__ unsigned trace_value() const { return s; }
};

template <>
struct Rng<RTOW_NOISE_BLUE> {
    unsigned s;                                                   // PerPixelNoise.n
    __device__ __forceinline__ void begin_pixel(const NoiseSite&, unsigned, unsigned seed) { s = seed + 1u; }   // n = seed; Advance() (:17-25)
    __device__ __forceinline__ void begin_sample(const NoiseSite&, unsigned, unsigned) {}
    __device__ __forceinline__ void texel(const NoiseSite& at, float& x, float& y)
    {
        const uint2 t = reinterpret_cast<const uint2*>(at.A->blueNoise)[noise_texel(s, at.A->blueRowStride, at.cx, at.cy)];   // half4
        s++;
        x = half_bits_to_float(t.x & 0xffffu); y = half_bits_to_float(t.x >> 16);
    }
    __device__ __forceinline__ float next(const NoiseSite& at) { float x, y; texel(at, x, y); return x; }              // BlueNoise.cs:26
    __device__ __forceinline__ void next2(const NoiseSite& at, float& a, float& b) { texel(at, a, b); }                  // BlueNoise.cs:28
    __device__ __forceinline__ void in_unit_disk(const NoiseSite& at, float& x, float& y)
    {
        const float theta = next(at) * 2 * kPi;
        const float radius = RTOW_SQRT(next(at));
        float sinT, cosT;
        det_sincos(theta, sinT, cosT);
        x = radius * cosT; y = radius * sinT;
    }
    __device__ __forceinline__ V3 cosine_hemisphere(const NoiseSite& at, V3 n) { float u, v; texel(at, u, v); return cosine_hemisphere_uv(u, v, n); }
    __device__ __forceinline__ void skip_cosine_hemisphere(const NoiseSite&) { s++; }
    __device__ __forceinline__ V3 direction(const NoiseSite& at) { float r0, r1; texel(at, r0, r1); return direction_uv(r0, r1); }
    __device__ __forceinline__ unsigned trace_value() const { return s; }
};

template <>
struct Rng<RTOW_NOISE_SPATIOTEMPORAL_BLUE> {
    unsigned s, v2, cs, u2, u3;                                   // n of perPixelScalar / Vector2 / CosineUnitVector3 / UnitVector2 / UnitVector3
    __device__ __forceinline__ void begin_pixel(const NoiseSite&, unsigned, unsigned seed) { s = v2 = cs = u2 = u3 = seed + 1u; }
    __device__ __forceinline__ void begin_sample(const NoiseSite&, unsigned, unsigned) {}
    __device__ __forceinline__ float next(const NoiseSite& at)                                                           // STBN :61
    {
        const unsigned i = noise_texel(s++, at.A->stbRowStride, at.cx, at.cy);
        return (float)at.A->stbScalar[i] / 256.0f;
    }
    __device__ __forceinline__ void next2(const NoiseSite& at, float& a, float& b)                                       // :63-67
    {
        const uint8_t* t = at.A->stbVector2 + (size_t)noise_texel(v2++, at.A->stbRowStride, at.cx, at.cy) * 3u;
        a = (float)t[0] / 256.0f; b = (float)t[1] / 256.0f;
    }
    __device__ __forceinline__ void in_unit_disk(const NoiseSite& at, float& x, float& y)                                // NextUnitVector2, :75-79
    {
        const uint8_t* t = at.A->stbUnitVector2 + (size_t)noise_texel(u2++, at.A->stbRowStride, at.cx, at.cy) * 3u;
        x = (float)t[0] / 256.0f * 2 - 1; y = (float)t[1] / 256.0f * 2 - 1;
    }
    __device__ __forceinline__ V3 cosine_hemisphere(const NoiseSite& at, V3 n)                                           // NextCosineUnitVector3, :69-73: (r, b, g)
    {
        const uint8_t* t = at.A->stbCosineUnitVector3 + (size_t)noise_texel(cs++, at.A->stbRowStride, at.cx, at.cy) * 4u;
        return tangent_to_world((float)t[0] / 256.0f * 2 - 1, (float)t[2] / 256.0f * 2 - 1, (float)t[1] / 256.0f * 2 - 1, n);
    }
    __device__ __forceinline__ void skip_cosine_hemisphere(const NoiseSite&) { cs++; }
    __device__ __forceinline__ V3 direction(const NoiseSite& at)                                                         // NextUnitVector3, :81-85
    {
        const uint8_t* t = at.A->stbUnitVector3 + (size_t)noise_texel(u3++, at.A->stbRowStride, at.cx, at.cy) * 3u;
        return v3((float)t[0] / 256.0f * 2 - 1, (float)t[1] / 256.0f * 2 - 1, (float)t[2] / 256.0f * 2 - 1);
    }
    __device__ __forceinline__ unsigned trace_value() const { return s; }
};

// Microfacet.TrowbridgeReitz.RoughnessToAlpha / Lambda, SmithMaskingShadowing (RT/Microfacet.cs:9-12,53-80)
__device__ __forceinline__ float roughness_to_alpha(float roughness)
{
    roughness = um_max(roughness, 1e-3f);
    const float x = det_log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
__device__ __forceinline__ float smith_g1(V3 w, V3 n, float alpha /* = RoughnessToAlpha(roughness), per material */)
{
    const float cosTheta = dot(n, w);
    const float sqCos = cosTheta * cosTheta;
    const float sqSin = um_max(0.0f, 1 - sqCos);
    const float sinTheta = RTOW_SQRT(sqSin);
    const float tanTheta = sinTheta / cosTheta;
    const float absTan = __builtin_fabsf(tanTheta);
    float lambda;
    if (__builtin_isinf(absTan)) {
        lambda = 0;
    } else {
        const float a2t2 = (alpha * absTan) * (alpha * absTan);
        lambda = (-1 + RTOW_SQRT(1 + a2t2)) / 2;
    }
    return RTOW_RCP(1 + lambda);
}

// ------------------------------------------------------------------------------------------------------------
// Cubemap.Sample (RT/Texture.cs:171-210): the face is the first axis whose |component| is the largest (x before y before z), the
// texel min((int2)((uv + 1) * halfFaceSize), faceSizeMinusOne) of that face, point sampled; RGBA half or byte channels.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 cubemap_sample(const SampleKernelArgs& hotArgs, V3 d)
{
    // the cubemap's nine launch constants are read where they are used - through a laundered pointer to the kernarg segment, like the pixel boundary's (REGEN) - instead of
    // living in scalar registers (or, spilled, in VGPR lanes and scratch) through every stage of every launch, most of which have a gradient sky
#if defined(__HIP_DEVICE_COMPILE__)
    const SampleKernelArgs* coldArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();   // the struct is the kernel's only argument
    asm volatile("" : "+s"(coldArgs));
    const SampleKernelArgs& A = *coldArgs;
#else
    const SampleKernelArgs& A = hotArgs;                                                                   // host pass of the HIP compiler: never executed
#endif
    (void)hotArgs;
    if (!A.cubemapData) return v3(0, 0, 0);
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const float m = um_max(um_max(um_max(ax, ay), az), 0.0f);                  // cmax(float4(abs(vector), 0))
    int lane;
    if (m == ax) lane = 0; else if (m == ay) lane = 1; else if (m == az) lane = 2; else return v3(0, 0, 0);   // NaN direction
    const float major = lane == 0 ? d.x : lane == 1 ? d.y : d.z;
    const float amajor = lane == 0 ? ax : lane == 1 ? ay : az;
    const bool positive = major >= 0;
    float u, v;
    if (lane == 0) { u = positive ? -d.z : d.z; v = -d.y; }
    else if (lane == 1) { u = d.x; v = positive ? d.z : -d.z; }
    else { u = positive ? d.x : -d.x; v = -d.y; }
    u = u / amajor;
    v = v / amajor;
    int cx = (int)((u + 1) * (float)A.cubemapHalfW), cy = (int)((v + 1) * (float)A.cubemapHalfH);
    cx = cx < A.cubemapW1 ? cx : A.cubemapW1;
    cy = cy < A.cubemapH1 ? cy : A.cubemapH1;
    const uint8_t* px = A.cubemapData + (size_t)(lane * 2 + (positive ? 0 : 1)) * (size_t)A.cubemapFaceStride + cx * A.cubemapPixelStride + cy * A.cubemapRowStride;
    if (A.cubemapChannelType == RTOW_CUBEMAP_UNSIGNED_BYTE) return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    const unsigned short* hp = reinterpret_cast<const unsigned short*>(px);
    return v3(half_bits_to_float(hp[0]), half_bits_to_float(hp[1]), half_bits_to_float(hp[2]));
}

// ------------------------------------------------------------------------------------------------------------
// Texture.SampleColor / SampleScalar (RT/Texture.cs:51-138) for the per-hit evaluation of textured materials.  Image: the texel
// (int2)(uv * ImageSize) - clamped into the image, where the reference would read out of bounds - as bytes / 255 * MainColor.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const uint8_t* texture_pixel(const SampleKernelArgs& A, const GpuTexture& t, float2 uv)
{
    const GpuImage im = reinterpret_cast<const GpuImage*>(A.texBlob + A.texLayout.imageOffset)[t.image];
    int x = (int)(uv.x * (float)im.width), y = (int)(uv.y * (float)im.height);
    x = x < 0 ? 0 : x > im.width - 1 ? im.width - 1 : x;
    y = y < 0 ? 0 : y > im.height - 1 ? im.height - 1 : y;
    return A.texBlob + A.texLayout.pixelOffset + im.offset + ((size_t)y * (size_t)im.width + (size_t)x) * (size_t)im.pixelStride;
}
__device__ __forceinline__ V3 texture_color(const SampleKernelArgs& A, const GpuTexture& t, float2 uv)
{
    if (t.type == RTOW_TEXTURE_CONSTANT) return v3(t.mainColor[0], t.mainColor[1], t.mainColor[2]);
    if (t.type == RTOW_TEXTURE_CONSTANT_SCALAR) return v3(t.parameter, t.parameter, t.parameter);
    if (t.type == RTOW_TEXTURE_IMAGE && t.image >= 0) {
        const uint8_t* px = texture_pixel(A, t, uv);
        return v3((float)px[0] / 255.0f * t.mainColor[0], (float)px[1] / 255.0f * t.mainColor[1], (float)px[2] / 255.0f * t.mainColor[2]);
    }
    return v3(0, 0, 0);
}
__device__ __forceinline__ float texture_scalar(const SampleKernelArgs& A, const GpuTexture& t, float2 uv)
{
    const float main = t.channel == 0 ? t.mainColor[0] : t.channel == 1 ? t.mainColor[1] : t.mainColor[2];
    if (t.type == RTOW_TEXTURE_CONSTANT) return main;
    if (t.type == RTOW_TEXTURE_CONSTANT_SCALAR) return t.parameter;
    if (t.type == RTOW_TEXTURE_IMAGE && t.image >= 0) return (float)texture_pixel(A, t, uv)[t.channel] / 255.0f * main;
    return 0.0f;
}

// [hit-list sorts: begin]  (tests/test_hitsort_host.py compiles the text between the two markers for the host and checks it against the oracle)
// ------------------------------------------------------------------------------------------------------------
// hitBuffer.Sort(DistanceComparer) (JOBS/SampleBatchJob.cs:473-474) for the hit list of a volume scene.
// The reference sorts a list that starts in its tree's leaf order with a sort that is not stable, and hits at bit-identical
// distances (coplanar faces) keep whatever order that leaves: put the hits in leaf order first (rank, rtow_reforder.h), then run
// the same sort (NativeSortExtension: compare-exchange for 2 and 3 elements, insertion up to 16, above that median-of-three Hoare
// partitions down to ranges of <= 16; the heap-sort fallback of the introsort is out of reach for <= 24 elements).
// A real call on purpose (the lists live in scratch anyway): it keeps this rarely run code and its temporaries out of the
// register allocation of the stage loop.
// ------------------------------------------------------------------------------------------------------------
constexpr unsigned kHitPrimMask = 0x3fffffffu;   // hit code = primitive index | bit 30: dot(normal, dir) < 0 | bit 31: dot(normal, dir) > 0
__device__ __noinline__ __attribute__((unused)) void sort_hit_list(float* hitT, float* hitTmin0, unsigned* hitCode, int nHits, const unsigned* rank)
{
    auto rankOf = [&](unsigned code) { return rank[code & kHitPrimMask]; };
    auto swapHits = [&](int a, int b) {
        const float t = hitT[a], tm = hitTmin0[a]; const unsigned c = hitCode[a];
        hitT[a] = hitT[b]; hitTmin0[a] = hitTmin0[b]; hitCode[a] = hitCode[b];
        hitT[b] = t; hitTmin0[b] = tm; hitCode[b] = c;
    };
    auto swapIfGreater = [&](int l, int r) { if (l != r && hitT[l] > hitT[r]) swapHits(l, r); };
    for (int i = 1; i < nHits; i++) {                                   // leaf order; an entity's exit hit was recorded after its entry
        const float t = hitT[i], tm = hitTmin0[i];
        const unsigned c = hitCode[i], r = rankOf(c);
        int j = i;                                                       // the hole: no index is ever formed from a negative value (see hit_spill_entry)
        while (j > 0 && rankOf(hitCode[j - 1]) > r) { hitT[j] = hitT[j - 1]; hitTmin0[j] = hitTmin0[j - 1]; hitCode[j] = hitCode[j - 1]; j--; }
        hitT[j] = t; hitTmin0[j] = tm; hitCode[j] = c;
    }
    // pending ranges of the introsort; a list of <= 24 hits needs at most four partition steps before every range is <= 16
    int rangeLo[8], rangeHi[8], ranges = 1;
    rangeLo[0] = 0; rangeHi[0] = nHits - 1;
    while (ranges > 0) {
        ranges--;
        int lo = rangeLo[ranges], hi = rangeHi[ranges];
        while (hi > lo) {
            const int size = hi - lo + 1;
            if (size == 2) { swapIfGreater(lo, hi); break; }
            if (size == 3) { swapIfGreater(lo, hi - 1); swapIfGreater(lo, hi); swapIfGreater(hi - 1, hi); break; }
            if (size <= 16) {
                for (int i = lo + 1; i <= hi; i++) {
                    const float t = hitT[i], tm = hitTmin0[i];
                    const unsigned c = hitCode[i];
                    int j = i;
                    while (j > lo && t < hitT[j - 1]) { hitT[j] = hitT[j - 1]; hitTmin0[j] = hitTmin0[j - 1]; hitCode[j] = hitCode[j - 1]; j--; }
                    hitT[j] = t; hitTmin0[j] = tm; hitCode[j] = c;
                }
                break;
            }
            // median-of-three Hoare partition; the right part is sorted on its own, the left part continues here
            const int mid = lo + (hi - lo) / 2;
            swapIfGreater(lo, mid); swapIfGreater(lo, hi); swapIfGreater(mid, hi);
            const float pivot = hitT[mid];
            swapHits(mid, hi - 1);
            int left = lo, right = hi - 1;
            while (left < right) {
                while (pivot > hitT[++left]) {}
                while (pivot < hitT[--right]) {}
                if (left >= right) break;
                swapHits(left, right);
            }
            swapHits(left, hi - 1);
            rangeLo[ranges] = left + 1; rangeHi[ranges] = hi; ranges++;
            hi = left - 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Hit lists longer than a lane holds itself.  The reference's hitRecordBuffer is a HybridList that grows on the heap
// (UTIL/HybridCollections.cs:65-71), so a ray may meet any number of surfaces; here entries [0, kLocalHits) stay in the lane's own
// arrays (scratch) and entry e >= kLocalHits lives in the lane's column of a spill area in HBM (SampleKernelArgs.hitSpill,
// [entry][lane] so that a wave's accesses coalesce).  Only lists that long pay for it.
// ------------------------------------------------------------------------------------------------------------
constexpr int kLocalHits = kLocalHitEntries;

struct HitSpill {
    uint4* column;       // this lane's entry kLocalHits; null when the context holds no spill area
    uint32_t stride;     // uint4s between successive entries of one lane
    uint32_t entries;    // entries per lane beyond kLocalHits
};

__device__ __forceinline__ HitSpill hit_spill_of(const SampleKernelArgs& A)
{
    return HitSpill{A.hitSpill ? A.hitSpill + (size_t)blockIdx.x * kBlockThreads + threadIdx.x : nullptr, A.hitSpillStride, A.hitSpill ? A.hitSpillEntries : 0u};
}

struct HitRec { float t, tmin0; unsigned code; };

// Index arithmetic of the sorts below: no index is ever formed by adding a constant to a value that can be negative.  The natural
// insertion loop (`j = i - 1; while (j >= lo && ...) { a[j + 1] = a[j]; j--; } a[j + 1] = x;`) reaches j = -1, and hipcc 7.2 at -O2 and
// above turned `a[j + 1]` into zext(j) * 4 + 4 (separate-const-offset-from-gep, pass 2598 of the device compile by opt-bisect; -O1 and
// a host build of the same source are correct): an address 16 GB past the array, outside every aperture, whenever an element moved to
// the front of its range (tests/test_gpu_hitsort.py).  The loops therefore track the hole (j >= lo >= 0) instead, and the spill entry
// index is formed in unsigned arithmetic.
__device__ __forceinline__ uint4* hit_spill_entry(const HitSpill& sp, int i)
{
    const unsigned e = (unsigned)i - (unsigned)kLocalHits;
    return sp.column + (size_t)e * (size_t)sp.stride;
}

__device__ __forceinline__ HitRec hit_get(const float* hitT, const float* hitTmin0, const unsigned* hitCode, const HitSpill& sp, int i)
{
    if (i < kLocalHits) return HitRec{hitT[i], hitTmin0[i], hitCode[i]};
    const uint4 v = *hit_spill_entry(sp, i);
    return HitRec{__uint_as_float(v.x), __uint_as_float(v.y), v.z};
}

__device__ __forceinline__ void hit_set(float* hitT, float* hitTmin0, unsigned* hitCode, const HitSpill& sp, int i, HitRec r)
{
    if (i < kLocalHits) { hitT[i] = r.t; hitTmin0[i] = r.tmin0; hitCode[i] = r.code; }
    else *hit_spill_entry(sp, i) = make_uint4(__float_as_uint(r.t), __float_as_uint(r.tmin0), r.code, 0u);
}

// sort_hit_list for a list of any length: leaf order first, then the whole NativeSortExtension introsort - the partition steps as above,
// plus the heap sort it falls back to once 2 * floor(log2(n)) partition levels are used up (within reach from 25 elements on).
__device__ __noinline__ __attribute__((unused)) void sort_hit_list_spilled(float* hitT, float* hitTmin0, unsigned* hitCode, HitSpill sp, int nHits, const unsigned* rank)
{
    auto get = [&](int i) { return hit_get(hitT, hitTmin0, hitCode, sp, i); };
    auto set = [&](int i, HitRec r) { hit_set(hitT, hitTmin0, hitCode, sp, i, r); };
    auto rankOf = [&](unsigned code) { return rank[code & kHitPrimMask]; };
    auto swapHits = [&](int a, int b) { const HitRec x = get(a), y = get(b); set(a, y); set(b, x); };
    auto swapIfGreater = [&](int l, int r) { if (l != r && get(l).t > get(r).t) swapHits(l, r); };
    for (int i = 1; i < nHits; i++) {                                   // leaf order; an entity's exit hit was recorded after its entry
        const HitRec h = get(i);
        const unsigned r = rankOf(h.code);
        int j = i;                                                       // the hole; see the note on index arithmetic above
        while (j > 0) { const HitRec o = get(j - 1); if (!(rankOf(o.code) > r)) break; set(j, o); j--; }
        set(j, h);
    }
    auto insertionSort = [&](int lo, int hi) {
        for (int i = lo + 1; i <= hi; i++) {
            const HitRec h = get(i);
            int j = i;
            while (j > lo) { const HitRec o = get(j - 1); if (!(h.t < o.t)) break; set(j, o); j--; }
            set(j, h);
        }
    };
    auto heapify = [&](int i, int n, int lo) {
        const HitRec val = get(lo + i - 1);
        while (i <= n / 2) {
            int child = 2 * i;
            HitRec c = get(lo + child - 1);
            if (child < n) { const HitRec c2 = get(lo + child); if (c.t < c2.t) { child++; c = c2; } }
            if (c.t < val.t) break;
            set(lo + i - 1, c);
            i = child;
        }
        set(lo + i - 1, val);
    };
    auto heapSort = [&](int lo, int hi) {
        const int n = hi - lo + 1;
        for (int i = n / 2; i >= 1; i--) heapify(i, n, lo);
        for (int i = n; i > 1; i--) { swapHits(lo, lo + i - 1); heapify(1, i - 1, lo); }
    };
    // pending ranges: every partition step pushes one and uses up one of the 2 * floor(log2(n)) levels
    constexpr int kRanges = 40;
    int rangeLo[kRanges], rangeHi[kRanges], rangeDepth[kRanges], ranges = 1;
    int log2floor = 0;
    while ((nHits >> (log2floor + 1)) != 0) log2floor++;
    rangeLo[0] = 0; rangeHi[0] = nHits - 1; rangeDepth[0] = 2 * log2floor;
    while (ranges > 0) {
        ranges--;
        int lo = rangeLo[ranges], hi = rangeHi[ranges], depth = rangeDepth[ranges];
        while (hi > lo) {
            const int size = hi - lo + 1;
            if (size == 2) { swapIfGreater(lo, hi); break; }
            if (size == 3) { swapIfGreater(lo, hi - 1); swapIfGreater(lo, hi); swapIfGreater(hi - 1, hi); break; }
            if (size <= 16) { insertionSort(lo, hi); break; }
            if (depth == 0) { heapSort(lo, hi); break; }
            depth--;
            const int mid = lo + (hi - lo) / 2;
            swapIfGreater(lo, mid); swapIfGreater(lo, hi); swapIfGreater(mid, hi);
            const float pivot = get(mid).t;
            swapHits(mid, hi - 1);
            int left = lo, right = hi - 1;
            while (left < right) {
                while (pivot > get(++left).t) {}
                while (pivot < get(--right).t) {}
                if (left >= right) break;
                swapHits(left, right);
            }
            swapHits(left, hi - 1);
            rangeLo[ranges] = left + 1; rangeHi[ranges] = hi; rangeDepth[ranges] = depth; ranges++;
            hi = left - 1;
        }
    }
}

// [hit-list sorts: end]
// ------------------------------------------------------------------------------------------------------------
// scene access: LDS image first, HBM/L2 for whatever did not fit
// ------------------------------------------------------------------------------------------------------------
struct SceneRefs {
    const uint8_t* lds;     // LDS copy of the blob prefix
    const uint8_t* glob;    // full blob in HBM
    uint32_t ldsNodeCount;
};

template <bool ALL_LDS, bool SPLIT = false>
__host__ __device__ __forceinline__ void load_node(const SceneRefs& sc, const SceneLayout& L, int idx, float4& q0, float4& q1, float4& q2, int& c0, int& c1)
{
    const uint32_t off = L.nodeOffset + (uint32_t)idx * 64u;
#if defined(__HIP_DEVICE_COMPILE__) && RTOW_SPLIT_NODE_LOADS
    if (!ALL_LDS && SPLIT) {
        // A tree that does not fit LDS keeps its first ldsNodeCount nodes (the top levels) there; the blob in memory holds every node.  Selecting the BASE per lane makes
        // every node load a flat_load (address-space check per lane, both memory counters, seven instructions to build the generic pointer), and a wave waits for its
        // slowest lane anyway: so the wave reads from LDS when ALL its walking lanes are in the top levels and through L1 / L2 otherwise - ds_read or global_load
        // (scalar base + 32-bit offset), never flat.  (Per-lane branches - ds_read under one EXEC mask, global_load under the other - make the compiler wait for the
        // first group before it issues the second: both write the same registers.)  SPLIT = the walk of the kernels with 16-bit codes: 10 000 spheres +0.6 %, same box,
        // three alternating runs each, every run above the other side's best.  The wide-code kernels keep the flat loads: their trees' lower levels miss L1 and L2, and
        // the lanes in the top levels are better off in LDS whatever the others do (250 882 triangles: -5 % with the split; profiles/r05q_node_loads.json).
        typedef __attribute__((address_space(3))) const uint8_t* LdsBytes;
        typedef __attribute__((address_space(1))) const uint8_t* GlobalBytes;
        if (__ballot((uint32_t)idx >= sc.ldsNodeCount) == 0ull) {      // wave-uniform: every lane that walks right now is in the top levels
            LdsBytes b = (LdsBytes)sc.lds + off;
            q0 = *(__attribute__((address_space(3))) const float4*)(b);
            q1 = *(__attribute__((address_space(3))) const float4*)(b + 16);
            q2 = *(__attribute__((address_space(3))) const float4*)(b + 32);
            // the two child codes through an asm statement (with its own wait: the compiler's wait-count bookkeeping does not see into it): as an ordinary load the
            // compiler sinks one of the two dwords behind the branch - as a flat load through a phi of both pointers, one more memory instruction per node visit
            typedef int i2 __attribute__((ext_vector_type(2)));
            i2 c;
            asm volatile("ds_read_b64 %0, %1 offset:48\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"((unsigned)(uintptr_t)b) : "memory");
            c0 = c.x; c1 = c.y;
        } else {
            GlobalBytes b = (GlobalBytes)sc.glob + off;
            q0 = *(__attribute__((address_space(1))) const float4*)(b);
            q1 = *(__attribute__((address_space(1))) const float4*)(b + 16);
            q2 = *(__attribute__((address_space(1))) const float4*)(b + 32);
            const int2 c = *(__attribute__((address_space(1))) const int2*)(b + 48);
            c0 = c.x; c1 = c.y;
        }
        return;
    }
#endif
    const uint8_t* base = (ALL_LDS || (uint32_t)idx < sc.ldsNodeCount) ? sc.lds : sc.glob;
    const float4* p = reinterpret_cast<const float4*>(base + off);
    q0 = p[0];
    q1 = p[1];
    q2 = p[2];
    const int2 c = *reinterpret_cast<const int2*>(base + off + 48);
    c0 = c.x;
    c1 = c.y;
}

template <bool ALL_LDS>
__host__ __device__ __forceinline__ const uint8_t* section(const SceneRefs& sc, uint32_t offset)
{
    return (ALL_LDS ? sc.lds : sc.glob) + offset;
}

// centre of primitive `i` at ray time `time` (Entity.TransformAtTime, RT/Entity.cs:124-127) and its signed radius
template <bool ALL_LDS, bool HAS_MOTION>
__host__ __device__ __forceinline__ void sphere_at(const SceneRefs& sc, const SceneLayout& L, int i, float time, V3& c, float& radius)
{
    const float4 s = *reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.sphereOffset) + (uint32_t)i * 16u);
    c = v3(s.x, s.y, s.z);
    radius = s.w;
    if (HAS_MOTION) {
        const uint8_t* mp = section<ALL_LDS>(sc, L.motionOffset) + (uint32_t)i * 32u;
        const float4 m0 = *reinterpret_cast<const float4*>(mp);      // dx dy dz t0
        const float2 m1 = *reinterpret_cast<const float2*>(mp + 16); // t1 moving
        if (__builtin_bit_cast(int, m1.y) != 0) {
            // clamp(unlerp(t0, t1, t), 0, 1); when every moving entity shares one TimeRange (L.commonTimeRange) `time` already IS that value:
            // the sample's ray time goes through the expression once, in REGEN, instead of once per sphere test (same operands, same result)
            const float f = L.commonTimeRange ? time : um_max(0.0f, um_min(1.0f, (time - m0.w) / (m1.x - m0.w)));
            c = v3(c.x + m0.x * f, c.y + m0.y * f, c.z + m0.z * f);
        }
    }
}

// (p.x / d, p.y / d, p.z / d): three IEEE divisions by one divisor (a sphere's outward normal, r.GetPoint(t) / radius, RT/HitTests.cs:56)
__host__ __device__ __forceinline__ V3 div3(V3 p, float d)
{
#if RTOW_EXACT_DIV3
    V3 q;
    rtow::exact_div3(p.x, p.y, p.z, d, q.x, q.y, q.z);
    return q;
#else
    return v3(p.x / d, p.y / d, p.z / d);
#endif
}

// HitTests.Hit(Sphere) (RT/HitTests.cs:23-60) in entity space (oc = origin - centre), tMin = 0, tMax = +inf
__host__ __device__ __forceinline__ bool sphere_hit(V3 oc, V3 d, float a, float radius, float& tOut)
{
    const float b = dot(oc, d);
    const float c = dot(oc, oc) - radius * radius;
    const float disc = b * b - a * c;
    if (disc > 0) {
        // t = (-b -+ sq) / a with a = dot(d, d) >= 0: a numerator that is not positive gives a quotient that is not positive (or NaN) and
        // fails `t > 0` whatever a is, so its IEEE division is skipped - bit-identical, and the common "sphere behind the origin" case
        // (every ray leaving the ground sphere) costs no division at all.
        const float sq = RTOW_SQRT(disc);
        const float n0 = -b - sq;
        if (n0 > 0) {
            const float t = n0 / a;
            if (t < __builtin_inff() && t > 0) { tOut = t; return true; }
        }
        const float n1 = -b + sq;
        if (n1 > 0) {
            const float t = n1 / a;
            if (t < __builtin_inff() && t > 0) { tOut = t; return true; }
        }
    }
    return false;
}

// the same test with an arbitrary tMin (strict: t > tMin), RT/HitTests.cs:40,49
__host__ __device__ __forceinline__ bool sphere_hit_tmin(V3 oc, V3 d, float a, float radius, float tMin, float& tOut)
{
    const float b = dot(oc, d);
    const float c = dot(oc, oc) - radius * radius;
    const float disc = b * b - a * c;
    if (disc > 0) {
        const float sq = RTOW_SQRT(disc);                    // tMin >= 0 here: the numerator shortcut of sphere_hit applies unchanged
        const float n0 = -b - sq;
        if (n0 > 0) {
            const float t = n0 / a;
            if (t < __builtin_inff() && t > tMin) { tOut = t; return true; }
        }
        const float n1 = -b + sq;
        if (n1 > 0) {
            const float t = n1 / a;
            if (t < __builtin_inff() && t > tMin) { tOut = t; return true; }
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------------
// general entities (SCENE_KIND_GENERAL): Rect / Box / Triangle and rotated or moving transforms, RT/Entity.cs:58-127
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// math.mul(quaternion q, float3 v): t = 2 * cross(q.xyz, v); v + q.w * t + cross(q.xyz, t)
__host__ __device__ __forceinline__ V3 rotate(float4 q, V3 v)
{
    const V3 qv = v3(q.x, q.y, q.z);
    const V3 t = scale(2.0f, cross(qv, v));
    const V3 c = cross(qv, t);
    return v3(v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z);
}
__host__ __device__ __forceinline__ float um_sign(float x) { return (x > 0.0f ? 1.0f : 0.0f) - (x < 0.0f ? 1.0f : 0.0f); }

// Entity.HitInternal + HitContent for primitive `i` (RT/Entity.cs:74-122) with tMax = +inf (tMin = 0 except for the exit-hit
// probe of volume hulls, JOBS/SampleBatchJob.cs:465).
// Returns the distance, the entity-space normal and the rotation that takes it to world space.
template <bool ALL_LDS, bool TRIANGLES_ONLY = false>
__host__ __device__ __forceinline__ bool general_hit(const SceneRefs& sc, const SceneLayout& L, int i, unsigned type, V3 ro, V3 rd, float time, float tMin,
                                            float& tOut, V3& nLocal, float4& rot, float2* texCoord = nullptr)
{
    const float4* p = reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.primOffset) + (uint32_t)i * 128u);
    if (texCoord) *texCoord = make_float2(0, 0);       // only triangles have texture coordinates (RT/Entity.cs:108, RT/HitTests.cs:123)
    if (TRIANGLES_ONLY || type == RTOW_ENTITY_TRIANGLE) {       // TRIANGLES_ONLY (SCENE_KIND_TRIANGLES): the other primitives' code is not compiled in
        // HitTests.Hit(Triangle) (RT/HitTests.cs:115-150); triangles are tested in world space (RT/Entity.cs:91-93)
        // the first three quads (edges, v0, first normal) decide the test; the rest of the record - its second cache line when it is read from HBM -
        // is only fetched for a hit
        const float4 a0 = p[0], a1 = p[1], a2 = p[2];
        const V3 e0 = v3(a0.x, a0.y, a0.z), e1 = v3(a0.w, a1.x, a1.y), v0 = v3(a1.z, a1.w, a2.x);
        const V3 pvec = cross(rd, e0);
        const float det = dot(e1, pvec);
        if (det == 0) return false;
        const float invDet = RTOW_RCP(det);
        const V3 tvec = sub(ro, v0);
        const float u = dot(tvec, pvec) * invDet;
        if (u < 0 || u > 1) return false;
        const V3 qvec = cross(tvec, e1);
        const float v = dot(rd, qvec) * invDet;
        if (v < 0 || u + v > 1) return false;
        const float dist = dot(e0, qvec) * invDet;
        if (dist < tMin || dist > __builtin_inff()) return false;
        const float b0 = 1 - u - v;
        const float4 a3 = p[3], a4 = p[4];
        rot = p[6];
        const V3 n0 = v3(a2.y, a2.z, a2.w), n1 = v3(a3.x, a3.y, a3.z), n2 = v3(a3.w, a4.x, a4.y);
        nLocal = v3(n0.x * b0 + n1.x * u + n2.x * v, n0.y * b0 + n1.y * u + n2.y * v, n0.z * b0 + n1.z * u + n2.z * v);
        if (texCoord) {                                  // mul(tri.TextureCoordinates, barycentricCoords) (:148): float2x3 columns t0 t1 t2
            const float4 a5 = p[5];
            *texCoord = make_float2(a4.z * b0 + a5.x * u + a5.z * v, a4.w * b0 + a5.y * u + a5.w * v);
        }
        tOut = dist;
        return true;
    }
    rot = p[0];
    const float4 invRot = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5];
    V3 invT = v3(q4.y, q4.z, q4.w);
    if (__builtin_bit_cast(int, q2.w) != 0) {
        // TransformAtTime (RT/Entity.cs:124-127) and its inverse (:87-88): invTranslation = mul(invRot, -pos(t))
        const float f = um_max(0.0f, um_min(1.0f, (time - q3.w) / (q4.x - q3.w)));
        const V3 pt = v3(q2.x + q3.x * f, q2.y + q3.y * f, q2.z + q3.z * f);
        invT = rotate(invRot, neg(pt));
    }
    const V3 oL = add(rotate(invRot, ro), invT);     // transform(inverseTransform, ray.Origin)
    const V3 dL = rotate(invRot, rd);                // rotate(inverseTransform, ray.Direction)
    if (type == RTOW_ENTITY_SPHERE) {
        float t;
        if (!sphere_hit_tmin(oL, dL, dot(dL, dL), q5.x, tMin, t)) return false;
        nLocal = div3(v3(oL.x + t * dL.x, oL.y + t * dL.y, oL.z + t * dL.z), q5.x);
        tOut = t;
        return true;
    }
    if (type == RTOW_ENTITY_RECT) {
        // HitTests.Hit(Rect) (RT/HitTests.cs:62-78)
        if (dL.z >= 0) return false;
        const float t = -oL.z / dL.z;
        if (t < tMin || t > __builtin_inff()) return false;
        const float x = oL.x + t * dL.x, y = oL.y + t * dL.y;
        if (x < q5.x || y < q5.y || x > q5.z || y > q5.w) return false;
        nLocal = v3(0, 0, 1);
        tOut = t;
        return true;
    }
    // HitTests.Hit(Box) (RT/HitTests.cs:80-113): the origin is first advanced by tMin (origin + direction * tMin)
    const float4 q6 = p[6];
    const V3 ext = v3(q5.x, q5.y, q5.z), invExt = v3(q5.w, q6.x, q6.y);
    const V3 o = v3(oL.x + dL.x * tMin, oL.y + dL.y * tMin, oL.z + dL.z * tMin);
    const float winding = um_max(um_max(__builtin_fabsf(o.x) * invExt.x, __builtin_fabsf(o.y) * invExt.y), __builtin_fabsf(o.z) * invExt.z) < 1 ? -1.0f : 1.0f;
    V3 sgn = v3(-um_sign(dL.x), -um_sign(dL.y), -um_sign(dL.z));
    const V3 dtp = v3((ext.x * winding * sgn.x - o.x) / dL.x, (ext.y * winding * sgn.y - o.y) / dL.y, (ext.z * winding * sgn.z - o.z) / dL.z);
    const bool tx = dtp.x >= 0 && __builtin_fabsf(o.y + dL.y * dtp.x) < ext.y && __builtin_fabsf(o.z + dL.z * dtp.x) < ext.z;
    const bool ty = dtp.y >= 0 && __builtin_fabsf(o.z + dL.z * dtp.y) < ext.z && __builtin_fabsf(o.x + dL.x * dtp.y) < ext.x;
    const bool tz = dtp.z >= 0 && __builtin_fabsf(o.x + dL.x * dtp.z) < ext.x && __builtin_fabsf(o.y + dL.y * dtp.z) < ext.y;
    sgn = tx ? v3(sgn.x, 0, 0) : ty ? v3(0, sgn.y, 0) : v3(0, 0, tz ? sgn.z : 0);
    if (!(sgn.x != 0 || sgn.y != 0 || sgn.z != 0)) return false;
    float dist = sgn.x != 0 ? dtp.x : sgn.y != 0 ? dtp.y : dtp.z;
    dist += tMin;
    if (dist > __builtin_inff()) return false;
    nLocal = sgn;
    tOut = dist;
    return true;
}

// HitTests.Hit(Triangle) (RT/HitTests.cs:115-150) on the compact record of an all-triangle scene (GpuTriHot, rtow_scene.h): the same expressions as general_hit's triangle
// branch on the same operands, up to the distance; the barycentric (u, v) are handed back instead of the blended normal, which only the ray's nearest hit needs (tri_normal_cold)
template <bool ALL_LDS>
__host__ __device__ __forceinline__ bool tri_hit_hot(const SceneRefs& sc, const SceneLayout& L, int i, V3 ro, V3 rd, float tMin, float& tOut, float& uOut, float& vOut)
{
    const float4* p = reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.triHotOffset) + (uint32_t)i * (uint32_t)sizeof(GpuTriHot));
    const float4 a0 = p[0], a1 = p[1];
    const float a2x = *reinterpret_cast<const float*>(p + 2);
    const V3 e0 = v3(a0.x, a0.y, a0.z), e1 = v3(a0.w, a1.x, a1.y), v0 = v3(a1.z, a1.w, a2x);
    const V3 pvec = cross(rd, e0);
    const float det = dot(e1, pvec);
    if (det == 0) return false;
    const float invDet = RTOW_RCP(det);
    const V3 tvec = sub(ro, v0);
    const float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    const V3 qvec = cross(tvec, e1);
    const float v = dot(rd, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    const float dist = dot(e0, qvec) * invDet;
    if (dist < tMin || dist > __builtin_inff()) return false;
    tOut = dist;
    uOut = u;
    vOut = v;
    return true;
}
// the rest of that test for the hit that won: mul(tri.Normals, barycentricCoords) (RT/HitTests.cs:140-146) from the GpuTriCold record, and the entity's rotation
template <bool ALL_LDS>
__host__ __device__ __forceinline__ V3 tri_normal_cold(const SceneRefs& sc, const SceneLayout& L, int i, float u, float v, float4& rot)
{
    const float4* p = reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.triColdOffset) + (uint32_t)i * (uint32_t)sizeof(GpuTriCold));
    const float4 c0 = p[0], c1 = p[1];
    const float c2x = *reinterpret_cast<const float*>(p + 2);
    rot = p[3];
    const float b0 = 1 - u - v;
    const V3 n0 = v3(c0.x, c0.y, c0.z), n1 = v3(c0.w, c1.x, c1.y), n2 = v3(c1.z, c1.w, c2x);
    return v3(n0.x * b0 + n1.x * u + n2.x * v, n0.y * b0 + n1.y * u + n2.y * v, n0.z * b0 + n1.z * u + n2.z * v);
}

typedef float f2 __attribute__((ext_vector_type(2)));

// A request that brings one memory sector into the caches and lands in no register: a global -> LDS load of one dword per lane into a 256-byte dump row of the workgroup
// (nothing reads it).  The walk of a tree that lives in HBM is a chain of dependent misses (profiles/r05_mesh_pmc_summary.json: 62 % of wave cycles in s_waitcnt, 5.7 L2
// misses per ray); what is known ahead of its use - the far child that was just pushed, the triangle that was just listed - is asked for at once and arrives while the
// near subtree is walked.  M0 (the LDS destination) is saved and restored around the instruction.
__device__ __forceinline__ void prefetch_sector(const uint8_t* base, uint32_t byteOffset, uint32_t ldsDump)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(byteOffset), "s"(base), "s"(ldsDump) : "memory");
#else
    (void)base; (void)byteOffset; (void)ldsDump;
#endif
}

// Raw VALU min/max (IEEE mode: a NaN operand yields the other operand).  __builtin_fminf/fmaxf would first canonicalise
// both inputs (v_max_f32 x, x), which doubles the instruction count of the slab test for nothing.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Wave priority per stage (s_setprio at every stage entry; two bits per stage number: REGEN TRAV TEST HIT SKY VOL - scheduler).  The exact
// tests run with a third of the lanes through dependent sqrt / division chains; letting a wave in that stage issue ahead of its three
// SIMD neighbours gets it back to the walk sooner: +2 ... +3.5 % on the headline workload, same box, alternating runs (DESIGN.md 4.1,
// profiles/r02_runs/run_r04m.sh).  Scheduling only - no effect on results.
#ifndef RTOW_STAGE_PRIO
#define RTOW_STAGE_PRIO (1 << (2 * 2))   // exact tests at priority 1, everything else 0
#endif
// (The first build with these instructions exposed a miscompiled tie update in the general-entity kernels - see the TEST stage - which
// is how that one was found; with the update written as selects every variant equals the oracle with and without them.)
#define STAGE_PRIO(k) do { if (HURRY && hurry != 0ull) __builtin_amdgcn_s_setprio(3); else if ((RTOW_STAGE_PRIO) != 0) __builtin_amdgcn_s_setprio((short)(((RTOW_STAGE_PRIO) >> (2 * (k))) & 3)); } while (0)
// instrumentation hooks: empty in the product.  profiles/experiments/instrumentation.patch (applied by profiles/experiments/build.sh to a COPY of this
// directory, never to the shipped sources) defines them for the stage-statistics build and adds the timing experiments of HISTORY.md.
#define DBG_TRACE(kind, primv, tv)
#define STAT_DECL
#define STAGE_DECL
#define STAGE_MARK(k) STAGE_PRIO(k)
#define STAGE_MARK_TOP(k) do { if ((RTOW_STAGE_PRIO) != 0) __builtin_amdgcn_s_setprio((short)(((RTOW_STAGE_PRIO) >> (2 * (k))) & 3)); } while (0)      // (top of a trip: before this trip's lanes in a hurry are known)
#define STAT_ADD(i, v)
#define STAT_LANES(i)

// ------------------------------------------------------------------------------------------------------------
// path history: one 16-bit code per surface hit (bit 15 = "reflectance was overridden to 1", bits 0..14 = material).
// The reference keeps float3 emission / attenuation stacks (JOBS/SampleBatchJob.cs:103-104,311,330) and folds them
// tail -> head (:384-396); the codes are re-expanded at the fold, which reproduces the fold order bit for bit.
// Depth <= 8 and <= 16 keep the codes in named 64-bit registers (an indexed array would be demoted to scratch).
// ------------------------------------------------------------------------------------------------------------
// Deeper paths (HW = 32: the generic variants, trace depth up to 64): the first kHistoryInRegisters codes in registers like the others, the rest in LDS - rows of 1024
// 16-bit codes behind the traversal stack, [depth - 8][lane] (LdsPlan, rtow_kernels.h).  Rounds 1 - 5 kept 32 words per lane in a private segment and cleared them per sample
// (the reference host's committed traceDepth 32: 186 GB of HBM writes per 10-batch launch); a path rarely gets that deep (2.5 segments on average), so the rows cost a few
// ds_write_b16 per thousand hits, and nothing is cleared: a row entry is written before the fold reads it.
template <bool SPILL> struct HistRowsT { unsigned short* lane; };     // this lane's entry of row 0 in LDS (null where the variant keeps every code in registers)
using HistRows = HistRowsT<false>;
// Rows that do not fit LDS (wide stack rows next to a deep trace depth; a scene that is kept whole in LDS instead) live in HBM, [row][lane] like the LDS rows
// (SampleKernelArgs.histSpill).  The launch constants are read from the kernarg segment on use (through a laundered pointer, like the pixel boundary's): paths that deep
// are rare, and nothing of this may sit in a register through the stages.  (Not a call: a call frame would give these variants their private segment back.)
// SPILL = the variant's launches may have rows in HBM: the kernels whose scene is not whole in LDS (a launch that keeps its scene whole has all its rows in LDS: planLds) -
// the extra code gave the scene-in-LDS variants their 36-byte private segment back.
template <bool SPILL>
__device__ __forceinline__ unsigned short* hist_row_slot(unsigned short* ldsLane, int row)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (SPILL) {
        const SampleKernelArgs* A = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();      // the struct is the kernel's only argument
        asm volatile("" : "+s"(A));
        if (row >= (int)A->ldsHistRows) return A->histSpill + (size_t)(row - (int)A->ldsHistRows) * (size_t)A->histSpillStride + (size_t)blockIdx.x * kBlockThreads + threadIdx.x;
    }
#endif
    return ldsLane + row * kBlockThreads;
}
template <int HW> struct Hist {
    unsigned w[HW];
    __device__ __forceinline__ void clear() { for (int i = 0; i < HW; i++) w[i] = 0; }
    template <class R> __device__ __forceinline__ void set(int depth, unsigned code, const R&) { w[depth >> 1] |= code << ((unsigned)(depth & 1) * 16u); }
    template <class R> __device__ __forceinline__ unsigned get(int depth, const R&) const { return (w[depth >> 1] >> ((unsigned)(depth & 1) * 16u)) & 0xffffu; }
};
template <> struct Hist<4> {
    unsigned long long a, b;
    __device__ __forceinline__ void clear() { a = 0; b = 0; }
    template <class R> __device__ __forceinline__ void set(int depth, unsigned code, const R&)
    {
        const unsigned long long v = (unsigned long long)code << ((unsigned)(depth & 3) * 16u);
        if (depth < 4) a |= v; else b |= v;
    }
    template <class R> __device__ __forceinline__ unsigned get(int depth, const R&) const { return (unsigned)((depth < 4 ? a : b) >> ((unsigned)(depth & 3) * 16u)) & 0xffffu; }
};
template <> struct Hist<8> {
    unsigned long long a, b, c, d;
    __device__ __forceinline__ void clear() { a = 0; b = 0; c = 0; d = 0; }
    template <class R> __device__ __forceinline__ void set(int depth, unsigned code, const R&)
    {
        const unsigned long long v = (unsigned long long)code << ((unsigned)(depth & 3) * 16u);
        const int q = depth >> 2;
        if (q == 0) a |= v; else if (q == 1) b |= v; else if (q == 2) c |= v; else d |= v;
    }
    template <class R> __device__ __forceinline__ unsigned get(int depth, const R&) const
    {
        const int q = depth >> 2;
        const unsigned long long v = q == 0 ? a : q == 1 ? b : q == 2 ? c : d;
        return (unsigned)(v >> ((unsigned)(depth & 3) * 16u)) & 0xffffu;
    }
};
template <> struct Hist<32> {
    Hist<4> head;
    static_assert(kHistoryInRegisters == 8, "Hist<4> holds the register-resident part");
    __device__ __forceinline__ void clear() { head.clear(); }
    template <bool SPILL> __device__ __forceinline__ void set(int depth, unsigned code, const HistRowsT<SPILL>& rows)
    {
        if (depth < kHistoryInRegisters) head.set(depth, code, rows);
        else *hist_row_slot<SPILL>(rows.lane, depth - kHistoryInRegisters) = (unsigned short)code;
    }
    template <bool SPILL> __device__ __forceinline__ unsigned get(int depth, const HistRowsT<SPILL>& rows) const
    {
        if (depth < kHistoryInRegisters) return head.get(depth, rows);
        return *hist_row_slot<SPILL>(rows.lane, depth - kHistoryInRegisters);
    }
};

// ------------------------------------------------------------------------------------------------------------
// Chained batches: accumulators handed from one batch to the next INSIDE a running kernel.  gfx950 has eight XCDs with one L2 each, and the L2s
// are not coherent with each other for ordinary device memory.  Measured on the part (profiles/calib/coherence_probe.hip, xcd_affinity_probe.hip):
//   * between workgroups on DIFFERENT XCDs plain accesses read stale lines every time; release / acquire fences at agent scope are correct but write
//     the whole L2 back (4 us per release); write-through stores + L2-bypassing loads (sc1 on both sides) are correct and cheap, but a write-through
//     store that leaves lane by lane is its own 32-byte memory write (round 2 shipped this: 3.7 x the algorithmic write traffic);
//   * between workgroups on the SAME XCD plain (write-back) stores followed by s_waitcnt vmcnt(0), and sc1 loads on the reading side - which miss
//     the CU's L1 and are served by the XCD's L2 from its own dirty lines - are correct: 0 stale dwords of 655 M (plain and sc0 loads read the
//     reader's stale L1 copy every time).
// So a chain keeps all batches of a pixel chunk on ONE XCD - whichever XCD takes a chunk for batch 0 owns it for the rest of the launch (XcdState
// below) - and its accumulator traffic is ordinary: plain stores that merge in L2 like a single launch's, loads that bypass L1.  A chunk is
// published by: stores -> s_waitcnt vmcnt(0) (coherent_flush) -> relaxed agent-scope add on the chunk's counter.
// The loads are inline assembly (there is no builtin for a 16- / 12-byte agent-scope load) and wait for their own data: the compiler's
// wait-count bookkeeping does not see into an asm statement.
// ------------------------------------------------------------------------------------------------------------
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef float fvec3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float coherent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_flush() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
__device__ __forceinline__ float4 coherent_load4(const float* p)
{
    fvec4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ V3 coherent_load3(const float* p)
{
    fvec3 v;
    asm volatile("global_load_dwordx3 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v3(v.x, v.y, v.z);
}
// which XCD this wave runs on (gfx940+: XCC_ID, bits 3..0)
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & (kMaxXcds - 1u);
}
constexpr unsigned kChainShift = 27;                       // ticket = batch << 27 | owned-pixel number (chains need fewer than 2^27 padded pixels)
constexpr unsigned kChainTicketMask = (1u << kChainShift) - 1u;

// ------------------------------------------------------------------------------------------------------------
// the megakernel
// ------------------------------------------------------------------------------------------------------------
// Lane states.  Every trip of the main loop the wavefront takes a population vote (__ballot + popcount per state) and
// runs ONLY the stage with the most lanes waiting; the other lanes keep their state and wait.  Rare, expensive stages
// (glass, rough metal) therefore execute with many lanes batched up instead of being dragged through every trip by
// one straggler, and the box walk never waits for the slowest ray: this is what replaces per-bounce compaction.
enum : int {
    ST_REGEN = 0,    // needs its next sample (or next pixel)
    ST_TRAV = 1,     // walking BVH boxes (resumable; at most A.travSlice node visits per trip)
    ST_TEST = 2,     // has leaf candidates awaiting the exact sphere test
    ST_HIT = 3,      // nearest hit known: shade
    ST_SKY = 4,      // missed everything: sky + fold
    ST_VOL = 5,      // VOLUMES scenes: all hits collected -> sort, containment probe, volume logic (JOBS/SampleBatchJob.cs:194-303)
    ST_DEAD = 6,
    ST_COUNT = 6,
    ST_IDLE = 7      // chained batches: the lane's next pixel belongs to a chunk whose previous batch is not stored yet; it asks again every few trips
};

// ------------------------------------------------------------------------------------------------------------
// Nearest-hit tie, the long way.  Scenes without volumes keep only the nearest hit and break a tie by leaf order - which is what the
// reference's sort of the whole hit list leaves in front as long as the ray has at most 16 hits; beyond that its partition steps
// decide.  When TEST saw a second surface at exactly the nearest distance, this runs the reference's own procedure for
// that one ray: every hit of the ray (the walk again, not pruned, same box tests), in leaf order, through the same sort; element 0 wins.
// A real call: it is rare, and its list lives in scratch.
// ------------------------------------------------------------------------------------------------------------
template <bool ALL_LDS, int KIND, typename Code, int BT>
__device__ __noinline__ __attribute__((unused)) int resolve_nearest_tie(const SceneRefs sc, const SceneLayout* layoutInKernarg, V3 ro, V3 rd, float rtime, Code* stack,
                                                                       uint32_t* overflowFlag, HitSpill spill)
{
    // The scene layout is read through a pointer into the kernarg segment, the scene references travel by value: taking the address of the kernel's
    // own copies for a by-reference parameter forced those copies - 25 dwords every stage reads - into scratch for the whole kernel (the exact-tie
    // variants ran 7 - 17 % behind their rank-rule twins "whether the call is taken or not": most of it was this).
    const SceneLayout& L = *layoutInKernarg;
    float hitT[kLocalHits], hitDummy[kLocalHits];
    unsigned hitCode[kLocalHits];
    int n = 0;
    V3 inv = v3(RTOW_RCP(rd.x), RTOW_RCP(rd.y), RTOW_RCP(rd.z));                     // startRay's rayInvDirection
    if (inv.x != inv.x) inv.x = __builtin_inff();
    if (inv.y != inv.y) inv.y = __builtin_inff();
    if (inv.z != inv.z) inv.z = __builtin_inff();
    const bool twoChildren = L.sphereCount > 1u;
    const float a = dot(rd, rd);
    int sp = 0, cur = 0;
    while (cur >= 0) {
        float4 q0, q1, q2;
        int c0, c1;
        load_node<ALL_LDS>(sc, L, cur, q0, q1, q2, c0, c1);
        // the walk's slab expressions (TRAV), one child at a time
        const float t0x = (q0.x - ro.x) * inv.x, t1x = (q1.z - ro.x) * inv.x, u0x = (q0.y - ro.x) * inv.x, u1x = (q1.w - ro.x) * inv.x;
        const float t0y = (q0.z - ro.y) * inv.y, t1y = (q2.x - ro.y) * inv.y, u0y = (q0.w - ro.y) * inv.y, u1y = (q2.y - ro.y) * inv.y;
        const float t0z = (q1.x - ro.z) * inv.z, t1z = (q2.z - ro.z) * inv.z, u0z = (q1.y - ro.z) * inv.z, u1z = (q2.w - ro.z) * inv.z;
        const float tmin0 = vmax3(vmin(t0x, t1x), vmin(t0y, t1y), vmax(vmin(t0z, t1z), 0.0f)), tfar0 = vmin3(vmax(t0x, t1x), vmax(t0y, t1y), vmax(t0z, t1z));
        const float tmin1 = vmax3(vmin(u0x, u1x), vmin(u0y, u1y), vmax(vmin(u0z, u1z), 0.0f)), tfar1 = vmin3(vmax(u0x, u1x), vmax(u0y, u1y), vmax(u0z, u1z));
        const bool hit0 = tmin0 <= tfar0, hit1 = tmin1 <= tfar1 && twoChildren;
        for (int side = 0; side < 2; side++) {
            const int cc = side ? c1 : c0;
            if (cc >= 0 || !(side ? (hit1 && tmin1 < tfar1) : (hit0 && tmin0 < tfar0))) continue;     // leaf children: AxisAlignedBoundingBox.Hit, tMin < tMax
            const int i = ~cc;
            float t;
            bool ok;
            if (KIND >= SCENE_KIND_GENERAL) {
                constexpr bool TRI = KIND == SCENE_KIND_TRIANGLES || KIND == SCENE_KIND_TRIANGLES_TEXTURED;
                const unsigned type = TRI ? (unsigned)RTOW_ENTITY_TRIANGLE : *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (uint32_t)i * 4u) >> kPrimTypeShift;
                V3 nl; float4 rq;
                if (TRI && RTOW_TRI_HOT) { float uu, vv; ok = tri_hit_hot<ALL_LDS>(sc, L, i, ro, rd, 0.0f, t, uu, vv); }
                else ok = general_hit<ALL_LDS, TRI>(sc, L, i, type, ro, rd, rtime, 0.0f, t, nl, rq);
            } else {
                V3 c; float r;
                sphere_at<ALL_LDS, KIND == SCENE_KIND_SPHERES_MOTION>(sc, L, i, rtime, c, r);
                ok = sphere_hit(sub(ro, c), rd, a, r, t);
            }
            if (!ok) continue;
            if ((unsigned)n < (unsigned)kLocalHits + spill.entries) { hit_set(hitT, hitDummy, hitCode, spill, n, HitRec{t, 0.0f, (unsigned)i}); n++; }
            else *overflowFlag = 1u;                                                                       // more hits than the context's hitListCapacity: RTOW_ERROR_CAPACITY on the host side
        }
        const bool in0 = hit0 && c0 >= 0, in1 = hit1 && c1 >= 0;
        if (in0 && in1) { stack[sp * BT] = (Code)c1; sp++; cur = c0; }
        else if (in0 || in1) cur = in0 ? c0 : c1;
        else if (sp > 0) { sp--; cur = (int)stack[sp * BT]; }
        else cur = -1;
    }
    if (n == 0) return -1;
    const unsigned* rank = reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.rankOffset));
    if (n > kLocalHits) sort_hit_list_spilled(hitT, hitDummy, hitCode, spill, n, rank);
    else if (n > 1) sort_hit_list(hitT, hitDummy, hitCode, n, rank);
    return (int)(hitCode[0] & kHitPrimMask);
}

// ------------------------------------------------------------------------------------------------------------
// FULL_DIAGNOSTICS as the reference counts them (RTOW_CONTEXT_REFERENCE_DIAGNOSTICS): FindHitCandidates walks the tree RebuildBvh built
// and counts every node whose box the ray passes and every entity of the leaves it reaches (JOBS/SampleBatchJob.cs:403-448, counters
// :427-440; consumed by the heat-map views, UNITY/Raytracer.cs:1015-1031).  The product walks its own, pruned tree, so in this mode the
// reference's tree (rtow_reforder.h: RefTreeNode, from HBM through L2) is walked a second time, unpruned, only to count.
// AxisAlignedBoundingBox.Hit as in the reference (RT/HitTests.cs:9-21).  A real call: only batches that asked for it pay.
// ------------------------------------------------------------------------------------------------------------
// (returns {boundsHits, candidates} by value: out-parameters would pin the caller's two counters to scratch for the whole kernel)
__device__ __noinline__ __attribute__((unused)) float2 reference_counts(const uint8_t* tree, V3 ro, V3 rd)
{
    V3 inv = v3(RTOW_RCP(rd.x), RTOW_RCP(rd.y), RTOW_RCP(rd.z));                     // rcp(ray.Direction), NaN -> +INF (:406-412)
    if (inv.x != inv.x) inv.x = __builtin_inff();
    if (inv.y != inv.y) inv.y = __builtin_inff();
    if (inv.z != inv.z) inv.z = __builtin_inff();
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    float bh = 0, cc = 0;
    while (sp > 0) {
        const int node = stack[--sp];
        const float4 a = reinterpret_cast<const float4*>(tree)[2 * node], b = reinterpret_cast<const float4*>(tree)[2 * node + 1];
        const float t0x = (a.x - ro.x) * inv.x, t0y = (a.y - ro.y) * inv.y, t0z = (a.z - ro.z) * inv.z;
        const float t1x = (b.x - ro.x) * inv.x, t1y = (b.y - ro.y) * inv.y, t1z = (b.z - ro.z) * inv.z;
        const float tMin = um_max(0.0f, um_max(um_max(um_min(t0x, t1x), um_min(t0y, t1y)), um_min(t0z, t1z)));
        const float tMax = um_min(um_min(um_max(t0x, t1x), um_max(t0y, t1y)), um_max(t0z, t1z));
        if (!(tMin < tMax)) continue;
        bh += 1.0f;                                                          // diagnostics.BoundsHitCount++
        const int left = __float_as_int(a.w);
        if (left < 0) cc += (float)(~left);                                  // diagnostics.CandidateCount += entityCount
        else if (sp <= 62) { stack[sp++] = left; stack[sp++] = __float_as_int(b.w); }     // Push(Left); Push(Right)
    }
    return make_float2(bh, cc);
}

// Launch geometry of a variant (template parameter GEO): one workgroup of 1024 lanes per CU (four waves per SIMD: the throughput shape; 512- and
// 256-lane workgroups for launches that own about one pixel per resident lane were built in round 3, measured - a lone wave gains 1.55 x, the machine
// loses 2.6 x - and removed in round 4); bit 2 = 32-bit traversal-stack / candidate codes (scenes of more than 65 535 entities or tree
// nodes; the tree is then read from HBM).
constexpr int kGeoWide = 4;
// bit 3 = pinhole camera (RtowView.lensRadius == 0: the book-cover, 4K and 10 000-sphere configurations, LEGACY/Final Scene (Book 1).asset:15-18): the lens draw, the lens offset and
// the view's `right` / `up` are not compiled in - six launch constants fewer in the scalar registers of kernels that spill two dozen of them into VGPR lanes.  Only the reference-stream
// variants that keep their whole path history in registers have the twin (launchByDiagGeo).
constexpr int kGeoPinhole = 8;
// bit 4 = lanes in a hurry (main loop: HURRY): the twins of the static-sphere kind's generic reference-stream variants that plain and chained launches with a bound on a pixel's rays run;
// batch groups (no bound) keep the variants without the code - its mere presence costs them 2.4 % (profiles/r06x_lanes_in_a_hurry.json)
constexpr int kGeoHurry = 16;
constexpr int geo_block_threads(int) { return kBlockThreads; }

// DIAG: 0 = RayCount only; 1 = the FULL_DIAGNOSTICS counters of this library's own walk; 2 = those, or - when the launch carries the reference's tree
// (RTOW_CONTEXT_REFERENCE_DIAGNOSTICS) - the reference's counts through reference_counts, whose 64-entry stack is a private segment the other variants do without
template <bool ALL_LDS, int KIND, int HW, int DIAG, int NOISE, bool PER_SAMPLE, int GEO = 0>
__global__ void __launch_bounds__(geo_block_threads(GEO)) sample_batch_kernel(const SampleKernelArgs A)
{
    constexpr bool FULL_DIAG = DIAG != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = (int)threadIdx.x;
    constexpr int BT = geo_block_threads(GEO);
    // the tie fix-up launch (SampleKernelArgs.redoMode) has nothing to do almost always: it leaves before it stages the scene
    if (A.redoMode) { if (__hip_atomic_load(A.tieRedo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return; }
    constexpr bool WIDE = (GEO & kGeoWide) != 0;
    constexpr bool PINHOLE = (GEO & kGeoPinhole) != 0;
    // Camera-ray lists hold up to eight leaf-parent nodes with 16-bit codes (one uint4 per pixel, four registers), four with 32-bit codes.  Four covered 95.6 % of the
    // cover scene's pixels (pinhole camera), 93.6 % of the 10 000 spheres', 56.7 % with moving spheres and a lens; eight cover 100 / 99.9 / 88.5 %.  The pixels beyond four
    // are sphere edges in tiles of sky - the minority lanes whose walks wait longest for their stage (profiles/r04n): cover scene +1.1 % (10 batches per launch), +6.6 % as
    // single launches; 10 000 spheres +4.1 %; moving + defocus +13 % (profiles/r04q_camera_ray_lists.json)
    // Sphere kinds only: the general-entity, textured and volume kernels spill already, and two more list registers cost them more than the lists bring
    // (image-textured spheres -13 %, mixed primitives -1.7 % with eight nodes: gpurun_out/r04am); their lists stay at four nodes (the first uint2 of the pixel's record).
    constexpr bool LONG_LISTS = !WIDE && (KIND & 7) <= SCENE_KIND_SPHERES_MOTION;
    constexpr bool LDS_VIEW = (RTOW_LDS_VIEW & (ALL_LDS ? 1 : 2)) != 0;   // ... or read from an LDS copy (bit 0: kernels with the scene in LDS, bit 1: the others)
    constexpr bool COLD_VIEW __attribute__((unused)) = RTOW_COLD_VIEW && !ALL_LDS && !LDS_VIEW;    // the view's and the sky's launch constants are read on use instead of held in scalar registers (REGEN)
    constexpr bool SPLIT_NODES = !ALL_LDS && !WIDE;      // node loads as ds_read / global_load behind a wave-uniform branch instead of flat loads (load_node)
    using Code = typename std::conditional<WIDE, unsigned, unsigned short>::type;
    const uint32_t ldsFront = A.ldsFrontBytes;          // this launch's LDS plan (LdsPlan, rtow_kernels.h): where the wave queues start

    // ---- stage the scene image into LDS: coalesced 16 B per lane ----
    // [level][lane] uint16 arrays; within a wave lane l sits at 2*(l&31) + (l>>5), so the 32 lanes the LDS services together
    // touch 32 different dwords (= banks) whatever level each of them is at
    // (32-bit codes: one dword per lane, the natural order already is conflict free)
    const int swizzled = (tid & ~63) + ((tid & 31) << 1) + ((tid >> 5) & 1);
    Code* const cand = reinterpret_cast<Code*>(smem) + (WIDE ? tid : swizzled);                // [slot][lane] leaf candidates: the first kCandCapacity rows
    Code* const stack = cand + kCandCapacity * BT;                                             // [level][lane]: A.ldsStackRows rows, one per inner level of this scene's tree
    HistRowsT<!ALL_LDS> histRows;
    histRows.lane = HW == 32 ? reinterpret_cast<unsigned short*>(smem + A.ldsHistOffset) + swizzled : nullptr;
    // {next, end} ticket chunk of this wave; chains: {.., needDone, chunk} = the chunk may only be handed out once chunkDone[chunk] >= needDone
    volatile unsigned* const waveQueue = reinterpret_cast<volatile unsigned*>(smem + ldsFront) + (tid >> 6) * 4;
    uint8_t* const ldsScene = smem + ldsFront + kQueueBytes;
    if ((tid & 63) == 0) { waveQueue[0] = 0; waveQueue[1] = 0; waveQueue[2] = 0; waveQueue[3] = 0; }
    // launch constants only REGEN and SKY read (view: 22 floats, sky: 7 dwords, frame size: 2 floats), parked in LDS behind the wave queues (RTOW_LDS_VIEW)
    float* const ldsConst = reinterpret_cast<float*>(smem + ldsFront + 256);
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS_VIEW && tid < 31) {
        // copied dword by dword from the kernarg segment (a struct assignment from the by-value argument goes through a private copy)
        const uint8_t* const ka = (const uint8_t*)__builtin_amdgcn_kernarg_segment_ptr();
        const size_t from = tid < 22 ? __builtin_offsetof(SampleKernelArgs, view) + 4u * (size_t)tid
                          : tid < 29 ? __builtin_offsetof(SampleKernelArgs, environment) + 4u * (size_t)(tid - 22)
                                     : __builtin_offsetof(SampleKernelArgs, sizeX) + 4u * (size_t)(tid - 29);
        ldsConst[tid] = *reinterpret_cast<const float*>(ka + from);
    }
#endif
    {
        const uint4* src = reinterpret_cast<const uint4*>(A.sceneBlob);
        uint4* dst = reinterpret_cast<uint4*>(ldsScene);
        const uint32_t n16 = A.ldsSceneBytes >> 4;
        for (uint32_t i = (uint32_t)tid; i < n16; i += BT) dst[i] = src[i];
    }
    __syncthreads();

    SceneRefs sc;
    sc.lds = ldsScene;
    sc.glob = A.sceneBlob;
    sc.ldsNodeCount = A.ldsNodeCount;
    const SceneLayout L = A.layout;
    const int traceDepth = A.traceDepth;
    const bool refDiag = DIAG == 2 && A.refTree != nullptr;   // BoundsHitCount / CandidateCount count the reference's tree (RTOW_CONTEXT_REFERENCE_DIAGNOSTICS)
    const bool chained = A.chainCount > 1u;      // several successive batches in this launch (wave-uniform): coherent accumulator accesses, per-chunk hand-off
    // a one-entity scene has a root whose second child is a placeholder; its (inverted) box cannot be told from a real one by the
    // symmetric slab test, so it is masked explicitly (wave-uniform, costs one scalar AND per node visit)
    const bool twoChildren = L.sphereCount > 1u;
    // KIND = scene kind (SCENE_KIND_*) | kExactTiesBit: with the bit, a tie at the nearest hit is settled by the reference's whole procedure
    // (resolve_nearest_tie, a call that costs the hot kernel ~10 % whether taken or not - hence a variant of its own, chosen at upload)
    constexpr int BASE = KIND & 7;
    constexpr bool EXACT_TIES = (KIND & kExactTiesBit) != 0;
    constexpr bool HAS_MOTION = BASE == SCENE_KIND_SPHERES_MOTION;
    constexpr bool GENERAL = BASE >= SCENE_KIND_GENERAL;
    constexpr bool VOLUMES = BASE == SCENE_KIND_VOLUMES || BASE == SCENE_KIND_VOLUMES_TEXTURED;   // ProbabilisticVolume materials present: every hit of a ray is needed, not only the nearest
    constexpr bool TEXTURED = BASE == SCENE_KIND_TEXTURED || BASE == SCENE_KIND_VOLUMES_TEXTURED || BASE == SCENE_KIND_TRIANGLES_TEXTURED; // Image textures present: albedo / emission / metallic / glossiness are per hit
    // Sphere kinds under the rank rule: a nearest hit shared by two DIFFERENT spheres is exact for rays of at most 16 hits only, so such a pixel is
    // not stored but handed to the exact-tie kernel of the same kind, which runs a second, tiny launch over the listed pixels (redo)
    // All-triangle scenes (round 6) are watched the same way: their exact-tie kernels carry the resolver's hit lists in scratch and spill twenty registers on top (the
    // 250 882-triangle mesh ran on them: 896 bytes of private segment per lane), and a mesh only ties where a ray meets a shared edge to the last bit - the rank-rule
    // kernel (119 registers, no scratch) traces the frame, the exact-tie kernel the handful of marked pixels.  Scenes that hold the same triangle twice tie everywhere
    // and keep the exact-tie kernels (SceneLayout.tieWatchOk, rtow_bvh.cpp); so does a scene whose first watched launch marks more than a few thousand pixels (rtow_api.hip).
    constexpr bool TRI_KIND = BASE == SCENE_KIND_TRIANGLES || BASE == SCENE_KIND_TRIANGLES_TEXTURED;
    constexpr bool TIE_WATCH = RTOW_TIE_WATCH && (!GENERAL || TRI_KIND) && !EXACT_TIES && !PER_SAMPLE;
    constexpr bool REDO_CAPABLE = (!GENERAL || TRI_KIND) && EXACT_TIES && !PER_SAMPLE;
    // (whether a launch watches / fixes up, and whether its batches are a group, are launch constants read from the kernarg segment where they are used - at pixel
    // boundaries and in the rare fallback store - not values that live in registers through every stage: the kernel sits at the edge of its register file)

    // ---- per-lane persistent state ----
    int st = ST_REGEN;
    int pix = -1;
    unsigned tick = 0;          // ticket (owned-pixel number) of the current pixel
    typename std::conditional<PER_SAMPLE && NOISE == RTOW_NOISE_WHITE, RngPerSample, Rng<NOISE>>::type rng{};
    V3 fbNormal = v3(0, 0, 0), fbAlbedo = v3(0, 0, 0);   // PER_SAMPLE: AOVs of the batch's sample 0 (the fallback when nothing succeeds)
    unsigned unitGroup = 0;                                // PER_SAMPLE: which 16-sample group of its pixel this lane works on
    unsigned smp = 0, nsamp = 0;
    int cx = 0, cy = 0;
    V3 colorAcc = v3(0, 0, 0), normalAcc = v3(0, 0, 0), albedoAcc = v3(0, 0, 0);
    float scwAcc = 0, scw0 = 0;
    int sampleCount = 0;
    float rayCount = 0, boundsHits = 0, candidates = 0;

    // per-path state
    V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1);
    float rtime = 0;
    int depth = 0;
    Hist<HW> hist;
    hist.clear();
    V3 sampleNormal = v3(0, 0, 0);
    // sampleAlbedo (:316-328,366-370) is not carried in registers: it is emission + reflectance of the first hit that is not perfectly specular -
    // which the path history already names (depth firstNs) - or the sky colour, or 0; endSample rebuilds it with the same additions
    int firstNs = -1;
    float randomEventsLocal = 0;

    // VOLUMES only: all hits of the current ray (FindHits' hitRecordBuffer, JOBS/SampleBatchJob.cs:450-475), the volume the path
    // is inside of (currentProbabilisticVolumeMaterial, :180) and RandomEvents left pending by ProbabilisticHit (RT/Material.cs:54)
    // TEXTURED only: what the fold needs of every hit of the current path (the 16-bit history code only names a material)
    float texHist[TEXTURED ? HW * 2 * 6 : 1];
    constexpr int kMaxHits = VOLUMES ? kLocalHits : 1;   // entries of the ray's hit list the lane holds itself; the rest spills (HitSpill)
    float hitT[kMaxHits], hitTmin0[kMaxHits];
    unsigned hitCode[kMaxHits];      // primitive | dot(normal, dir) < 0 -> bit 30 | dot > 0 -> bit 31
    bool tieAtBest = false;          // scenes without volumes: a second surface at exactly the nearest distance was seen (resolve_nearest_tie)
    bool hitOverflow = false;
    int nHits = 0;
    int curVol = -1;
    float pendRE = 0;
    bool insideHit = false;          // the chosen "hit" is a scattering event inside the volume
    float hitTmin = 0;               // tMin of the test that produced the chosen hit (exit hits use entry + 0.001)
    constexpr bool TRIANGLES_ONLY = BASE == SCENE_KIND_TRIANGLES || BASE == SCENE_KIND_TRIANGLES_TEXTURED;   // every entity is a triangle: no type dispatch, no transform code
    constexpr bool TRI_HOT = TRIANGLES_ONLY && RTOW_TRI_HOT;                                                    // ... tested from the compact GpuTriHot records; keptNormal.x / .y then carry the winner's (u, v)
    constexpr bool PREFETCH_FAR = WIDE && !ALL_LDS && (RTOW_PREFETCH & 1) != 0;
    constexpr bool PREFETCH_TRI = WIDE && !ALL_LDS && TRI_HOT && (RTOW_PREFETCH & 2) != 0;
    const uint32_t ldsDump __attribute__((unused)) = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + ((uint32_t)kCandCapacity + A.ldsStackRows) * (uint32_t)BT * 4u);
    constexpr bool KEEP_NORMAL = BASE == SCENE_KIND_GENERAL || BASE == SCENE_KIND_TRIANGLES;   // the winning test's entity-space normal travels from TEST to HIT (else HIT re-runs the test)
    V3 keptNormal = v3(0, 0, 0);

    // per-ray traversal state (resumable across trips)
    V3 inv = v3(0, 0, 0);
    int cur = 0, sp = 0, nc = 0, prim = -1;
    float best = 0;

    // end of a sample (JOBS/SampleBatchJob.cs:137-156)
    auto endSample = [&](bool ok, V3 sampleColor, V3 skyColor) {
        // the sample's albedo AOV: first non-specular hit (emission + reflectance, reflectance overridden to 1 by a specular reflection), else
        // the sky the path ended in, else the default 0 (a path cut off at TraceDepth that only met perfect mirrors / glass)
        V3 sampleAlbedo = ok ? skyColor : v3(0, 0, 0);
        if (firstNs >= 0) {
            const unsigned code = hist.get(firstNs, histRows);
            const bool white = (code & 0x8000u) != 0;
            if (TEXTURED) {
                const V3 refl = white ? v3(1, 1, 1) : v3(texHist[firstNs * 6 + 0], texHist[firstNs * 6 + 1], texHist[firstNs * 6 + 2]);
                sampleAlbedo = add(v3(texHist[firstNs * 6 + 3], texHist[firstNs * 6 + 4], texHist[firstNs * 6 + 5]), refl);
            } else {
                const uint8_t* mp = section<ALL_LDS>(sc, L.materialOffset) + (code & 0x7fffu) * 64u;
                const float4 m0 = *reinterpret_cast<const float4*>(mp);       // albedo.xyz emission.x
                const float2 m1 = *reinterpret_cast<const float2*>(mp + 16);  // emission.yz
                const V3 refl = white ? v3(1, 1, 1) : v3(m0.x, m0.y, m0.z);
                sampleAlbedo = add(v3(m0.w, m1.x, m1.y), refl);
            }
        }
        if (ok) {                                                                         // :145-149, :398
            scwAcc += randomEventsLocal;
            colorAcc = add(colorAcc, sampleColor);
            normalAcc = add(normalAcc, sampleNormal);
            albedoAcc = add(albedoAcc, sampleAlbedo);
            sampleCount++;
        } else if (PER_SAMPLE) {
            if (smp == 0) { fbNormal = sampleNormal; fbAlbedo = sampleAlbedo; }
        } else if (smp == 0 && sampleCount == 0 && !A.probeOnly) {
            // sample 0 failed: its AOVs are the fallback if NO sample of this pixel succeeds (:152-156,160-161).  Stored now and overwritten at the end of the pixel iff
            // sampleCount != 0 - so only where nothing has succeeded yet (earlier batches included: the count came in with the inputs); otherwise the record is overwritten anyway.
            const SampleKernelArgs& R = A;
            float* fbN = R.outNormal;
            float* fbA = R.outAlbedo;
            if (chained && R.chainIndependent) { fbN = R.chainBatches[tick >> kChainShift].outNormal; fbA = R.chainBatches[tick >> kChainShift].outAlbedo; }      // a batch group: this batch's own buffers
            fbN += 3 * (size_t)pix;
            fbA += 3 * (size_t)pix;
            fbN[0] = sampleNormal.x; fbN[1] = sampleNormal.y; fbN[2] = sampleNormal.z;
            fbA[0] = sampleAlbedo.x; fbA[1] = sampleAlbedo.y; fbA[2] = sampleAlbedo.z;
        }
        smp++;
        st = ST_REGEN;
    };
    // a new ray segment starts: reset the traversal state
    auto startRay = [&]() {
        // rayInvDirection = rcp(ray.Direction), NaN -> +INF (JOBS/SampleBatchJob.cs:408-412).  The IEEE quotient, not v_rcp_f32: the leaf
        // children of the tree carry the reference's own entity boxes and must pass or fail the reference's own slab test (RT/HitTests.cs:9-21)
        inv = v3(RTOW_RCP_NAN_TO_INF(rd.x), RTOW_RCP_NAN_TO_INF(rd.y), RTOW_RCP_NAN_TO_INF(rd.z));
        cur = 0; sp = 0; nc = 0; prim = -1;
        best = __builtin_inff();
        tieAtBest = false;
        nHits = 0;
        st = ST_TRAV;
        if (DIAG == 2) { if (refDiag) { const float2 rc = reference_counts(A.refTree, ro, rd); boundsHits += rc.x; candidates += rc.y; } }   // FindHitCandidates(ray, ...) of this segment (:186)
    };
    // traversal finished: classify the result
    auto classify = [&]() {
        rayCount += 1.0f;                                                                  // :203
        if (VOLUMES) { st = ST_VOL; return; }
        st = prim < 0 ? ST_SKY : ST_HIT;
    };

    // this pixel's camera-ray candidate list: leaf-parent node indices, 4 x 16 bit in .x / .y (kNoPrimaryList in .x: none) - with wide codes
    // 4 x 32 bit (0xffffffff = empty slot; first slot empty, second not: no list)
    uint4 pcand = make_uint4(WIDE ? 0xffffffffu : kNoPrimaryList, 0u, 0u, 0u);
    int force = -1;
    constexpr bool HURRY = RTOW_URGENT_LANES && (GEO & kGeoHurry) != 0;      // lanes in a hurry (below): twins of the static-sphere kind's generic reference-stream variants (launchByDiagGeo)
    unsigned trip = 0;          // chained batches only: paces the polls of parked lanes
    STAT_DECL;
    STAGE_DECL;
    for (;;) {
        STAT_ADD(0, 1);
        STAGE_MARK_TOP(7);
        // Stages run in pipeline order; each one only if enough lanes wait in it, so a lane can still advance a whole path segment per
        // trip when the wave is dense, while sparse stages batch up.  A.tune[] holds the thresholds in 64ths of the wave's LIVE lanes
        // (lanes that still have pixels): 1 = "any lane", 48 = three quarters of them.  Depth-0 rays skip the box walk (camera-ray lists),
        // so without a threshold the walk would run every trip for the ~60 % of lanes on a bounce segment; holding it back until
        // most live lanes want it lets the camera segments (REGEN -> TEST -> HIT) of the others catch up first.
        // chained batches: parked lanes (ST_IDLE) ask for their pixel again every 8th trip - a poll is a round trip to memory that the
        // wave's working lanes would otherwise pay on every trip - and do not count as live: the thresholds below are fractions of the
        // lanes that have work, and a stage that cannot make progress must never keep the others from being forced
        if (chained) { trip++; if ((trip & 7u) == 0u && st == ST_IDLE) st = ST_REGEN; }
        const unsigned long long liveMask = __ballot(st != ST_DEAD && st != ST_IDLE);
        const int live = (int)__popcll(liveMask);
        auto need = [&](int sixtyFourths) { const int t = (live * sixtyFourths + 63) >> 6; return t < 1 ? 1 : t; };
        // Lanes in a hurry (round 6).  One generator per pixel and batch: a pixel's samples run in a row on one lane, so a launch is never shorter than its slowest pixel - at the
        // reference host's trace depth 32 a few dozen pixels (rays trapped in glass: nearly every path runs to the depth limit) take 1 100 rays per 50-sample batch where the mean
        // takes 128 - and as plain or chained launches (each batch waits for the one before it) the frame waits for them: 16.5 ms per batch where the batch groups need 11.7.  A lane
        // whose pixel runs at more than c rays per sample - rayCount > c x (samples done + 2) - does not wait for company: the stage it waits in runs, whoever else waits there rides
        // along, and its wave issues ahead of the three it shares a SIMD with (STAGE_PRIO).  Such a pixel is found within two or three samples; a bound on the batch's total
        // (c x spp: the first form) found it after a third of its batch (chains +8 ... 10 % between the two forms).  Scheduling only; batch groups run a pixel's batches side by
        // side and launch the variants without this code.  c shares A.tune[7] with the pixel gate - its float bits above the gate's eight - because one more launch constant held
        // through the loop costs the variants beyond LDS their freedom from a private segment, and reading it on use a scalar-memory wait per trip.
        constexpr unsigned kHurryGrace = 2u;
        unsigned long long hurry = 0ull;                                            // the live lanes in a hurry, as of the top of the trip
        if (HURRY) hurry = liveMask & __ballot(rayCount > __int_as_float(A.tune[7] & ~255) * (float)(smp + kHurryGrace));
        auto due = [&](bool waiting, int stage, int sixtyFourths) {
            const unsigned long long m = __ballot(waiting);
            return (int)__popcll(m) >= (force == stage ? 1 : need(sixtyFourths)) || (HURRY && (m & hurry) != 0ull);
        };
        bool ran = false;
        // Pixel boundaries in company (round 6).  A lane that has finished its pixel (unit) runs two to three hundred instructions - stores, ticket, loads, seed, sample count - that
        // nothing else in its wave takes part in: 1.2 lanes on average (profiles/r05_runs/run_r05u.sh), and the wave issues every one of them.  With A.tune[7] = K > 1 such a lane
        // waits in ST_REGEN until K of the wave's live lanes want a boundary (or nothing else can run): the block then runs once for K lanes.  Scheduling only.
        // Only the variants whose workloads have frequent boundaries carry the code (the generic ones: the reference host's 50 samples per batch at depth 32; the per-sample
        // policies' 16-sample units): its mere presence brought the moving-sphere headline kernel's 36-byte private segment back.
        constexpr bool PIXEL_COMPANY = HW == 32 || PER_SAMPLE;
        bool regenReady = st == ST_REGEN;
        const int pixelGate = HURRY ? A.tune[7] & 255 : A.tune[7];
        if (PIXEL_COMPANY && pixelGate > 1) {
            const int wantPixel = (int)__popcll(__ballot(st == ST_REGEN && smp >= nsamp));
            const int company = live < pixelGate ? live : pixelGate;
            if (wantPixel < company && force != ST_REGEN) regenReady = st == ST_REGEN && (smp < nsamp || (HURRY && ((hurry >> (tid & 63)) & 1ull) != 0ull));   // (a lane in a hurry stores at once)
        }
        if (due(regenReady, ST_REGEN, A.tune[0])) {
            ran = true;
            STAGE_MARK(0);
            // ================= next sample of this pixel, or next pixel =================
            STAT_ADD(1, 1);
            if (regenReady) {
                STAT_LANES(2);
                while (smp >= nsamp) {
                    // Pixel boundaries are rare (one per `spp` samples) and touch two dozen launch constants nothing else needs - buffer pointers,
                    // slice and sample-count parameters.  Read through a laundered pointer to the kernarg segment they are s_load-ed here, on
                    // use, instead of sitting in SGPRs (and, past 102 of them, in VGPR lanes read back with v_readlane) through every stage.
#if defined(__HIP_DEVICE_COMPILE__)
                    const SampleKernelArgs* coldArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();   // the struct is the kernel's only argument
                    asm volatile("" : "+s"(coldArgs));
#else
                    const SampleKernelArgs* coldArgs = &A;                                                                // host pass of the HIP compiler: never executed
#endif
                    const SampleKernelArgs& C = *coldArgs;
                    const bool grouped = chained && C.chainIndependent != 0;             // a batch group: same inputs, own outputs, nothing handed over
                    const bool redo = REDO_CAPABLE && C.redoMode != 0;                   // this launch IS the fix-up launch
                    const unsigned tk = chained ? (tick & kChainTicketMask) : tick;      // owned-pixel (unit) number inside its batch
                    const unsigned batch = chained ? (tick >> kChainShift) : 0u;         // which batch of the chain the finished pixel belongs to
                    if (pix >= 0 && C.pixelCost) {
                        // cost map for the next launch's chunk order: this pixel's ray count, in ticket order (a plain 2-byte store that
                        // merges in L2 with its chunk's other 63; per-chunk atomics cost a memory-side transaction each)
                        const unsigned rc = (unsigned)rayCount;
                        C.pixelCost[tk] = (unsigned short)(rc < 65535u ? rc : 65535u);
                    }
                    if (VOLUMES && hitOverflow) { *C.overflowFlag = 1u; hitOverflow = false; }            // RTOW_ERROR_CAPACITY on the host side
                    if (pix >= 0 && C.probeOnly) pix = -1;                                                 // cost probe: nothing is stored
                    bool redoContinue = false;
                    if (PER_SAMPLE && pix >= 0) {
                        // ---- unit done: its partial sums go to the record the fold kernel adds up in group order ----
                        const bool fallback = unitGroup == 0 && sampleCount == 0;       // then sample 0 failed: the record carries its AOVs instead of sums
                        float4* rec = reinterpret_cast<float4*>(C.unitRecords) + (size_t)tick * 4u;
                        rec[0] = make_float4(colorAcc.x, colorAcc.y, colorAcc.z, (float)sampleCount);
                        rec[1] = fallback ? make_float4(fbNormal.x, fbNormal.y, fbNormal.z, rayCount) : make_float4(normalAcc.x, normalAcc.y, normalAcc.z, rayCount);
                        rec[2] = fallback ? make_float4(fbAlbedo.x, fbAlbedo.y, fbAlbedo.z, scwAcc) : make_float4(albedoAcc.x, albedoAcc.y, albedoAcc.z, scwAcc);
                        rec[3] = make_float4(boundsHits, candidates, 0, 0);
                        pix = -1;
                    }
                    if (pix >= 0 && chained) {
                        // ---- pixel done, chained batches: the same stores (:159-163), then publish the pixel ----
                        float *oc = C.outColor, *on = C.outNormal, *oa = C.outAlbedo, *os = C.outScw;
                        if (grouped) { const ChainBatch cb = C.chainBatches[batch]; oc = cb.outColor; on = cb.outNormal; oa = cb.outAlbedo; os = cb.outScw; }   // a batch group: this batch's own outputs
                        reinterpret_cast<float4*>(oc)[pix] = make_float4(colorAcc.x, colorAcc.y, colorAcc.z, (float)sampleCount);
                        if (sampleCount != 0 || nsamp == 0) {
                            const V3 nrm = sampleCount != 0 ? normalAcc : v3(0, 0, 0), alb = sampleCount != 0 ? albedoAcc : v3(0, 0, 0);
                            on[3 * (size_t)pix + 0] = nrm.x; on[3 * (size_t)pix + 1] = nrm.y; on[3 * (size_t)pix + 2] = nrm.z;
                            oa[3 * (size_t)pix + 0] = alb.x; oa[3 * (size_t)pix + 1] = alb.y; oa[3 * (size_t)pix + 2] = alb.z;
                        }
                        os[pix] = scwAcc;
                        uint8_t* dg = C.chainBatches[batch].diagnostics;
                        if (dg) {
                            if (FULL_DIAG && C.diagnosticsStride >= 16)
                                *reinterpret_cast<float4*>(dg + (size_t)pix * 16u) = make_float4(rayCount, boundsHits, candidates, scw0);
                            else
                                *reinterpret_cast<float*>(dg + (size_t)pix * 4u) = rayCount;
                        }
                        // every store above has reached this XCD's L2 (s_waitcnt vmcnt(0); the workgroup-scope release keeps the compiler from reordering)
                        // before the chunk's counter moves; the chunk's next batch runs on this XCD too and reads through that L2
                        // (a batch group hands nothing over: its stores are a plain batch's)
                        if (!grouped && !redo) {
                            coherent_flush();
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __hip_atomic_fetch_add(C.chunkDone + (tk >> 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        // the fix-up launch carries a listed pixel through the REST of its chain: what was just stored is the next batch's input, and it is still in registers
                        if (REDO_CAPABLE) { if (redo && !grouped && batch + 1u < C.chainCount) { redoContinue = true; tick += 1u << kChainShift; } }
                        if (!redoContinue) pix = -1;
                    }
                    if (pix >= 0 && !redoContinue) {
                        // ---- pixel done: store (JOBS/SampleBatchJob.cs:159-163) ----
                        reinterpret_cast<float4*>(C.outColor)[pix] = make_float4(colorAcc.x, colorAcc.y, colorAcc.z, (float)sampleCount);
                        if (sampleCount != 0) {
                            C.outNormal[3 * (size_t)pix + 0] = normalAcc.x; C.outNormal[3 * (size_t)pix + 1] = normalAcc.y; C.outNormal[3 * (size_t)pix + 2] = normalAcc.z;
                            C.outAlbedo[3 * (size_t)pix + 0] = albedoAcc.x; C.outAlbedo[3 * (size_t)pix + 1] = albedoAcc.y; C.outAlbedo[3 * (size_t)pix + 2] = albedoAcc.z;
                        } else if (nsamp == 0) {
                            // no sample ran: the fallbacks keep their default (0) value (:115)
                            C.outNormal[3 * (size_t)pix + 0] = 0; C.outNormal[3 * (size_t)pix + 1] = 0; C.outNormal[3 * (size_t)pix + 2] = 0;
                            C.outAlbedo[3 * (size_t)pix + 0] = 0; C.outAlbedo[3 * (size_t)pix + 1] = 0; C.outAlbedo[3 * (size_t)pix + 2] = 0;
                        } // else: sample 0 failed and its AOVs were stored as the fallback by endSample
                        C.outScw[pix] = scwAcc;
                        if (C.diagnostics) {
                            if (FULL_DIAG && C.diagnosticsStride >= 16)
                                *reinterpret_cast<float4*>(C.diagnostics + (size_t)pix * 16u) = make_float4(rayCount, boundsHits, candidates, scw0);
                            else
                                *reinterpret_cast<float*>(C.diagnostics + (size_t)pix * 4u) = rayCount;
                        }
                        pix = -1;
                    }
                    // ---- pull the next owned pixel ----
                    // Tickets are handed to WAVES in chunks of 64 consecutive pixels (one global atomic per chunk) and to lanes
                    // from the wave's chunk by ballot rank, so every 64-byte line of the accumulator arrays is read and written
                    // by a single CU within about one pixel-time and coalesces in that XCD's L2 instead of being fetched and
                    // written back once per pixel from eight different L2s.
                    unsigned newBatch = 0u;
                    if (REDO_CAPABLE && redoContinue) {
                        // the fix-up launch, same pixel, next batch of its chain: colorAcc / normalAcc / albedoAcc / scwAcc / sampleCount hold what was just stored = this batch's inputs
                        newBatch = tick >> kChainShift;
                        if (sampleCount == 0) {
                            // nothing has succeeded in this pixel so far: the AOV inputs of the next batch are what was stored - zeros, or the fallback of a failed sample 0,
                            // which went to the output record directly (endSample) and is not in registers
                            normalAcc = v3(C.outNormal[3 * (size_t)pix], C.outNormal[3 * (size_t)pix + 1], C.outNormal[3 * (size_t)pix + 2]);
                            albedoAcc = v3(C.outAlbedo[3 * (size_t)pix], C.outAlbedo[3 * (size_t)pix + 1], C.outAlbedo[3 * (size_t)pix + 2]);
                        }
                    } else {
                    unsigned ticket = 0xffffffffu;
                    bool parked = false;
                    for (bool got = false; !got;) {
                        const unsigned long long need = __ballot(1);                  // lanes asking right now (all still in this loop)
                        const int lane = tid & 63;
                        const int leader = __builtin_ctzll(need);
                        const int rank = __popcll(need & ((1ull << lane) - 1ull));
                        const unsigned next = waveQueue[0], end = waveQueue[1];      // wave-private: same value in every lane
                        if (next == 0xffffffffu) break;                               // queue exhausted (or cancelled)
                        if (next == end) {
                            if (chained && waveQueue[2] == 0xffffffffu) {
                                // the leader found the chain's hand-over from batch 0 to the per-XCD lists not complete yet (below): everybody asks again later
                                if (lane == leader) waveQueue[2] = 0u;
                                parked = true;
                                break;
                            }
                            if (redo) {
                                // the fix-up launch: tickets are places in the list the first launch filled, 64 at a time
                                if (lane == leader) {
                                    bool cancelled = false;
                                    if (C.cancelFlag) cancelled = *C.cancelFlag != 0u;
                                    const unsigned listed = __hip_atomic_load(C.tieRedo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    const unsigned n = listed < C.tieRedoCapacity ? listed : C.tieRedoCapacity;
                                    const unsigned slot = cancelled ? 0xffffffffu : atomicAdd(C.workCounter, 1u);
                                    if (cancelled || slot >= (n + 63u) / 64u) { waveQueue[0] = 0xffffffffu; waveQueue[1] = 0xffffffffu; }
                                    else { waveQueue[2] = 0u; waveQueue[3] = 0u; waveQueue[0] = slot * 64u; waveQueue[1] = n < slot * 64u + 64u ? n : slot * 64u + 64u; }
                                }
                                continue;
                            }
                            if (grouped) {
                                // a batch group: the queue holds (chunk, batch) slots, every batch of the most expensive chunk first - no batch waits for another.  A wave reserves
                                // slotBlock slots at a time and its tickets are slot << 6 | pixel of the chunk: the range below spans the slots, and a lane finds chunk and batch from
                                // its ticket (below) - so the pixels a wave holds next to each other are the same tile's under other seeds, or its neighbours' in the cost order:
                                // like work (profiles/r05x_launch_constants.json "queue slots": +1.4 ... 1.8 % on groups; chains and plain launches keep one chunk per pull)
                                if (lane == leader) {
                                    bool cancelled = false;
                                    if (C.cancelFlag) cancelled = *C.cancelFlag != 0u;
                                    const unsigned slots = C.chunkCount * C.chainCount;
                                    const unsigned first = cancelled ? 0xffffffffu : atomicAdd(C.workCounter, C.slotBlock);
                                    if (first >= slots) { waveQueue[0] = 0xffffffffu; waveQueue[1] = 0xffffffffu; }
                                    else { waveQueue[2] = 0u; waveQueue[0] = first << 6; waveQueue[1] = (first + C.slotBlock < slots ? first + C.slotBlock : slots) << 6; }
                                }
                                continue;
                            }
                            if (lane == leader) {
                                bool cancelled = false;
                                if (C.cancelFlag) cancelled = *C.cancelFlag != 0u;
                                // most expensive chunks first (cost map of the previous launch, or of a 1-spp probe), so that the
                                // chunks handed out last - the ones that decide when a wave can retire - are the cheap ones
                                unsigned slot = cancelled ? 0xffffffffu : atomicAdd(C.workCounter, 1u);
                                unsigned b = 0u, chunk = 0u;
                                bool exhausted = slot >= C.chunkCount, notReady = false;
                                if (!exhausted) chunk = C.chunkOrder ? C.chunkOrder[slot] : slot;
                                if (chained && !grouped && !cancelled) {
                                    // Batch 0 of every chunk comes from the one device-wide queue above; the XCD whose wave takes it owns the chunk for
                                    // the rest of the chain (its accumulator lines then live in that XCD's L2: see the note on chained batches).
                                    // Batches 1 .. chainCount - 1 are handed out per XCD, batch after batch over the XCD's own list in the order it
                                    // was filled (still most expensive first), once every chunk has an owner.
                                    XcdState* const xs = C.xcdState;
                                    unsigned* const list = reinterpret_cast<unsigned*>(xs + 1) + (size_t)xcc_id() * C.chunkCount;
                                    unsigned* const owned = &xs->owned[xcc_id()];
                                    if (!exhausted) {
                                        const unsigned k = __hip_atomic_fetch_add(owned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        __hip_atomic_store(list + k, chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        coherent_flush();                            // the entry is written before it is counted as listed
                                        __hip_atomic_fetch_add(&xs->listed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    } else if (__hip_atomic_load(&xs->listed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < C.chunkCount) {
                                        notReady = true;                             // a wave holds a batch-0 slot it has not listed yet (nanoseconds)
                                    } else {
                                        const unsigned mine = __hip_atomic_load(owned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        const unsigned t = mine ? __hip_atomic_fetch_add(&xs->ticket[xcc_id()], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                                        if (mine != 0u && t < mine * (C.chainCount - 1u)) {
                                            b = 1u + t / mine;
                                            chunk = __hip_atomic_load(list + (t - (b - 1u) * mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            exhausted = chunk >= C.chunkCount;       // 0xffffffff = an entry that was never written: cannot happen once `listed` is complete - and if it
                                                                                     // ever did, the ticket is dropped (the host sees a batch that did not finish its pixels) rather than read as a chunk
                                        }
                                    }
                                }
                                if (notReady) { waveQueue[2] = 0xffffffffu; }
                                else if (exhausted) { waveQueue[0] = 0xffffffffu; waveQueue[1] = 0xffffffffu; }
                                else {
                                    const unsigned base = chunk * 64u;
                                    const unsigned last = (C.totalWork - base < 64u) ? C.totalWork : base + 64u;
                                    waveQueue[2] = b * (last - base);               // pixels of this chunk that must be stored before batch b may read them
                                    waveQueue[3] = chunk;
                                    waveQueue[0] = base | (b << kChainShift);
                                    waveQueue[1] = last | (b << kChainShift);
                                }
                            }
                            continue;
                        }
                        if (chained) {
                            // batch b of a chunk reads what batch b - 1 of the same chunk stored, possibly on another CU / XCD, possibly in other
                            // lanes of this very wave: never spin here - lanes that cannot be served leave and ask again on the next trip
                            if (waveQueue[2] != 0u) {
                                if (lane == leader && __hip_atomic_load(C.chunkDone + waveQueue[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= waveQueue[2]) waveQueue[2] = 0u;
                                if (waveQueue[2] != 0u) { parked = true; break; }
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            }
                        }
                        const unsigned take = (unsigned)__popcll(need) < end - next ? (unsigned)__popcll(need) : end - next;
                        if ((unsigned)rank < take) { ticket = next + (unsigned)rank; got = true; }
                        if (lane == leader) waveQueue[0] = next + take;
                    }
                    if (ticket == 0xffffffffu) { st = parked ? ST_IDLE : ST_DEAD; break; }
                    if (REDO_CAPABLE) { if (redo) ticket = C.tieRedo[4u + ticket]; }          // the listed entry: batch << 27 | FRAME pixel index (a chain's pixels are listed with batch 0 and carried through)
                    if (grouped && !redo) {
                        // a group's ticket = slot << 6 | pixel of the chunk, slot = place in the order * chainCount + batch (see above)
                        const unsigned slot = ticket >> 6;
                        const unsigned place = __umulhi(slot, C.groupRecip);            // slot / chainCount
                        const unsigned chunk = C.chunkOrder ? C.chunkOrder[place] : place;
                        ticket = (chunk << 6) | (ticket & 63u);
                        if (ticket >= C.totalWork) { pix = -1; nsamp = 0; smp = 0; continue; }      // the last chunk's tail: no pixel behind this ticket - ask again
                        ticket |= (slot - place * C.chainCount) << kChainShift;
                    }
                    tick = ticket;
                    newBatch = chained ? (ticket >> kChainShift) : 0u;
                    if (chained) ticket &= kChainTicketMask;
                    if (PER_SAMPLE) { unitGroup = ticket % C.groupsPerPixel; ticket = ticket / C.groupsPerPixel; }   // unit = (owned pixel, sample group)
                    // which pixel a ticket stands for: by default the ticket's place in its 8 x 8 tile; with a map, the pixels of a super-tile of tiles sorted by the ray count
                    // of the previous launch and dealt out 64 at a time (SampleKernelArgs.ticketMap), so that a wave's lanes hold pixels of like cost.  `tick` stays the ticket:
                    // the chunk's hand-over counter and the cost map are per ticket
                    else if (!(REDO_CAPABLE && redo)) { const unsigned* const map = C.ticketMap; if (map && ticket < C.tiledPixels) ticket = map[ticket]; }
                    int ownedRow;
                    owned_pixel_xy(ticket, (unsigned)C.width, C.tilesPerRow, C.tiledPixels, cx, ownedRow);   // a chunk's 64 tickets: an 8 x 8 tile of the owned pixels (rtow_kernels.h), or a strip
                    cy = C.sliceOffset + ownedRow * C.sliceDivider;      // rows with row % SliceDivider == SliceOffset (:69-70)
                    if (REDO_CAPABLE) { if (redo) { cy = (int)(ticket / (unsigned)C.width); cx = (int)(ticket - (unsigned)cy * (unsigned)C.width); } }      // (the fix-up launch lists frame pixels)
                    pix = cy * C.width + cx;

                    float4 last = make_float4(0, 0, 0, 0);
                    if (PER_SAMPLE) {
                        // a unit only needs what decides the pixel's sample count (:118-126); the fold kernel reads the accumulators
                        last.w = C.inColor[4 * (size_t)pix + 3];
                        scwAcc = C.inScw[pix];
                    } else if (chained && !grouped && !redo) {
                        // batch 0 reads the launch's inputs, every later batch what the batch before it stored for this pixel (device-coherent loads)
                        const float* ic = (newBatch == 0u ? C.inColor : C.outColor) + 4 * (size_t)pix;
                        const float* in_ = (newBatch == 0u ? C.inNormal : C.outNormal) + 3 * (size_t)pix;
                        const float* ia = (newBatch == 0u ? C.inAlbedo : C.outAlbedo) + 3 * (size_t)pix;
                        last = coherent_load4(ic);
                        normalAcc = coherent_load3(in_);
                        albedoAcc = coherent_load3(ia);
                        scwAcc = coherent_load((newBatch == 0u ? C.inScw : C.outScw) + pix);
                    } else if (!C.probeOnly) {
                        last = reinterpret_cast<const float4*>(C.inColor)[pix];                           // :72-78
                        normalAcc = v3(C.inNormal[3 * (size_t)pix], C.inNormal[3 * (size_t)pix + 1], C.inNormal[3 * (size_t)pix + 2]);
                        albedoAcc = v3(C.inAlbedo[3 * (size_t)pix], C.inAlbedo[3 * (size_t)pix + 1], C.inAlbedo[3 * (size_t)pix + 2]);
                        scwAcc = C.inScw[pix];
                    }
                    colorAcc = v3(last.x, last.y, last.z);
                    sampleCount = (int)last.w;
                    }      // (not a continued pixel of the fix-up launch)
                    const float scwIn = scwAcc;
                    const int countIn = sampleCount;

                    // :91  new Random((Seed * 0x8C4CA03Fu) ^ (uint)(index * 0x7383ED49u)); the ctor discards one NextState()
                    rng.begin_pixel(NoiseSite{&A, (unsigned)cx, (unsigned)cy}, (unsigned)pix, chained ? C.chainBatches[newBatch].seed : C.seed);

                    // :118-126
                    const float w = scwIn / (float)countIn;
                    if (w == 0) {
                        nsamp = C.sampleCountMin;
                    } else {
                        const float nw = um_saturate((w - C.extremaX) / (C.extremaY - C.extremaX));
                        const float lo = (float)C.sampleCountMin, hi = (float)C.sampleCountMax;
                        nsamp = (unsigned)__builtin_rintf(lo + nw * (hi - lo));
                    }
                    if (C.probeOnly) nsamp = (unsigned)C.probeOnly;                                     // probes: 1 sample (cost map) or a few (threshold tuning), nothing stored
                    scw0 = w;
                    smp = 0;
                    if (PER_SAMPLE) {
                        // this unit: samples [16 g, 16 g + 16) of the pixel's nsamp, accumulated from zero
                        smp = unitGroup * kSampleGroup;
                        nsamp = nsamp < smp + kSampleGroup ? nsamp : smp + kSampleGroup;
                        if (nsamp < smp) nsamp = smp;
                        colorAcc = v3(0, 0, 0); normalAcc = v3(0, 0, 0); albedoAcc = v3(0, 0, 0);
                        sampleCount = 0;
                        scwAcc = 0;
                    }
                    rayCount = 0; boundsHits = 0; candidates = 0;
                    // the pixel's camera-ray candidates (primary_candidates_kernel): up to 4 primitive indices, 0xFFFF = none
                    if (WIDE || LONG_LISTS) pcand = C.pixelCandidates ? reinterpret_cast<const uint4*>(C.pixelCandidates)[pix] : make_uint4(WIDE ? 0xffffffffu : kNoPrimaryList, 0u, 0u, 0u);
                    else { const uint2 pc = C.pixelCandidates ? C.pixelCandidates[2 * (size_t)pix] : make_uint2(kNoPrimaryList, 0u); pcand = make_uint4(pc.x, pc.y, 0u, 0u); }
                }
                if (st == ST_REGEN) {
                    // ---- camera ray (:134-135, RT/View.cs:38-48) ----
                    // The view's nineteen constants, the frame size and (SKY) the sky's seven live in scalar registers through every stage if nothing is done - a kernel has 102
                    // and spills the rest into VGPR lanes, read back with v_readlane in the walk and the exact tests (25 - 33 spilled before).  Three places were measured, same
                    // box, three alternating runs each (profiles/r05x_launch_constants.json): registers; read on use through a laundered pointer to the kernarg segment (COLD_VIEW:
                    // one scalar-cache round trip per REGEN / SKY run); read on use from a 128-byte LDS copy behind the wave queues (LDS_VIEW).  Kernels whose tree does not fit
                    // LDS - their waves wait for memory anyway - gain from both: 10 000 spheres +1.2 % (kernarg) / +2.0 % (LDS), 250 882 triangles +2.2 / +2.4 %: they read the LDS
                    // copy.  Kernels with the scene in LDS LOSE with both (cover -1.1 / -1.7 %, depth 32 groups -2.8 %: their REGEN and SKY runs are short and the round trip
                    // shows): they keep the registers.  (The cubemap's nine constants are read on use in every kernel: cubemap_sample.)
                    const SampleKernelArgs* viewArgs = &A;
#if defined(__HIP_DEVICE_COMPILE__)
                    if (COLD_VIEW) { viewArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(viewArgs)); }
#endif
                    const SampleKernelArgs& VA = *viewArgs;
                    const RtowView& VW = LDS_VIEW ? *reinterpret_cast<const RtowView*>(ldsConst) : VA.view;
                    const float frameX = LDS_VIEW ? ldsConst[29] : VA.sizeX, frameY = LDS_VIEW ? ldsConst[30] : VA.sizeY;
                    const V3 viewRight = PINHOLE ? v3(0, 0, 0) : v3(VW.right), viewUp = PINHOLE ? v3(0, 0, 0) : v3(VW.up);
                    const V3 viewLLC = v3(VW.lowerLeftCorner), viewH = v3(VW.horizontal), viewV = v3(VW.vertical);
                    const float lensRadius = PINHOLE ? 0.0f : VW.lensRadius;
                    float jx = 0.5f, jy = 0.5f;
                    const NoiseSite at{&A, (unsigned)cx, (unsigned)cy};
                    if (PER_SAMPLE) rng.begin_sample(at, (unsigned)pix, smp);
                    if (A.subPixelJitter) rng.next2(at, jx, jy);
                    const float u = ((float)cx + jx) / frameX;
                    const float v = ((float)cy + jy) / frameY;
                    float rdx = 0, rdy = 0;
                    if (lensRadius != 0) {
                        float dx, dy;
                        rng.in_unit_disk(at, dx, dy);                                         // RandomSource.InUnitDisk (RT/RandomSource.cs:40-61)
                        rdx = lensRadius * dx;
                        rdy = lensRadius * dy;
                    }
                    const V3 offset = v3(viewRight.x * rdx + viewUp.x * rdy, viewRight.y * rdx + viewUp.y * rdy, viewRight.z * rdx + viewUp.z * rdy);
                    ro = add(v3(VW.origin), offset);
                    rd = normalize(v3(viewLLC.x - offset.x + u * viewH.x + v * viewV.x,
                                      viewLLC.y - offset.y + u * viewH.y + v * viewV.y,
                                      viewLLC.z - offset.z + u * viewH.z + v * viewV.z));
                    rtime = rng.next(at);
                    // moving-sphere scenes whose entities share one TimeRange: carry clamp(unlerp(t0, t1, Time), 0, 1) instead of Time (see sphere_at);
                    // only sphere_at reads it in this scene kind, and scattered rays inherit the ray's time unchanged (RT/Material.cs:94,102,107,153)
                    if (HAS_MOTION) { if (L.commonTimeRange) rtime = um_max(0.0f, um_min(1.0f, (rtime - L.commonT0) / (L.commonT1 - L.commonT0))); }

                    depth = 0;
                    hist.clear();
                    sampleNormal = v3(0, 0, 0);
                    firstNs = -1;
                    randomEventsLocal = 0;
                    curVol = -1;
                    pendRE = 0;
                    startRay();
                    if (WIDE ? !(pcand.x == 0xffffffffu && pcand.y != 0xffffffffu) : pcand.x != kNoPrimaryList) {
                        // Every camera ray of this pixel can only hit primitives under the leaf-parent nodes of the pixel's list (at most eight; four with 32-bit
                        // codes): instead of walking the tree, visit just those nodes - with the walk's own slab test
                        // of this very ray against their leaf boxes (same expressions, so the same candidates the walk would find: a ray that misses a leaf's
                        // box must not reach that leaf's exact test, whose rounding can report a hit for a far, small sphere it passes closely).
                        const f2 invx = {inv.x, inv.x}, invy = {inv.y, inv.y}, invz = {inv.z, inv.z};
                        const f2 ox = {ro.x, ro.x}, oy = {ro.y, ro.y}, oz = {ro.z, ro.z};
                        cur = -1;
                        for (int k = 0; k < 4; k++) {
                            const unsigned node = WIDE ? (k == 0 ? pcand.x : k == 1 ? pcand.y : k == 2 ? pcand.z : pcand.w)
                                                       : ((k < 2 ? pcand.x >> (16 * k) : pcand.y >> (16 * (k - 2))) & 0xffffu);
                            if (node == (WIDE ? 0xffffffffu : 0xffffu)) break;
                            float4 q0, q1, q2;
                            int c0, c1;
                            load_node<ALL_LDS, SPLIT_NODES>(sc, L, (int)node, q0, q1, q2, c0, c1);
                            const f2 tlx = (f2{q0.x, q0.y} - ox) * invx, thx = (f2{q1.z, q1.w} - ox) * invx;
                            const f2 tly = (f2{q0.z, q0.w} - oy) * invy, thy = (f2{q2.x, q2.y} - oy) * invy;
                            const f2 tlz = (f2{q1.x, q1.y} - oz) * invz, thz = (f2{q2.z, q2.w} - oz) * invz;
                            const float tmin0 = vmax3(vmin(tlx.x, thx.x), vmin(tly.x, thy.x), vmax(vmin(tlz.x, thz.x), 0.0f));
                            const float tfar0 = vmin3(vmax(tlx.x, thx.x), vmax(tly.x, thy.x), vmax(tlz.x, thz.x));
                            const float tmin1 = vmax3(vmin(tlx.y, thx.y), vmin(tly.y, thy.y), vmax(vmin(tlz.y, thz.y), 0.0f));
                            const float tfar1 = vmin3(vmax(tlx.y, thx.y), vmax(tly.y, thy.y), vmax(tlz.y, thz.y));
                            const bool leaf0 = c0 < 0 && tmin0 < tfar0;                        // AxisAlignedBoundingBox.Hit on the entity's own box
                            const bool leaf1 = c1 < 0 && tmin1 < tfar1 && twoChildren;
                            if (FULL_DIAG && !refDiag) boundsHits += (leaf0 ? 1.0f : 0.0f) + (leaf1 ? 1.0f : 0.0f);
                            cand[nc * BT] = (Code)~c0;
                            nc += leaf0 ? 1 : 0;
                            cand[nc * BT] = (Code)~c1;
                            nc += leaf1 ? 1 : 0;
                        }
                        // nodes five to eight: the exact-test stage comes back for them when these candidates are done (cur = -2 - k: node k is next)
                        if (LONG_LISTS) { if ((pcand.z & 0xffffu) != 0xffffu) cur = -6; }
                        if (nc == 0 && cur == -1) classify(); else st = ST_TEST;
                    }
                }
            }
        }
        if (due(st == ST_TRAV, ST_TRAV, A.tune[1])) {
            ran = true;
            STAGE_MARK(1);
            // ================= box walk: FindHitCandidates (JOBS/SampleBatchJob.cs:403-448), resumable =================
            if (st == ST_TRAV) {
                const f2 invx = {inv.x, inv.x}, invy = {inv.y, inv.y}, invz = {inv.z, inv.z};
                const f2 ox = {ro.x, ro.x}, oy = {ro.y, ro.y}, oz = {ro.z, ro.z};
                int budget = A.travSlice;
                // Pruning bound.  A subtree whose box the ray enters beyond the nearest hit so far cannot hold a nearer one - but it can hold an
                // EQUAL one: the twin of the sphere that set `best`.  The two distances come from different float programs (slab entry of the
                // twin's box against the quadratic's root), and where the ray meets the sphere at a point that touches its box they differ by
                // rounding only; one ulp the wrong way pruned the twin, TEST never saw the tie, and the resolver was never asked (twin spheres
                // moving, 1 pixel in 42.8 M rays: tests/soak_frames.py at 2.5x).  So the walk prunes with 2^-12 of slack - in every variant: one multiply per slice, and
                // without duplicates the same margin keeps a root that errs a few ulps below its own box entry from being dropped
                // (the entry distance is good to an ulp, the root to a few where the two can meet); TEST still compares against `best` itself.
                const float bestPrune = best * 1.000244140625f;
                // Branch-free node visit: every LDS access of the iteration is issued up front (node, plus the stack slot a
                // pop would need), candidate / stack slots are written unconditionally and only the counters are predicated,
                // so the wave's EXEC mask changes only at the loop test.
                // A walk hands its candidates to TEST as soon as it holds A.tune[6] of them (3; at most 7, the list holds 8 and a visit adds up to 2): the
                // exact tests are deferred to keep the walk branch-free, but the nearest hit among the first few candidates - the walk visits near
                // children first - prunes most of what is left of the walk, which a list filled to the brim (round 1 - 3: 7) never got to use.
                // 1 / 2 / 3 / 4 / 7 candidates: cover 9.18 / 8.87 / 9.52 / 9.41 / 9.18 Gsamples/s, 10 000 spheres 7.37 / 7.42 / 7.85 / 7.64 / 7.37,
                // 250 k-triangle mesh 1.84 / 1.90 / 1.98 / 1.96 / 1.84 (gpurun_out/r03bb; "1" = the 7 of before in that run's encoding)
                // (volume scenes keep every hit of a ray and prune nothing: their lists fill up as before)
                const int candExit = VOLUMES ? kCandCapacity - 2 : (A.tune[6] < kCandCapacity - 1 ? A.tune[6] : kCandCapacity - 1) - 1;
                while (budget > 0 && cur >= 0 && nc <= candExit) {
                    budget--;
                    STAT_ADD(3, (threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) ? 1 : 0);
                    STAT_LANES(4);
                    float4 q0, q1, q2;
                    int c0, c1;
                    load_node<ALL_LDS, SPLIT_NODES>(sc, L, cur, q0, q1, q2, c0, c1);
                    const int spm1 = sp > 0 ? sp - 1 : 0;
                    const int popped = (int)stack[spm1 * BT];
                    // q0 = (lo0.x lo1.x lo0.y lo1.y)  q1 = (lo0.z lo1.z hi0.x hi1.x)  q2 = (hi0.y hi1.y hi0.z hi1.z): pairs = (child0, child1)
                    const f2 tlx = (f2{q0.x, q0.y} - ox) * invx, thx = (f2{q1.z, q1.w} - ox) * invx;
                    const f2 tly = (f2{q0.z, q0.w} - oy) * invy, thy = (f2{q2.x, q2.y} - oy) * invy;
                    const f2 tlz = (f2{q1.x, q1.y} - oz) * invz, thz = (f2{q2.z, q2.w} - oz) * invz;
                    const float tmin0 = vmax3(vmin(tlx.x, thx.x), vmin(tly.x, thy.x), vmax(vmin(tlz.x, thz.x), 0.0f));
                    const float tfar0 = vmin3(vmax(tlx.x, thx.x), vmax(tly.x, thy.x), vmax(tlz.x, thz.x));
                    const float tmin1 = vmax3(vmin(tlx.y, thx.y), vmin(tly.y, thy.y), vmax(vmin(tlz.y, thz.y), 0.0f));
                    const float tfar1 = vmin3(vmax(tlx.y, thx.y), vmax(tly.y, thy.y), vmax(tlz.y, thz.y));
                    // inner child (padded box): conservative, pruned by the nearest hit so far.  Leaf child (the reference's own entity box):
                    // AxisAlignedBoundingBox.Hit itself, tMin < tMax (RT/HitTests.cs:15-20) - pruning by `best` on top (<=, so that a tie at
                    // exactly `best` is still tested) cannot change which hit is nearest.
                    const bool hit0 = tmin0 <= vmin(tfar0, bestPrune);
                    const bool hit1 = tmin1 <= vmin(tfar1, bestPrune) && twoChildren;
                    const bool leaf0 = c0 < 0 && hit0 && tmin0 < tfar0, leaf1 = c1 < 0 && hit1 && tmin1 < tfar1;
                    if (FULL_DIAG && !refDiag) boundsHits += ((c0 < 0 ? leaf0 : hit0) ? 1.0f : 0.0f) + ((c1 < 0 ? leaf1 : hit1) ? 1.0f : 0.0f);
                    cand[nc * BT] = (Code)~c0;
                    nc += leaf0 ? 1 : 0;
                    cand[nc * BT] = (Code)~c1;
                    nc += leaf1 ? 1 : 0;
                    if (PREFETCH_TRI) {
                        // the listed triangle's record is on its way while the walk goes on; both leaves of one parent are neighbours in leaf order
                        if (leaf0) prefetch_sector(sc.glob, L.triHotOffset + (uint32_t)~c0 * (uint32_t)sizeof(GpuTriHot), ldsDump);
                        if (leaf1) prefetch_sector(sc.glob, L.triHotOffset + (uint32_t)~c1 * (uint32_t)sizeof(GpuTriHot), ldsDump);
                    }
                    const bool in0 = hit0 && c0 >= 0;
                    const bool in1 = hit1 && c1 >= 0;
                    const bool both = in0 && in1;
                    const bool swap = tmin1 < tmin0;                         // near child first
                    if (PREFETCH_FAR) {
                        // every pushed node is visited later (nothing is dropped at the pop): ask for it now
                        const uint32_t farNode = (uint32_t)(swap ? c0 : c1);
                        if (both && farNode >= sc.ldsNodeCount) prefetch_sector(sc.glob, L.nodeOffset + farNode * 64u, ldsDump);
                    }
                    stack[sp * BT] = (Code)(swap ? c0 : c1);
                    const int next = both ? (swap ? c1 : c0) : (in0 ? c0 : c1);
                    const bool any = in0 || in1;
                    cur = any ? next : (sp > 0 ? popped : -1);
                    sp = any ? sp + (both ? 1 : 0) : spm1;
                }
                if (cur < 0 || nc > candExit) {
                    if (nc > 0) st = ST_TEST;      // exact tests pending (walk finished, or the list is full)
                    else classify();               // walk finished with nothing left to test
                }
            }
        }
        if (due(st == ST_TEST, ST_TEST, A.tune[2])) {
            ran = true;
            STAGE_MARK(2);
            // ================= exact sphere tests: FindHits (JOBS/SampleBatchJob.cs:450-475) =================
            if (st == ST_TEST) {
                const float a = dot(rd, rd);
                for (;;) {
                if (LONG_LISTS) {
                    if (cur <= -2) {
                        // ---- the rest of a camera ray's long list: as many nodes at a time as the candidate list has room for (see REGEN) ----
                        const f2 invx = {inv.x, inv.x}, invy = {inv.y, inv.y}, invz = {inv.z, inv.z};
                        const f2 ox = {ro.x, ro.x}, oy = {ro.y, ro.y}, oz = {ro.z, ro.z};
                        int k = -2 - cur;
                        cur = -1;
                        for (; k < 8; k++) {
                            if (nc > kCandCapacity - 2) { cur = -2 - k; break; }                    // no room for two more: test what is here, then come back
                            const unsigned word = k < 6 ? pcand.z : pcand.w;
                            const unsigned node = (word >> (16 * (k & 1))) & 0xffffu;
                            if (node == 0xffffu) break;
                            float4 q0, q1, q2;
                            int c0, c1;
                            load_node<ALL_LDS, SPLIT_NODES>(sc, L, (int)node, q0, q1, q2, c0, c1);
                            const f2 tlx = (f2{q0.x, q0.y} - ox) * invx, thx = (f2{q1.z, q1.w} - ox) * invx;
                            const f2 tly = (f2{q0.z, q0.w} - oy) * invy, thy = (f2{q2.x, q2.y} - oy) * invy;
                            const f2 tlz = (f2{q1.x, q1.y} - oz) * invz, thz = (f2{q2.z, q2.w} - oz) * invz;
                            const float tmin0 = vmax3(vmin(tlx.x, thx.x), vmin(tly.x, thy.x), vmax(vmin(tlz.x, thz.x), 0.0f));
                            const float tfar0 = vmin3(vmax(tlx.x, thx.x), vmax(tly.x, thy.x), vmax(tlz.x, thz.x));
                            const float tmin1 = vmax3(vmin(tlx.y, thx.y), vmin(tly.y, thy.y), vmax(vmin(tlz.y, thz.y), 0.0f));
                            const float tfar1 = vmin3(vmax(tlx.y, thx.y), vmax(tly.y, thy.y), vmax(tlz.y, thz.y));
                            const bool leaf0 = c0 < 0 && tmin0 < tfar0;
                            const bool leaf1 = c1 < 0 && tmin1 < tfar1 && twoChildren;
                            if (FULL_DIAG && !refDiag) boundsHits += (leaf0 ? 1.0f : 0.0f) + (leaf1 ? 1.0f : 0.0f);
                            cand[nc * BT] = (Code)~c0;
                            nc += leaf0 ? 1 : 0;
                            cand[nc * BT] = (Code)~c1;
                            nc += leaf1 ? 1 : 0;
                        }
                    }
                }
                if (FULL_DIAG && !refDiag) candidates += (float)nc;
                while (nc > 0) {
                    STAT_ADD(5, (threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) ? 1 : 0);
                    STAT_LANES(6);
                    nc--;
                    const int i = (int)cand[nc * BT];
                    if (VOLUMES) {
                        // FindHits keeps EVERY hit (:457-460) and injects an exit hit for volume hulls (Box / Sphere, :463-469)
                        const unsigned mw = *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (uint32_t)i * 4u);
                        const unsigned type = mw >> kPrimTypeShift;
                        float tmin = 0.0f;
                        for (int pass = 0; pass < 2; pass++) {
                            float t; V3 nl; float4 rq;
                            if (!general_hit<ALL_LDS>(sc, L, i, type, ro, rd, rtime, tmin, t, nl, rq)) break;
                            const float dn = dot(normalize(rotate(rq, nl)), rd);
                            const unsigned code = (unsigned)i | (dn < 0 ? 0x40000000u : 0u) | (dn > 0 ? 0x80000000u : 0u);
                            if (nHits < kMaxHits) {
                                hitT[nHits] = t; hitTmin0[nHits] = tmin; hitCode[nHits] = code;
                                nHits++;
                            } else if ((unsigned)(nHits - kMaxHits) < A.hitSpillEntries && A.hitSpill) {
                                hit_set(hitT, hitTmin0, hitCode, hit_spill_of(A), nHits, HitRec{t, tmin, code});      // the list grows into the lane's spill column
                                nHits++;
                            } else {
                                hitOverflow = true;                                      // more surfaces than the context's hitListCapacity: reported, not ignored
                            }
                            if (((mw >> 16) & 3u) != MAT_CLASS_VOLUME || !(type == RTOW_ENTITY_BOX || type == RTOW_ENTITY_SPHERE)) break;
                            tmin = t + 0.001f;
                        }
                    } else if (GENERAL) {
                        const unsigned type = TRIANGLES_ONLY ? (unsigned)RTOW_ENTITY_TRIANGLE : *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (uint32_t)i * 4u) >> kPrimTypeShift;
                        float t; V3 nl = v3(0, 0, 0); float4 rq;
                        if (TRI_HOT ? tri_hit_hot<ALL_LDS>(sc, L, i, ro, rd, 0.0f, t, nl.x, nl.y) : general_hit<ALL_LDS, TRIANGLES_ONLY>(sc, L, i, type, ro, rd, rtime, 0.0f, t, nl, rq)) {
                            // two surfaces at the bit-identical distance: the reference's sorted hit list starts with the one that
                            // comes first in its tree's leaf order (rtow_reforder.h)
                            const unsigned* rank = reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.rankOffset));
                            // Straight-line selects on purpose.  Written as `if (t < best || (t == best && ...)) { best = t; prim = i; keptNormal = nl; }`
                            // one build of this kernel (hipcc 7.2, found when s_setprio instructions shifted its code generation) merged the nested
                            // conditions so that a lane winning BY THE TIE RULE took nl.x but kept the old normal's y and z: 1 pixel of the 1080p mesh
                            // frame with a 1-ulp wrong normal AOV, 4 097 pixels of the coplanar frame shaded off the wrong twin.
                            const bool tie = t == best && prim >= 0;
                            if (TIE_WATCH) {
                                // the watch of the all-triangle kinds (see TIE_WATCH above): same mark as the sphere kinds' below
                                if (tie) {
#if defined(__HIP_DEVICE_COMPILE__)
                                    const SampleKernelArgs* rareArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                                    asm volatile("" : "+s"(rareArgs));
#else
                                    const SampleKernelArgs* rareArgs = &A;
#endif
                                    unsigned* const bits = rareArgs->tieBits;
                                    if (bits) atomicOr(bits + ((unsigned)pix >> 5), 1u << ((unsigned)pix & 31u));
                                }
                            }
                            const unsigned rankHeld = tie ? rank[prim] : 0u;
                            const bool take = t < best || (tie && rank[i] < rankHeld);
                            if (EXACT_TIES) tieAtBest = tie ? true : (t < best ? false : tieAtBest);
                            best = take ? t : best;
                            prim = take ? i : prim;
                            if (KEEP_NORMAL) { keptNormal.x = take ? nl.x : keptNormal.x; keptNormal.y = take ? nl.y : keptNormal.y; if (!TRI_HOT) keptNormal.z = take ? nl.z : keptNormal.z; }
                        }
                    } else {
                        V3 c; float r, t;
                        sphere_at<ALL_LDS, HAS_MOTION>(sc, L, i, rtime, c, r);
                        if (sphere_hit(sub(ro, c), rd, a, r, t) && t <= best) {
                            // same tie rule as above (duplicate or exactly tangent spheres)
                            const unsigned* rank = reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.rankOffset));
                            if (EXACT_TIES) tieAtBest = !(t < best) && prim >= 0;                    // t == best here
                            // The watch: a second sphere at exactly the nearest distance so far - the rank rule below is exact for rays of at most 16 hits only.
                            // The event is as good as absent from real scenes, so it costs the lane no state: the pixel's bit is set in a bitmap in memory, and after the launch the
                            // exact-tie kernel of this kind renders the marked pixels again, from the launch's inputs, over what this kernel stored.  Per pixel and sticky: a tie at a
                            // distance that a later, nearer hit supersedes marks the pixel too - a pixel rendered twice, never a different result.
                            if (TIE_WATCH) {
                                // (one rare region for the rank rule's two lookups AND the mark: an extra conditional region per accepted hit cost the headline 1 %)
                                bool take = t < best;
                                if (!take && prim >= 0) {
                                    take = rank[i] < rank[prim];
#if defined(__HIP_DEVICE_COMPILE__)
                                    const SampleKernelArgs* rareArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();      // loaded here, on use: no register holds the bitmap's address through the stages
                                    asm volatile("" : "+s"(rareArgs));
#else
                                    const SampleKernelArgs* rareArgs = &A;
#endif
                                    unsigned* const bits = rareArgs->tieBits;
                                    if (bits) atomicOr(bits + ((unsigned)pix >> 5), 1u << ((unsigned)pix & 31u));
                                }
                                if (take) { best = t; prim = i; }
                            } else if (t < best || (prim >= 0 && rank[i] < rank[prim])) { best = t; prim = i; }
                        }
                    }
                }
                if (!LONG_LISTS || cur > -2) break;               // (else: more of the camera ray's list)
                }
                if (cur >= 0) st = ST_TRAV;        // the list was full: resume the walk, now pruned by `best`
                else classify();
            }
        }
        if (due(st == ST_HIT, ST_HIT, A.tune[3])) {
            ran = true;
            STAGE_MARK(3);
            // ================= surface hit: Entity.Hit record + Material.Scatter =================
            STAT_ADD(7, 1);
            if (st == ST_HIT) {
                STAT_LANES(8);
                DBG_TRACE(VOLUMES && insideHit ? 1 : 0, prim, best);
                if (EXACT_TIES && !VOLUMES && tieAtBest) {
                    // two surfaces at exactly this distance: let the reference's own procedure pick (rare; see resolve_nearest_tie)
                    tieAtBest = false;
#if defined(__HIP_DEVICE_COMPILE__)
                    const SampleKernelArgs* argsInKernarg = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
                    const SampleKernelArgs* argsInKernarg = &A;                                                           // host pass of the HIP compiler: never executed
#endif
                    const int winner = resolve_nearest_tie<ALL_LDS, BASE, Code, BT>(sc, &argsInKernarg->layout, ro, rd, rtime, stack, A.overflowFlag, hit_spill_of(A));
                    if (winner >= 0 && winner != prim) {
                        prim = winner;
                        if (KEEP_NORMAL) {
                            const unsigned type = TRIANGLES_ONLY ? (unsigned)RTOW_ENTITY_TRIANGLE : *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (uint32_t)prim * 4u) >> kPrimTypeShift;
                            float t2; float4 rq;
                            if (TRI_HOT) (void)tri_hit_hot<ALL_LDS>(sc, L, prim, ro, rd, 0.0f, t2, keptNormal.x, keptNormal.y);
                            else (void)general_hit<ALL_LDS, TRIANGLES_ONLY>(sc, L, prim, type, ro, rd, rtime, 0.0f, t2, keptNormal, rq);
                        }
                    }
                }
                unsigned mi = 0;
                if (!(VOLUMES && insideHit)) mi = *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (uint32_t)prim * 4u);
                unsigned matIdx = mi & 0xffffu;
                unsigned cls = (mi >> 16) & 3u;                                // shading class packed by the scene compiler
                const float t = best;
                const V3 P = v3(ro.x + t * rd.x, ro.y + t * rd.y, ro.z + t * rd.z);           // ray.GetPoint(distance), world space
                V3 N;
                float2 hitUv = make_float2(0, 0);                               // rec.TexCoords
                if (VOLUMES && insideHit) {
                    // new HitRecord(totalDistance, ray.GetPoint(totalDistance), -ray.Direction, default); material = the volume (:272-273)
                    N = neg(rd);
                    matIdx = (unsigned)curVol;
                    cls = MAT_CLASS_VOLUME;
                } else if (GENERAL) {
                    // re-run the winning primitive's test for its entity-space normal, then rotate it out (RT/Entity.cs:62-66)
                    if (KEEP_NORMAL && TRI_HOT) {
                        // TEST kept this hit's barycentric (u, v): the normal blend and the rotation come from the winner's GpuTriCold record
                        float4 rq;
                        const V3 nLocal = tri_normal_cold<ALL_LDS>(sc, L, prim, keptNormal.x, keptNormal.y, rq);
                        N = normalize(rotate(rq, nLocal));
                    } else if (KEEP_NORMAL) {
                        // TEST kept the entity-space normal of this very hit; only the rotation is fetched again (GpuPrim: [6] for triangles, else [0])
                        const float4* pp = reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.primOffset) + (uint32_t)prim * 128u);
                        const float4 rq = pp[(TRIANGLES_ONLY || (mi >> kPrimTypeShift) == RTOW_ENTITY_TRIANGLE) ? 6 : 0];
                        N = normalize(rotate(rq, keptNormal));
                    } else {
                        float t2; V3 nLocal; float4 rq;
                        (void)general_hit<ALL_LDS, TRIANGLES_ONLY>(sc, L, prim, mi >> kPrimTypeShift, ro, rd, rtime, hitTmin, t2, nLocal, rq, TEXTURED ? &hitUv : nullptr);
                        N = normalize(rotate(rq, nLocal));
                    }
                } else {
                    V3 c; float radius;
                    sphere_at<ALL_LDS, HAS_MOTION>(sc, L, prim, rtime, c, radius);
                    const V3 oc = sub(ro, c);
                    const V3 nLocal = div3(v3(oc.x + t * rd.x, oc.y + t * rd.y, oc.z + t * rd.z), radius);                   // r.GetPoint(t) / radius
                    N = normalize(nLocal);                                                    // RT/Entity.cs:65
                }
                const uint8_t* mp = section<ALL_LDS>(sc, L.materialOffset) + matIdx * 64u;
                // A material record that comes through L1 / L2 is fetched WHOLE here, four quads in flight at once: left to the compiler the second half is loaded dword by dword
                // inside the class bodies that use it - up to three more dependent round trips per hit (roughness for the shared hemisphere pass, then the general / dielectric
                // constants) - because that is where the values are used (RTOW_WHOLE_MATERIAL=0: A/B build without it)
                constexpr bool WHOLE_MATERIAL = RTOW_WHOLE_MATERIAL && !ALL_LDS && !TEXTURED;
                float4 m0, m1, m2whole = make_float4(0, 0, 0, 0), m3whole = make_float4(0, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
                if (WHOLE_MATERIAL) {
                    fvec4 w0, w1, w2, w3;
                    const unsigned recordOffset = L.materialOffset + matIdx * 64u;
                    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:16\n\tglobal_load_dwordx4 %2, %4, %5 offset:32\n\tglobal_load_dwordx4 %3, %4, %5 offset:48\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(recordOffset), "s"(sc.glob) : "memory");
                    m0 = make_float4(w0.x, w0.y, w0.z, w0.w); m1 = make_float4(w1.x, w1.y, w1.z, w1.w);
                    m2whole = make_float4(w2.x, w2.y, w2.z, w2.w); m3whole = make_float4(w3.x, w3.y, w3.z, w3.w);
                } else
#endif
                {
                    m0 = *reinterpret_cast<const float4*>(mp);       // albedo.xyz emission.x
                    m1 = *reinterpret_cast<const float4*>(mp + 16);  // emission.yz type metallic
                }
                V3 reflectance = v3(m0.x, m0.y, m0.z);
                V3 emission = v3(m0.w, m1.x, m1.y);
                float metallicHit = m1.w;
                float4 m2hit = make_float4(0, 0, 0, 0), m3hit = make_float4(0, 0, 0, 0);
                if (TEXTURED) {
                    m2hit = *reinterpret_cast<const float4*>(mp + 32);
                    m3hit = *reinterpret_cast<const float4*>(mp + 48);
                    if (__float_as_uint(m2hit.z) & MAT_FLAG_TEXTURED) {
                        // Material.Scatter / Emit evaluate the four textures at rec.TexCoords (RT/Material.cs:71,77-78,123,176-179), and
                        // everything derived from metallic / glossiness (prepare_materials_kernel's program) follows per hit
                        const GpuTexMaterial tm = reinterpret_cast<const GpuTexMaterial*>(A.texBlob + A.texLayout.materialOffset)[matIdx];
                        reflectance = texture_color(A, tm.albedo, hitUv);
                        emission = texture_color(A, tm.emission, hitUv);
                        const float glossiness = texture_scalar(A, tm.glossiness, hitUv);
                        float roughness, ior, invIor = 0.0f, alpha = 0.0f;
                        if (__float_as_int(m1.z) == RTOW_MATERIAL_STANDARD) {
                            metallicHit = texture_scalar(A, tm.metallic, hitUv);
                            roughness = det_sq(1 - glossiness);
                            ior = 1.5f + metallicHit * (1.1f - 1.5f);
                            alpha = roughness_to_alpha(roughness);
                        } else {
                            roughness = 1 - glossiness;
                            ior = m2hit.y;
                            invIor = RTOW_RCP(ior);
                        }
                        float r0 = (1 - ior) / (1 + ior);
                        r0 *= r0;
                        m2hit = make_float4(glossiness, m2hit.y, m2hit.z, roughness);
                        m3hit = make_float4(alpha, ior, r0, invIor);
                    }
                    texHist[depth * 6 + 0] = reflectance.x; texHist[depth * 6 + 1] = reflectance.y; texHist[depth * 6 + 2] = reflectance.z;
                    texHist[depth * 6 + 3] = emission.x; texHist[depth * 6 + 4] = emission.y; texHist[depth * 6 + 5] = emission.z;
                }
                bool white = false;
                bool perfectSpecular = false;
                V3 sdir;
                const NoiseSite at{&A, (unsigned)cx, (unsigned)cy};
                float randomEvents = VOLUMES ? pendRE : 0.0f;     // rng.RandomEvents may already hold ProbabilisticHit's increments
                pendRE = 0;

                // The first cosine-hemisphere draw around N of both Standard classes in one place: the scatter direction of the lambert class
                // and the rough normal of the general class are the same code on each lane's own random stream, so the lanes of both classes
                // run it together instead of one class after the other (the order of draws per lane is unchanged).
                constexpr bool kSharedHemisphere = !TEXTURED;
                V3 hemi = v3(0, 0, 0);
                if (kSharedHemisphere) {
                    bool want = cls == MAT_CLASS_LAMBERT;
                    if (cls == MAT_CLASS_LAMBERT) rng.skip_cosine_hemisphere(at);                       // the unused rough-normal draw (see below)
                    if (cls == MAT_CLASS_GENERAL) want = (WHOLE_MATERIAL ? m2whole.w : *reinterpret_cast<const float*>(mp + 44)) > 0;  // roughness > 0
                    if (want) hemi = rng.cosine_hemisphere(at, N);
                }
                if (VOLUMES && cls == MAT_CLASS_VOLUME) {
                    // ProbabilisticVolume (RT/Material.cs:163-168): isotropic scatter, ray time reset to 0, RandomEvents += 2
                    sdir = rng.direction(at);                          // NextFloat3Direction
                    rtime = 0;
                    randomEvents += 2;
                } else if (cls == MAT_CLASS_LAMBERT) {
                    STAT_ADD(9, (threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) ? 1 : 0); STAT_LANES(10);
                    // Standard with glossiness == 0 and metallic == 0 (RT/Material.cs:75-119): roughness = 1, so the rough normal
                    // costs one cosine-hemisphere draw (two white-noise numbers) whose result is never used (reflectionChance = saturate(fresnel * 0 * g1) = 0, and the
                    // rough-metal branch needs metallic > 0); RandomEvents = 0 + 0 + 1 * 0 + 1 * 1.
                    if (kSharedHemisphere) sdir = hemi;
                    else { rng.skip_cosine_hemisphere(at); sdir = rng.cosine_hemisphere(at, N); }
                    randomEvents += 1.0f;                          // 0 + 0 + 1 * 0 + 1 * 1 on top of whatever was pending
                } else if (cls == MAT_CLASS_GENERAL) {                                        // RT/Material.cs:75-119
                    STAT_ADD(11, (threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) ? 1 : 0); STAT_LANES(12);
                    const float4 m2 = TEXTURED ? m2hit : WHOLE_MATERIAL ? m2whole : *reinterpret_cast<const float4*>(mp + 32);  // glossiness parameter flags roughness
                    const float4 m3 = TEXTURED ? m3hit : WHOLE_MATERIAL ? m3whole : *reinterpret_cast<const float4*>(mp + 48);  // alpha ior r0 1/ior
                    const float metallic = TEXTURED ? metallicHit : m1.w;
                    const float glossiness = m2.x;
                    const float roughness = m2.w;                                 // pow(1 - glossiness, 2)
                    perfectSpecular = (__float_as_uint(m2.z) & MAT_FLAG_PERFECT_SPECULAR) != 0;
                    V3 roughN = N;
                    if (roughness > 0) {
                        const V3 h = kSharedHemisphere ? hemi : rng.cosine_hemisphere(at, N);
                        roughN = normalize(v3(N.x + roughness * (h.x - N.x), N.y + roughness * (h.y - N.y), N.z + roughness * (h.z - N.z)));
                    }
                    const float incidentCosine = -dot(rd, roughN);
                    const float fresnel = m3.z + (1 - m3.z) * det_pow5(1 - incidentCosine);   // Schlick, r0 from lerp(1.5, 1.1, metallic)
                    const float g1 = smith_g1(rd, N, m3.x);
                    const float reflectionChance = um_saturate(fresnel * glossiness * g1);

                    if (reflectionChance > 0 && rng.next(at) < reflectionChance) {
                        sdir = reflect(rd, roughN);
                        reflectance = v3(1, 1, 1);
                        white = true;
                    } else if (metallic > 0 && rng.next(at) < metallic) {
                        sdir = reflect(rd, roughN);
                    } else {
                        sdir = rng.cosine_hemisphere(at, N);
                    }
                    if (reflectionChance > 0 && reflectionChance < 1) randomEvents++;
                    if (metallic > 0 && metallic < 1) randomEvents++;
                    randomEvents += roughness * (reflectionChance + (1 - reflectionChance) * metallic);
                    randomEvents += (1 - reflectionChance) * (1 - metallic);
                } else {                                                                      // Dielectric, RT/Material.cs:121-161
                    const float4 m2 = TEXTURED ? m2hit : WHOLE_MATERIAL ? m2whole : *reinterpret_cast<const float4*>(mp + 32);
                    const float4 m3 = TEXTURED ? m3hit : WHOLE_MATERIAL ? m3whole : *reinterpret_cast<const float4*>(mp + 48);
                    STAT_ADD(13, (threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) ? 1 : 0); STAT_LANES(14);
                    perfectSpecular = true;
                    const float ior = m2.y;
                    const float roughness = m2.w;                                 // 1 - glossiness
                    const V3 rdir = rng.direction(at);                             // RandomSource.NextFloat3Direction (RT/RandomSource.cs:113-128)
                    const V3 roughN = normalize(v3(N.x + roughness * rdir.x, N.y + roughness * rdir.y, N.z + roughness * rdir.z));

                    float niOverNt, cosine;
                    V3 outwardN;
                    const float dDotN = dot(rd, roughN);
                    if (dDotN > 0) { outwardN = neg(roughN); niOverNt = ior; cosine = ior * dDotN; }
                    else { outwardN = roughN; niOverNt = m3.w; cosine = -dDotN; }

                    // Refract (:198-210)
                    const float dt = dot(rd, outwardN);
                    const float disc = 1 - niOverNt * niOverNt * (1 - dt * dt);
                    bool refractOk = false;
                    if (disc > 0) {
                        const float sq = RTOW_SQRT(disc);
                        const V3 refracted = v3(niOverNt * (rd.x - outwardN.x * dt) - outwardN.x * sq,
                                                niOverNt * (rd.y - outwardN.y * dt) - outwardN.y * sq,
                                                niOverNt * (rd.z - outwardN.z * dt) - outwardN.z * sq);
                        const float schlickV = m3.z + (1 - m3.z) * det_pow5(1 - cosine);
                        if (rng.next(at) > schlickV) { sdir = refracted; refractOk = true; }
                    }
                    if (!refractOk) {
                        sdir = reflect(rd, roughN);
                        reflectance = v3(1, 1, 1);
                        white = true;
                    }
                    randomEvents++;
                    randomEvents += roughness;
                }

                hist.set(depth, (white ? 0x8000u : 0u) | matIdx, histRows);                             // :311,330 (re-expanded at the fold)
                if (depth == 0) sampleNormal = N;                                             // :313-314
                if (firstNs < 0 && !perfectSpecular) {                                        // :316-328: sampleAlbedo = emission + reflectance of THIS hit (see endSample)
                    sampleNormal = N;
                    firstNs = depth;
                }
                randomEventsLocal += randomEvents * inv_pow2(depth);                          // RandomEvents / pow(2, depth), :332

                // ray = scattered.OffsetTowards(dot(dir, N) >= 0 ? N : -N)  (:335-336, RT/Ray.cs:18)
                const V3 offN = dot(sdir, N) >= 0 ? N : neg(N);
                ro = v3(P.x + 0.001f * offN.x, P.y + 0.001f * offN.y, P.z + 0.001f * offN.z);
                rd = sdir;
                depth++;
                if (depth == traceDepth) endSample(false, v3(0, 0, 0), v3(0, 0, 0));                       // :379-381
                else startRay();
            }
        }
        if (VOLUMES && due(st == ST_VOL, ST_VOL, A.tune[5])) {
            ran = true;
            STAGE_MARK(5);
            if (st == ST_VOL) {
                auto matOf = [&](unsigned code) { return *reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.matIndexOffset) + (code & kHitPrimMask) * 4u); };
                auto isVolume = [&](unsigned code) { return ((matOf(code) >> 16) & 3u) == MAT_CLASS_VOLUME; };
                // ---- hitBuffer.Sort(DistanceComparer) (:473-474) comes first (below); the rest of the stage reads the list through accessors ----
                auto volumeLogic = [&](auto distAt, auto codeAt, auto tminAt) {
                    // ---- DetermineVolumeContainment (:477-508) ----
                    if (curVol < 0) {
                        for (int i = 0; i < nHits; i++) {
                            const unsigned c = codeAt(i);
                            if (!isVolume(c)) continue;
                            if (c & 0x40000000u) break;                                   // entry hit, early out
                            // exit hit before an entry hit: throw a ray backwards; inside iff it meets the inner side of a volume hull
                            const V3 bd = neg(rd);
                            if (DIAG == 2) { if (refDiag) { const float2 rc = reference_counts(A.refTree, ro, bd); boundsHits += rc.x; candidates += rc.y; } }   // FindHitCandidates(backwardsRay, ...) counts too (:495)
                            V3 einv = v3(RTOW_RCP(bd.x), RTOW_RCP(bd.y), RTOW_RCP(bd.z));                    // math.rcp + "convert NaN to INFINITY" (:409-412)
                            if (einv.x != einv.x) einv.x = __builtin_inff();
                            if (einv.y != einv.y) einv.y = __builtin_inff();
                            if (einv.z != einv.z) einv.z = __builtin_inff();
                            const V3 binv = einv;                                                    // leaf boxes are exact: the walk needs the exact reciprocal too
                            bool insideVolume = false;
                            int bsp = 0, bcur = 0;
                            while (bcur >= 0) {                                              // FindHitCandidates(backwardsRay): no pruning
                                float4 q0, q1, q2;
                                int c0, c1;
                                load_node<ALL_LDS, SPLIT_NODES>(sc, L, bcur, q0, q1, q2, c0, c1);
                                const float t0x = (q0.x - ro.x) * binv.x, t1x = (q1.z - ro.x) * binv.x, u0x = (q0.y - ro.x) * binv.x, u1x = (q1.w - ro.x) * binv.x;
                                const float t0y = (q0.z - ro.y) * binv.y, t1y = (q2.x - ro.y) * binv.y, u0y = (q0.w - ro.y) * binv.y, u1y = (q2.y - ro.y) * binv.y;
                                const float t0z = (q1.x - ro.z) * binv.z, t1z = (q2.z - ro.z) * binv.z, u0z = (q1.y - ro.z) * binv.z, u1z = (q2.w - ro.z) * binv.z;
                                const bool h0 = vmax3(vmin(t0x, t1x), vmin(t0y, t1y), vmax(vmin(t0z, t1z), 0.0f)) <= vmin3(vmax(t0x, t1x), vmax(t0y, t1y), vmax(t0z, t1z));
                                const bool h1 = vmax3(vmin(u0x, u1x), vmin(u0y, u1y), vmax(vmin(u0z, u1z), 0.0f)) <= vmin3(vmax(u0x, u1x), vmax(u0y, u1y), vmax(u0z, u1z)) && twoChildren;
                                for (int side = 0; side < 2; side++) {
                                    const int cc = side ? c1 : c0;
                                    if (!(side ? h1 : h0) || cc >= 0) continue;
                                    const unsigned mw = matOf((unsigned)~cc);
                                    if (((mw >> 16) & 3u) != MAT_CLASS_VOLUME) continue;    // AnyBackwardsVolumeEntryHit (:510-524)
                                    {
                                        // The probe starts ON a surface with tMin = 0, so whether the hull is a candidate at all is decided by the
                                        // reference's slab test (RT/HitTests.cs:9-21) on the reference tree's box of this entity; repeat it exactly.
                                        const float4* cb = reinterpret_cast<const float4*>(section<ALL_LDS>(sc, L.cullOffset) + (unsigned)~cc * 32u);
                                        const float4 lo = cb[0], hi = cb[1];
                                        const float a0x = (lo.x - ro.x) * einv.x, a1x = (hi.x - ro.x) * einv.x;
                                        const float a0y = (lo.y - ro.y) * einv.y, a1y = (hi.y - ro.y) * einv.y;
                                        const float a0z = (lo.z - ro.z) * einv.z, a1z = (hi.z - ro.z) * einv.z;
                                        const float tn = um_max(0.0f, um_max(um_max(um_min(a0x, a1x), um_min(a0y, a1y)), um_min(a0z, a1z)));
                                        const float tf = um_min(um_min(um_max(a0x, a1x), um_max(a0y, a1y)), um_max(a0z, a1z));
                                        if (!(tn < tf)) continue;
                                    }
                                    float t; V3 nl; float4 rq;
                                    if (general_hit<ALL_LDS>(sc, L, ~cc, mw >> kPrimTypeShift, ro, bd, rtime, 0.0f, t, nl, rq) && dot(normalize(rotate(rq, nl)), bd) > 0) insideVolume = true;
                                }
                                const bool in0 = h0 && c0 >= 0, in1 = h1 && c1 >= 0;
                                if (in0 && in1) { stack[bsp * BT] = (Code)c1; bsp++; bcur = c0; }
                                else if (in0 || in1) bcur = in0 ? c0 : c1;
                                else if (bsp > 0) { bsp--; bcur = (int)stack[bsp * BT]; }
                                else bcur = -1;
                            }
                            if (insideVolume) { curVol = (int)(matOf(c) & 0xffffu); break; }
                        }
                    }
                    for (int i = 0; i < nHits; i++) DBG_TRACE(10 + i, codeAt(i), distAt(i));
                    DBG_TRACE(9, 0, 0.0f);
                    // ---- the hit loop of Sample with the volume branch (:205-303) ----
                    int hitIndex = 0;
                    int chosen = -1;
                    insideHit = false;
                    while (hitIndex < nHits) {
                        const unsigned c = codeAt(hitIndex);
                        const unsigned mw = matOf(c);
                        if (curVol >= 0 || ((mw >> 16) & 3u) == MAT_CLASS_VOLUME) {
                            const bool isEntryHit = curVol < 0;
                            if (curVol < 0) curVol = (int)(mw & 0xffffu);
                            int exitHitIndex = hitIndex, lastExitIndex = -1, sameMaterialEntries = 0;
                            while (exitHitIndex < nHits) {
                                const unsigned ec = codeAt(exitHitIndex);
                                if ((int)(matOf(ec) & 0xffffu) == curVol) {
                                    if (ec & 0x40000000u) sameMaterialEntries++;
                                    else { sameMaterialEntries--; lastExitIndex = exitHitIndex; }
                                    if (sameMaterialEntries <= 0) break;
                                } else
                                    break;
                                exitHitIndex++;
                            }
                            if (sameMaterialEntries > 0 && lastExitIndex != -1) exitHitIndex = lastExitIndex;
                            if (exitHitIndex < nHits) {
                                float distanceInVolume = distAt(exitHitIndex);
                                float entryDistance = 0;
                                if (isEntryHit) { entryDistance = distAt(hitIndex); distanceInVolume -= distAt(hitIndex); }
                                // Material.ProbabilisticHit (RT/Material.cs:49-65)
                                const float density = *reinterpret_cast<const float*>(section<ALL_LDS>(sc, L.materialOffset) + (unsigned)curVol * 64u + 36u);
                                pendRE++;
                                const float volumeHitDistance = -(RTOW_RCP(um_max(density, 1.1920928955078125e-7f))) * det_log(rng.next(NoiseSite{&A, (unsigned)cx, (unsigned)cy}));
                                if (volumeHitDistance < distanceInVolume) {
                                    best = entryDistance + volumeHitDistance;                    // we hit inside the volume
                                    insideHit = true;
                                    break;
                                }
                                curVol = -1;                                                     // no hit inside the volume, exit it
                                const unsigned xc = codeAt(exitHitIndex);
                                if (isVolume(xc) && (xc & 0x80000000u)) { hitIndex = exitHitIndex + 1; continue; }   // volume exit: next hit
                                chosen = exitHitIndex;                                           // obstacle
                                break;
                            }
                            nHits = 0;                                                           // no more surfaces (volume has holes)
                            break;
                        }
                        chosen = hitIndex;
                        break;
                    }
                    if (insideHit) { prim = -1; st = ST_HIT; }
                    else if (chosen >= 0 && chosen < nHits) { best = distAt(chosen); hitTmin = tminAt(chosen); prim = (int)(codeAt(chosen) & kHitPrimMask); st = ST_HIT; }
                    else st = ST_SKY;
                };
                const HitSpill spill = hit_spill_of(A);
                if (nHits > kMaxHits) sort_hit_list_spilled(hitT, hitTmin0, hitCode, spill, nHits, reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.rankOffset)));
                else if (nHits > 1) sort_hit_list(hitT, hitTmin0, hitCode, nHits, reinterpret_cast<const unsigned*>(section<ALL_LDS>(sc, L.rankOffset)));
                volumeLogic([&](int i) { return hit_get(hitT, hitTmin0, hitCode, spill, i).t; }, [&](int i) { return hit_get(hitT, hitTmin0, hitCode, spill, i).code; },
                            [&](int i) { return hit_get(hitT, hitTmin0, hitCode, spill, i).tmin0; });
            }
        }
        if (due(st == ST_SKY, ST_SKY, A.tune[4])) {
            ran = true;
            STAGE_MARK(4);
            // ================= sky (:341-374), then fold tail -> head (:384-396) =================
            STAT_ADD(15, 1);
            if (st == ST_SKY) {
                DBG_TRACE(2, 0xffff, 0.0f);
                V3 sky = v3(0, 0, 0);
                const SampleKernelArgs* skyArgs = &A;                      // the sky's seven constants: read on use in the kernels that read the view's so (REGEN)
#if defined(__HIP_DEVICE_COMPILE__)
                if (COLD_VIEW) { skyArgs = (const SampleKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(skyArgs)); }
#endif
                const SampleKernelArgs& EA = *skyArgs;
                const RtowEnvironment& ENV = LDS_VIEW ? *reinterpret_cast<const RtowEnvironment*>(ldsConst + 22) : EA.environment;
                if (ENV.skyType == RTOW_SKY_GRADIENT) {
                    const float s = 0.5f * (rd.y + 1);
                    const V3 b = v3(ENV.skyBottomColor), tp = v3(ENV.skyTopColor);
                    sky = v3(b.x + s * (tp.x - b.x), b.y + s * (tp.y - b.y), b.z + s * (tp.z - b.z));
                } else if (ENV.skyType == RTOW_SKY_CUBEMAP) {
                    sky = cubemap_sample(A, rd);
                }
                // randomEventsLocalAcc += rng.RandomEvents / pow(2, depth) (:363): RandomEvents is 0 here unless a ProbabilisticHit
                // that found nothing left its increment pending
                if (VOLUMES) { randomEventsLocal += pendRE * inv_pow2(depth); pendRE = 0; }
                if (firstNs < 0) sampleNormal = neg(rd);                                      // and sampleAlbedo = the sky colour (endSample)

                V3 col = sky; // 0 * 1 + sky
                for (int i = depth - 1; i >= 0; i--) {
                    const unsigned code = hist.get(i, histRows);
                    const uint8_t* mp = section<ALL_LDS>(sc, L.materialOffset) + (code & 0x7fffu) * 64u;
                    const float4 m0 = *reinterpret_cast<const float4*>(mp);
                    const float2 m1 = *reinterpret_cast<const float2*>(mp + 16);
                    const bool white = (code & 0x8000u) != 0;
                    if (TEXTURED) {
                        const V3 att = white ? v3(1, 1, 1) : v3(texHist[i * 6 + 0], texHist[i * 6 + 1], texHist[i * 6 + 2]);
                        col = v3(col.x * att.x + texHist[i * 6 + 3], col.y * att.y + texHist[i * 6 + 4], col.z * att.z + texHist[i * 6 + 5]);
                        continue;
                    }
                    // attenuation = white ? 1 : albedo, as a bit merge rather than a select: a select on a loaded value is turned into a
                    // branch around the load (with its own s_waitcnt lgkmcnt(0)), which serialises the LDS latencies of the unrolled fold
                    const unsigned wmask = (unsigned)((int)(code << 16) >> 31);                     // all ones iff bit 15 (white) is set
                    auto pick = [&](float albedo) { return __uint_as_float((__float_as_uint(albedo) & ~wmask) | (0x3f800000u & wmask)); };
                    const V3 att = v3(pick(m0.x), pick(m0.y), pick(m0.z));
                    col = v3(col.x * att.x + m0.w, col.y * att.y + m1.x, col.z * att.z + m1.y);
                }
                endSample(true, col, sky);
            }
        }
        STAGE_MARK(7);
        // nothing met its threshold: force the most populated stage next trip (or stop when every lane is dead)
        if (ran) {
            force = -1;
        } else {
            int top = 0;
            force = -1;
            for (int k = ST_REGEN; k < ST_COUNT; k++) {
                const int n = (int)__popcll(__ballot(st == k));
                if (n > top) { top = n; force = k; }
            }
            if (top == 0) {
                if (!chained || __ballot(st == ST_IDLE) == 0ull) break;
                // every lane of this wave that is not done is parked behind pixels other waves are still tracing: wait a little, ask again
                __builtin_amdgcn_s_sleep(64);
                if (st == ST_IDLE) st = ST_REGEN;
                force = ST_REGEN;
            }
        }
    }
}

template <bool ALL_LDS, int KIND, int HW, int DIAG, int NOISE, bool PER_SAMPLE, int GEO = 0>
hipError_t launchVariant(const SampleKernelArgs& args, int numBlocks, hipStream_t stream)
{
    auto k = sample_batch_kernel<ALL_LDS, KIND, HW, DIAG, NOISE, PER_SAMPLE, GEO>;
    const size_t ldsBytes = (size_t)args.ldsFrontBytes + (size_t)kQueueBytes + args.ldsSceneBytes;
    // the launch's LDS plan must be the plan of THIS variant: history rows where the codes beyond the registers go, a stack row per level of the tree
    if (args.ldsStackRows < 1u || ldsBytes > (size_t)kLdsBytesMax) return hipErrorInvalidValue;
    if (HW == 32 && args.traceDepth > kHistoryInRegisters) {
        const uint32_t rows = (uint32_t)(args.traceDepth - kHistoryInRegisters);
        if ((args.ldsHistRows > 0u && args.ldsHistOffset == 0u) || (args.ldsHistRows < rows && (ALL_LDS || !args.histSpill || args.histSpillRows < rows - args.ldsHistRows))) return hipErrorInvalidValue;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(numBlocks), dim3(geo_block_threads(GEO)), ldsBytes, stream, args);
    return hipGetLastError();
}

// Which instantiation serves a batch.  The kernel is specialised where it pays - the reference stream and the per-sample policy at trace
// depth <= 8 (4 history words), the reference stream at depth <= 16 (8 words), both without the FULL_DIAGNOSTICS counters - and generic
// elsewhere: ONE variant with the full history (32 words) and the counters switched on serves every deeper path, every 16-byte
// diagnostics record, the texture-driven noise sources and the per-sample policy beyond depth 8 (the counters cost ~2 % there; the record
// format is chosen at run time from diagnosticsStride).  Scenes that need the exact-tie resolver (kExactTiesBit) are rare: depth <= 16 shares
// the 8-word variant.  On top of these (tests/test_gpu_variants.py runs every variant on every build):
//   * wide codes (args.wideCodes, scenes beyond 65 535 entities / nodes; the tree is read from HBM, so ALL_LDS = false only): every scene kind
//     (a host that ingests triangle meshes produces spheres, general entities, triangles, textured ones - and a volume scene as soon as one fog
//     volume stands among the meshes), each with and without the exact-tie resolver where the kind has one - as the specialised
//     reference-stream variant (4 words, or 8 with the resolver) plus the generic one per noise source / RNG policy.
// the exact-tie kinds get the 4-word history variant (depth <= 8) too (round 2 let them share the 8-word one up to depth 16; on the kinds whose
// kernels spill - general, textured, triangles - the four registers are worth 3-4 %)
template <int KIND>
constexpr bool kTiesWithShortHistory = true;

template <bool ALL_LDS, int KIND, int GEO>
hipError_t launchByDiagGeo(const SampleKernelArgs& args, int numBlocks, hipStream_t stream)
{
    constexpr bool TIES = (KIND & kExactTiesBit) != 0;
    constexpr bool WIDE = (GEO & kGeoWide) != 0;
    const bool fullDiag = args.diagnostics && args.diagnosticsStride >= 16;
    // (which history width serves the launch is historyWords' decision - rtow_kernels.h - because the host sizes the launch's LDS from it)
    const int hw = historyWords(args.noiseColor, args.unitRecords != nullptr, WIDE, TIES, fullDiag, args.traceDepth);
    // lanes in a hurry (kGeoHurry): the launch carries a rate in tune[7], above the pixel gate's eight bits (plain and chained launches of static-sphere scenes: rtow_api.hip) - the
    // twins of that kind's generic variants serve it.  Every other variant reads tune[7] as the pixel gate alone and would wait for millions of lanes at every pixel boundary:
    // a launch with a rate that no twin serves is refused, loudly, instead of running slowly
    constexpr bool HAS_HURRY_TWIN = RTOW_URGENT_LANES && !WIDE && !TIES && (KIND & 7) == SCENE_KIND_SPHERES;
    const bool hurry = ((uint32_t)args.tune[7] >> 8) != 0u;
    if (hurry && !(HAS_HURRY_TWIN && args.noiseColor == RTOW_NOISE_WHITE && !args.unitRecords && hw == 32 && !(fullDiag && args.refTree))) return hipErrorInvalidValue;
    if (args.noiseColor == RTOW_NOISE_BLUE) return launchVariant<ALL_LDS, KIND, 32, 2, RTOW_NOISE_BLUE, false, GEO>(args, numBlocks, stream);
    if (args.noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) return launchVariant<ALL_LDS, KIND, 32, 2, RTOW_NOISE_SPATIOTEMPORAL_BLUE, false, GEO>(args, numBlocks, stream);
    if (args.unitRecords) {      // RTOW_RNG_PER_SAMPLE
        if constexpr (!TIES && !WIDE) if (hw == 4) return launchVariant<ALL_LDS, KIND, 4, 0, RTOW_NOISE_WHITE, true, GEO>(args, numBlocks, stream);
        return launchVariant<ALL_LDS, KIND, 32, 2, RTOW_NOISE_WHITE, true, GEO>(args, numBlocks, stream);
    }
    // pinhole twins (kGeoPinhole) of the sphere kinds' register-history variants: what the benchmark configurations with aperture 0 run
    // Only the kernels whose tree is beyond LDS have the twin: 10 000 spheres +0.9 % (118 VGPRs, nothing spilled); with the scene in LDS the twin spills 13 SGPRs instead of 25
    // but four VGPRs into a 12-byte private segment, and runs exactly as fast as the general variant (profiles/r06e_pinhole_relax_alllambert.json)
    constexpr bool HAS_PINHOLE_TWIN = RTOW_PINHOLE && !ALL_LDS && !WIDE && (KIND & 7) == SCENE_KIND_SPHERES;
    // (with a lens radius of 0 the reference still adds right * 0 + up * 0 - a zero of either sign - to the origin and subtracts it from the lower left corner: that leaves every
    // NON-ZERO component as it is, and only those; a view with a component that is exactly zero keeps the general variant, whose zeros carry the reference's signs)
    const RtowView& vw = args.view;
    const bool pinhole = vw.lensRadius == 0.0f && vw.origin.x != 0.0f && vw.origin.y != 0.0f && vw.origin.z != 0.0f &&
                         vw.lowerLeftCorner.x != 0.0f && vw.lowerLeftCorner.y != 0.0f && vw.lowerLeftCorner.z != 0.0f;
    if constexpr (HAS_PINHOLE_TWIN) {
        if (pinhole && hw == 4) return launchVariant<ALL_LDS, KIND, 4, 0, RTOW_NOISE_WHITE, false, GEO | kGeoPinhole>(args, numBlocks, stream);
        if (pinhole && hw == 8) return launchVariant<ALL_LDS, KIND, 8, 0, RTOW_NOISE_WHITE, false, GEO | kGeoPinhole>(args, numBlocks, stream);
    }
    if (hw == 4) return launchVariant<ALL_LDS, KIND, 4, 0, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream);
    if constexpr (WIDE) {
        if constexpr (TIES) { if (hw == 8) return launchVariant<ALL_LDS, KIND, 8, 0, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream); }
    } else {
        // (16-word history variants for depth 17 .. 32 - the reference host's own default traceDepth - were built again in round 5 and measured under group launches, the
        // adaptive schedule and single launches: 2 - 10 % SLOWER than the generic 32-word kernels everywhere (profiles/r05e_history16_variants.json); removed again.  Round 6:
        // the generic kernels keep the codes beyond depth 8 in LDS rows instead of a private segment)
        if (hw == 8) return launchVariant<ALL_LDS, KIND, 8, 0, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream);
        if constexpr (HAS_HURRY_TWIN) { if (hurry && !fullDiag) return launchVariant<ALL_LDS, KIND, 32, 0, RTOW_NOISE_WHITE, false, GEO | kGeoHurry>(args, numBlocks, stream); }
        if (!fullDiag) return launchVariant<ALL_LDS, KIND, 32, 0, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream);
    }
    // FULL_DIAGNOSTICS records (and every deeper wide-code launch): the counters of this library's own walk, or - a variant of its own, with a private segment - the reference's
    if (args.refTree) return launchVariant<ALL_LDS, KIND, 32, 2, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream);
    if constexpr (HAS_HURRY_TWIN) { if (hurry) return launchVariant<ALL_LDS, KIND, 32, 1, RTOW_NOISE_WHITE, false, GEO | kGeoHurry>(args, numBlocks, stream); }
    return launchVariant<ALL_LDS, KIND, 32, 1, RTOW_NOISE_WHITE, false, GEO>(args, numBlocks, stream);
}

template <int KIND>
constexpr bool kind_has_wide_codes() { return true; }   // every scene kind (volume kinds since round 3: a triangle-mesh scene with one fog volume is a volume scene)

template <bool ALL_LDS, int KIND>
hipError_t launchByDiag(const SampleKernelArgs& args, int numBlocks, hipStream_t stream)
{
    if constexpr (!ALL_LDS && kind_has_wide_codes<KIND>()) {
        if (args.wideCodes) return launchByDiagGeo<false, KIND, kGeoWide>(args, numBlocks, stream);
    }
    if (args.wideCodes) return hipErrorInvalidValue;                                    // refused at upload (rtow_api.hip): never reached
    if (args.blockThreads != kBlockThreads) return hipErrorInvalidValue;
    return launchByDiagGeo<ALL_LDS, KIND, 0>(args, numBlocks, stream);
}

} // namespace

// One launcher per scene kind, each in its own translation unit (rtow_sample_kind*.hip).
#define RTOW_DEFINE_KIND_LAUNCHER(NAME, KIND)                                                                                         \
    hipError_t NAME(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds)                                    \
    {                                                                                                                                 \
        return allLds ? launchByDiag<true, KIND>(args, numBlocks, stream) : launchByDiag<false, KIND>(args, numBlocks, stream);     \
    }

} // namespace rtow
