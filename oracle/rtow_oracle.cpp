/*
 * rtow_oracle.cpp - CPU ORACLE for the sample-batch path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (oracle/README.md).  The product (librtow_hip.so) shares no code with it.
 *
 * What it is: a function-by-function restatement, in strict IEEE-754 binary32, of the reference's
 * Burst job `SampleBatchJob` and everything beneath it, keeping the reference's ALGORITHMIC SHAPE:
 * collect every overlapped BVH leaf without closest-hit pruning, test every candidate through the
 * general Entity transform, sort the hits, take element 0 (JOBS/SampleBatchJob.cs:403-475).
 * Each function cites the reference file:line it follows.  Paths are relative to the reference root:
 *   JOBS/ = RaytracingInOneWeekend/Assets/Scripts/Runtime/Jobs/
 *   RT/   = RaytracingInOneWeekend/Assets/Scripts/Runtime/
 *   UNITY/= RaytracingInOneWeekend/Assets/Scripts/Unity/
 *   UTIL/ = RaytracingInOneWeekend/Assets/Scripts/Util/
 *
 * PARITY UNPINNED at one boundary: the reference ships no tests, golden
 * images or known-answer vectors, its C# cannot be compiled or run here (no dotnet/mono/Unity/Burst),
 * and the arithmetic it calls lives in third-party packages that are NOT vendored under /root/reference:
 *   com.unity.mathematics 1.2.5  (Packages/manifest.json:7)  - Random (xorshift32), math.* intrinsics
 *   com.unity.burst 1.7.0-pre.1  (Packages/manifest.json:3)  - FloatMode.Fast codegen
 *   com.unity.collections 1.0.0-pre.6 (manifest.json:4)      - NativeSortExtension.Sort
 * Their published semantics are restated below (section "Unity.Mathematics restatement") and pinned by
 * analytic / float64 known-answer tests in tests/test_oracle_kat.py; there is nothing of the reference's
 * own to pin them against.  Burst's FloatMode.Fast (reassociation, FMA contraction, ~3.5 ulp
 * transcendentals) is deliberately NOT imitated: this oracle is the literal left-to-right IEEE evaluation
 * of the C# source, with sin/cos/log/pow replaced by the deterministic functions in detmath.h.
 *
 * Build: oracle/Makefile (g++ -O2 -ffp-contract=off -fno-fast-math; a second -O3 -ffast-math build of the
 * same source is the timed "CPU baseline" that mirrors [BurstCompile(FloatMode.Fast)]).
 */
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rtow.h" /* POD scene / parameter structs only (data layout of the boundary) */
#include "detmath.h"

#define ORACLE_API extern "C" __attribute__((visibility("default")))

namespace {

/* ===================================================================================================
 * Unity.Mathematics restatement (com.unity.mathematics 1.2.5 - NOT under /root/reference; assumed
 * semantics from the published package, see header).
 * =================================================================================================== */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

inline float3 f3(float x, float y, float z) { return float3{x, y, z}; }
inline float3 f3(float s) { return float3{s, s, s}; }
inline float3 f3(const RtowFloat3& v) { return float3{v.x, v.y, v.z}; }
inline float4 f4(const RtowFloat4& v) { return float4{v.x, v.y, v.z, v.w}; }

inline float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator-(float3 a) { return f3(-a.x, -a.y, -a.z); }
inline float3 operator*(float3 a, float3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline float3 operator*(float s, float3 a) { return f3(s * a.x, s * a.y, s * a.z); }
inline float3 operator/(float3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
inline float3 operator/(float3 a, float3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }

/* bit-pattern tests so that the -ffast-math timing build cannot fold them away */
inline bool um_isnan(float x) { return (dm_asuint(x) & 0x7fffffffu) > 0x7f800000u; }
inline bool um_isinf(float x) { return (dm_asuint(x) & 0x7fffffffu) == 0x7f800000u; }
/* math.min/max: `float.IsNaN(y) || x < y ? x : y` - the FIRST operand wins when the second is NaN. */
inline float um_min(float x, float y) { return (um_isnan(y) || x < y) ? x : y; }
inline float um_max(float x, float y) { return (um_isnan(y) || x > y) ? x : y; }
inline float3 um_min(float3 a, float3 b) { return f3(um_min(a.x, b.x), um_min(a.y, b.y), um_min(a.z, b.z)); }
inline float3 um_max(float3 a, float3 b) { return f3(um_max(a.x, b.x), um_max(a.y, b.y), um_max(a.z, b.z)); }
inline float um_cmax(float3 v) { return um_max(um_max(v.x, v.y), v.z); }
inline float um_cmin(float3 v) { return um_min(um_min(v.x, v.y), v.z); }
inline float um_clamp(float x, float a, float b) { return um_max(a, um_min(b, x)); }
inline float um_saturate(float x) { return um_clamp(x, 0.0f, 1.0f); }
inline float3 um_saturate(float3 v) { return f3(um_saturate(v.x), um_saturate(v.y), um_saturate(v.z)); }
inline float um_lerp(float x, float y, float s) { return x + s * (y - x); }
inline float3 um_lerp(float3 x, float3 y, float s) { return x + s * (y - x); }
inline float um_unlerp(float a, float b, float x) { return (x - a) / (b - a); }
inline float um_rcp(float x) { return 1.0f / x; }
inline float um_round(float x) { return rintf(x); } /* (float)System.Math.Round(x): half to even */
inline float um_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float um_dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
/* cross(x, y) = (x * y.yzx - x.yzx * y).yzx */
inline float3 um_cross(float3 a, float3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float um_rsqrt(float x) { return 1.0f / sqrtf(x); }
inline float3 um_normalize(float3 v) { return um_rsqrt(um_dot(v, v)) * v; }
inline float3 um_normalizesafe(float3 v)
{
    const float len = um_dot(v, v);
    return len > 1.175494351e-38f ? v * um_rsqrt(len) : f3(0.0f);
}
/* reflect(i, n) = i - 2f * n * dot(i, n) */
inline float3 um_reflect(float3 i, float3 n) { return i - 2.0f * n * um_dot(i, n); }
/* mul(quaternion q, float3 v): t = 2 * cross(q.xyz, v); v + q.w * t + cross(q.xyz, t) */
inline float3 um_rotate(float4 q, float3 v)
{
    const float3 qv = f3(q.x, q.y, q.z);
    const float3 t = 2.0f * um_cross(qv, v);
    return v + q.w * t + um_cross(qv, t);
}
inline float4 um_inverse(float4 q)
{
    const float r = um_rcp(um_dot(q, q));
    return float4{r * q.x * -1.0f, r * q.y * -1.0f, r * q.z * -1.0f, r * q.w * 1.0f};
}
struct RigidTransform { float4 rot; float3 pos; };
inline float3 um_transform(const RigidTransform& a, float3 p) { return um_rotate(a.rot, p) + a.pos; }
inline float3 um_rotate(const RigidTransform& a, float3 d) { return um_rotate(a.rot, d); }
inline RigidTransform um_inverse(const RigidTransform& t)
{
    const float4 invRot = um_inverse(t.rot);
    return RigidTransform{invRot, um_rotate(invRot, -t.pos)};
}
const float UM_PI = 3.14159265f;
const float UM_EPSILON = 1.1920928955078125e-7f;

/* Unity.Mathematics.Random (xorshift32).  Random(seed): state = seed; NextState().
 * NextState(): t = state; state ^= state << 13; state ^= state >> 17; state ^= state << 5; return t. */
struct UmRandom {
    uint32_t state;
    /* RTOW_RNG_PER_SAMPLE_XOROSHIRO (include/rtow.h; not the reference): xoroshiro64** - state words (state, s1), output rotl(s0 * 0x9E3779BB, 5) * 5,
     * then s1 ^= s0; s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9); s1 = rotl(s1, 13) (Blackman & Vigna, public domain reference implementation restated) */
    uint32_t s1 = 0;
    bool xoroshiro = false;
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    uint32_t NextState()
    {
        if (xoroshiro) {
            const uint32_t s0 = state;
            uint32_t t1 = s1;
            const uint32_t result = rotl(s0 * 0x9E3779BBu, 5) * 5u;
            t1 ^= s0;
            state = rotl(s0, 26) ^ t1 ^ (t1 << 9);
            s1 = rotl(t1, 13);
            return result;
        }
        const uint32_t t = state;
        state ^= state << 13;
        state ^= state >> 17;
        state ^= state << 5;
        return t;
    }
    void Init(uint32_t seed) { xoroshiro = false; state = seed; NextState(); }
    void InitXoroshiro(uint32_t seed)
    {
        xoroshiro = true;
        state = seed;
        s1 = (seed * 0x85EBCA6Bu) ^ 0xC2B2AE35u;
        if ((state | s1) == 0u) s1 = 0x9E3779B9u;
        NextState();
    }
    float NextFloat() { return dm_asfloat(0x3f800000u | (NextState() >> 9)) - 1.0f; }
    float2 NextFloat2() { float2 r; r.x = NextFloat(); r.y = NextFloat(); return r; }
    float NextFloat(float mn, float mx) { return NextFloat() * (mx - mn) + mn; }
};

/* ===================================================================================================
 * RT/ runtime structs
 * =================================================================================================== */

/* RT/Ray.cs:5-20 */
struct Ray {
    float3 Origin, Direction;
    float Time;
    Ray() : Origin(f3(0)), Direction(f3(0)), Time(0) {}
    Ray(float3 o, float3 d, float t = 0) : Origin(o), Direction(d), Time(t) {}
    Ray OffsetTowards(float3 n) const { return Ray(Origin + 0.001f * n, Direction, Time); } /* :18 */
    float3 GetPoint(float t) const { return Origin + t * Direction; }                       /* :20 */
};

struct Entity;

/* RT/HitRecord.cs:6-26 */
/* ===================================================================================================
 * NativeSortExtension.Sort (com.unity.collections 1.0.0-pre.6 - a package dependency of the reference,
 * Packages/manifest.json:4, whose sources are NOT under /root/reference).  Restated from the published
 * algorithm: the introsort of .NET Core's ArraySortHelper - partitions of <= 16 elements use a 2- and a
 * 3-element compare-exchange network or a straight insertion sort, larger ones a median-of-three Hoare
 * partition, with heap sort once 2*floor(log2(n)) levels are used up.  It is NOT stable, and the order it
 * leaves equal keys in is part of the reference's behaviour (hits at bit-identical distances, entities
 * with equal Bounds.Min on the split axis).  `cmp(a, b)` returns the IComparer<T>.Compare integer.
 * =================================================================================================== */
static std::atomic<int> g_heapSortCalls{0};   /* test instrumentation: how often the depth limit was reached (oracle_kat_unity_sort_heapsorts) */
template <class T, class Cmp>
struct UnitySort {
    T* a;
    Cmp cmp;
    void SwapIfGreater(int l, int r) { if (l != r && cmp(a[l], a[r]) > 0) std::swap(a[l], a[r]); }
    void InsertionSort(int lo, int hi)
    {
        for (int i = lo; i < hi; i++) {
            int j = i;
            const T t = a[i + 1];
            while (j >= lo && cmp(t, a[j]) < 0) { a[j + 1] = a[j]; j--; }
            a[j + 1] = t;
        }
    }
    void Heapify(int i, int n, int lo)
    {
        const T val = a[lo + i - 1];
        while (i <= n / 2) {
            int child = 2 * i;
            if (child < n && cmp(a[lo + child - 1], a[lo + child]) < 0) child++;
            if (cmp(a[lo + child - 1], val) < 0) break;
            a[lo + i - 1] = a[lo + child - 1];
            i = child;
        }
        a[lo + i - 1] = val;
    }
    void HeapSort(int lo, int hi)
    {
        g_heapSortCalls.fetch_add(1, std::memory_order_relaxed);
        const int n = hi - lo + 1;
        for (int i = n / 2; i >= 1; i--) Heapify(i, n, lo);
        for (int i = n; i > 1; i--) { std::swap(a[lo], a[lo + i - 1]); Heapify(1, i - 1, lo); }
    }
    int Partition(int lo, int hi)
    {
        const int mid = lo + (hi - lo) / 2;
        SwapIfGreater(lo, mid);
        SwapIfGreater(lo, hi);
        SwapIfGreater(mid, hi);
        const T pivot = a[mid];
        std::swap(a[mid], a[hi - 1]);
        int left = lo, right = hi - 1;
        while (left < right) {
            while (cmp(pivot, a[++left]) > 0) {}
            while (cmp(pivot, a[--right]) < 0) {}
            if (left >= right) break;
            std::swap(a[left], a[right]);
        }
        std::swap(a[left], a[hi - 1]);
        return left;
    }
    void IntroSort(int lo, int hi, int depth)
    {
        while (hi > lo) {
            const int partitionSize = hi - lo + 1;
            if (partitionSize <= 16) {
                if (partitionSize == 1) return;
                if (partitionSize == 2) { SwapIfGreater(lo, hi); return; }
                if (partitionSize == 3) { SwapIfGreater(lo, hi - 1); SwapIfGreater(lo, hi); SwapIfGreater(hi - 1, hi); return; }
                InsertionSort(lo, hi);
                return;
            }
            if (depth == 0) { HeapSort(lo, hi); return; }
            depth--;
            const int p = Partition(lo, hi);
            IntroSort(p + 1, hi, depth);
            hi = p - 1;
        }
    }
};
template <class T, class Cmp>
inline void unity_sort(T* array, int length, Cmp cmp)
{
    if (length < 2) return;
    int log2floor = 0;
    while ((length >> (log2floor + 1)) != 0) log2floor++;
    UnitySort<T, Cmp>{array, cmp}.IntroSort(0, length - 1, 2 * log2floor);
}
inline int float_compare_to(float x, float y) /* System.Single.CompareTo */
{
    if (x < y) return -1;
    if (x > y) return 1;
    if (x == y) return 0;
    if (um_isnan(x)) return um_isnan(y) ? 0 : -1;
    return 1;
}

struct HitRecord {
    float Distance;
    float3 Point, Normal;
    float2 TexCoords;
    const Entity* EntityPtr;
};

/* RT/AxisAlignedBoundingBox.cs:6-22 */
struct AABB {
    float3 Min, Max;
    float3 Size() const { return Max - Min; }
    static AABB Enclose(const AABB& l, const AABB& r) { return AABB{um_min(l.Min, r.Min), um_max(l.Max, r.Max)}; }
};

/* RT/Texture.cs:23-139, constant branches only (:55-59, :100-104); None samples as 0 (:92,137). */
struct Texture {
    int Type;
    float3 MainColor;
    float Parameter;
    int ScalarValueChannel;
    int ImageSizeX = 0, ImageSizeY = 0;           /* ImageSize */
    const uint8_t* ImagePointer = nullptr;
    int PixelStride = 3;
    /* (int2)(textureCoordinates * ImageSize), then the pixel (:85-87).  The reference indexes out of the image for coordinates
     * outside [0, 1) (undefined); clamped here, as the product does. */
    const uint8_t* Pixel(float2 uv) const
    {
        int x = (int)(uv.x * (float)ImageSizeX), y = (int)(uv.y * (float)ImageSizeY);
        x = x < 0 ? 0 : x > ImageSizeX - 1 ? ImageSizeX - 1 : x;
        y = y < 0 ? 0 : y > ImageSizeY - 1 ? ImageSizeY - 1 : y;
        return ImagePointer + ((size_t)y * ImageSizeX + x) * PixelStride;
    }
    float3 SampleColor(float2 uv) const                                             /* RT/Texture.cs:51-93 */
    {
        switch (Type) {
            case RTOW_TEXTURE_CONSTANT: return MainColor;
            case RTOW_TEXTURE_CONSTANT_SCALAR: return f3(Parameter);
            case RTOW_TEXTURE_IMAGE: {
                if (ImagePointer == nullptr) return f3(0);
                const uint8_t* p = Pixel(uv);
                return f3((float)p[0], (float)p[1], (float)p[2]) / 255.0f * MainColor;
            }
        }
        return f3(0);
    }
    float SampleScalar(float2 uv) const                                             /* :96-138 */
    {
        const float main = ScalarValueChannel == 0 ? MainColor.x : ScalarValueChannel == 1 ? MainColor.y : MainColor.z;
        switch (Type) {
            case RTOW_TEXTURE_CONSTANT: return main;
            case RTOW_TEXTURE_CONSTANT_SCALAR: return Parameter;
            case RTOW_TEXTURE_IMAGE: {
                if (ImagePointer == nullptr) return 0.0f;
                return (float)Pixel(uv)[ScalarValueChannel] / 255.0f * main;
            }
        }
        return 0.0f;
    }
};

/* RT/R2.cs:8-16.  The constants are C# `const float` expressions; folded one operation at a time in binary32 here (an assumption
 * about Roslyn's constant folding).  float * uint converts the uint to float; `% 1` is the IEEE remainder with truncation (fmodf). */
inline float2 R2Next(uint32_t n)
{
    const float g = 1.32471795724474602596f;
    const float a1 = 1.0f / g;
    const float a2 = 1.0f / (g * g);
    return float2{fmodf(0.5f + a1 * (float)n, 1.0f), fmodf(0.5f + a2 * (float)n, 1.0f)};
}
inline float half_to_float(uint16_t h) /* Unity.Mathematics.half -> float: exact */
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal: man * 2^-24 */
            const float v = (float)man * 5.9604644775390625e-8f;
            bits = dm_asuint(v) | sign;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    return dm_asfloat(bits);
}
/* RT/PerPixelNoise.cs:8-37: a walk over one square noise texture, the same R2 offsets for every pixel, shifted by the pixel's coordinates */
struct PerPixelNoise {
    uint32_t cx = 0, cy = 0, rowStride = 1;
    uint32_t offX = 0, offY = 0, n = 0;
    void Init(uint32_t seed, uint32_t x, uint32_t y, uint32_t stride) { cx = x; cy = y; rowStride = stride; n = seed; Advance(); }   /* :17-25 */
    size_t Next()                                                                                                                    /* :27-33: index of the texel */
    {
        const uint32_t wx = (cx + offX) % rowStride, wy = (cy + offY) % rowStride;
        const size_t texel = (size_t)wy * rowStride + wx;
        Advance();
        return texel;
    }
    void Advance()                                                                                                                   /* :35-38 */
    {
        const float2 r = R2Next(n++);
        offX = (uint32_t)floorf(r.x * (float)rowStride);
        offY = (uint32_t)floorf(r.y * (float)rowStride);
    }
};
/* the host's noise textures (UNITY/BlueNoiseData.cs, UNITY/SpatioTemporalBlueNoiseData.cs), one texture of each set */
struct NoiseTextures {
    uint32_t blueRowStride = 0;
    const uint16_t* blue = nullptr;                 /* half4 texels */
    uint32_t stbRowStride = 0;
    const uint8_t* stbScalar = nullptr;             /* byte */
    const uint8_t* stbVector2 = nullptr;            /* RGB24 */
    const uint8_t* stbCosineUnitVector3 = nullptr;  /* RGBA32 */
    const uint8_t* stbUnitVector2 = nullptr;        /* RGB24 */
    const uint8_t* stbUnitVector3 = nullptr;        /* RGB24 */
};

/* RT/RandomSource.cs:15-150 (+ RT/BlueNoise.cs, RT/SpatioTemporalBlueNoise.cs); RandomEvents counter :33-37. */
struct RandomSource {
    int noiseColor = RTOW_NOISE_WHITE;
    UmRandom whiteNoise;
    const NoiseTextures* tex = nullptr;
    PerPixelNoise blueNoise;                                                       /* BlueNoise.perPixelNoise */
    PerPixelNoise stbScalar, stbVector2, stbCosine, stbUnit2, stbUnit3;             /* SpatioTemporalBlueNoise.perPixel* */
    float RandomEvents;
    uint32_t draws; /* oracle-only: number of NextState() / texel fetches, for the draw-order KAT */

    /* BlueNoise.Coordinates / SpatioTemporalBlueNoise.Coordinates setters (JOBS/SampleBatchJob.cs:83-88) */
    void SetCoordinates(uint32_t seed, uint32_t x, uint32_t y)
    {
        if (noiseColor == RTOW_NOISE_BLUE) blueNoise.Init(seed, x, y, tex->blueRowStride);
        if (noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) {
            stbScalar.Init(seed, x, y, tex->stbRowStride); stbVector2.Init(seed, x, y, tex->stbRowStride); stbCosine.Init(seed, x, y, tex->stbRowStride);
            stbUnit2.Init(seed, x, y, tex->stbRowStride); stbUnit3.Init(seed, x, y, tex->stbRowStride);
        }
    }
    float2 BlueTexel() { draws += 1; const uint16_t* t = tex->blue + blueNoise.Next() * 4; return float2{half_to_float(t[0]), half_to_float(t[1])}; }
    float StbFloat() { draws += 1; return (float)tex->stbScalar[stbScalar.Next()] / 256.0f; }                                         /* STBN :61 */
    float2 StbFloat2() { draws += 1; const uint8_t* t = tex->stbVector2 + stbVector2.Next() * 3; return float2{(float)t[0] / 256.0f, (float)t[1] / 256.0f}; }   /* :63-67 */

    float NextFloat()                                                             /* :130-139 */
    {
        switch (noiseColor) {
            case RTOW_NOISE_BLUE: return BlueTexel().x;                           /* BlueNoise.cs:26 */
            case RTOW_NOISE_SPATIOTEMPORAL_BLUE: return StbFloat();
            default: draws += 1; return whiteNoise.NextFloat();
        }
    }
    float2 NextFloat2()                                                           /* :141-150 */
    {
        switch (noiseColor) {
            case RTOW_NOISE_BLUE: return BlueTexel();                             /* BlueNoise.cs:28: x and y of ONE texel */
            case RTOW_NOISE_SPATIOTEMPORAL_BLUE: return StbFloat2();
            default: draws += 2; return whiteNoise.NextFloat2();
        }
    }
    float2 InUnitDisk()                                                           /* :40-61 */
    {
        float theta, radius;
        switch (noiseColor) {
            case RTOW_NOISE_BLUE:
                theta = BlueTexel().x * 2 * UM_PI;
                radius = sqrtf(BlueTexel().x);
                break;
            case RTOW_NOISE_SPATIOTEMPORAL_BLUE: {                                /* NextUnitVector2, STBN :75-79 */
                draws += 1;
                const uint8_t* t = tex->stbUnitVector2 + stbUnit2.Next() * 3;
                return float2{(float)t[0] / 256.0f * 2 - 1, (float)t[1] / 256.0f * 2 - 1};
            }
            default:
                draws += 2;
                theta = whiteNoise.NextFloat(0.0f, 2.0f * UM_PI);
                radius = sqrtf(whiteNoise.NextFloat());
                break;
        }
        float sinTheta, cosTheta;
        dm_sincosf(theta, &sinTheta, &cosTheta);
        return float2{radius * cosTheta, radius * sinTheta};
    }
    float3 OnCosineWeightedHemisphere(float3 normal);                             /* :63-89 */
    float3 NextFloat3Direction()                                                  /* :113-128 */
    {
        if (noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) {                       /* NextUnitVector3, STBN :81-85 */
            draws += 1;
            const uint8_t* t = tex->stbUnitVector3 + stbUnit3.Next() * 3;
            return f3((float)t[0] / 256.0f * 2 - 1, (float)t[1] / 256.0f * 2 - 1, (float)t[2] / 256.0f * 2 - 1);
        }
        const float2 rnd = NextFloat2();
        const float z = rnd.x * 2.0f - 1.0f;
        const float r = sqrtf(um_max(1.0f - z * z, 0.0f));
        const float angle = rnd.y * UM_PI * 2.0f;
        float s, c;
        dm_sincosf(angle, &s, &c);
        return f3(c * r, s * r, z);
    }
};

/* UTIL/Tools.cs:19-28 (corrected Frisvad / Pixar orthonormal basis) */
inline void GetOrthonormalBasis(float3 normal, float3* tangent, float3* bitangent)
{
    const float s = normal.z >= 0 ? 1.0f : -1.0f;
    const float a = -1 / (s + normal.z);
    const float b = normal.x * normal.y * a;
    *tangent = f3(1 + s * normal.x * normal.x * a, s * b, -s * normal.x);
    *bitangent = f3(b, s + normal.y * normal.y * a, -normal.y);
}
/* UTIL/Tools.cs:30-37; float3x3(c0,c1,c2) takes COLUMNS, mul(M, v) = c0*v.x + c1*v.y + c2*v.z */
inline float3 TangentToWorldSpace(float3 v, float3 normal)
{
    float3 tangent, bitangent;
    GetOrthonormalBasis(normal, &tangent, &bitangent);
    const float3 result = tangent * v.x + normal * v.y + bitangent * v.z;
    return um_normalize(result);
}
float3 RandomSource::OnCosineWeightedHemisphere(float3 normal)
{
    if (noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) {                           /* NextCosineUnitVector3, STBN :69-73: (r, b, g) */
        draws += 1;
        const uint8_t* t = tex->stbCosineUnitVector3 + stbCosine.Next() * 4;
        const float3 tangentSpaceDirection = f3((float)t[0] / 256.0f * 2 - 1, (float)t[2] / 256.0f * 2 - 1, (float)t[1] / 256.0f * 2 - 1);
        return TangentToWorldSpace(tangentSpaceDirection, normal);
    }
    const float2 uv = NextFloat2();
    const float u = uv.x;
    const float radius = sqrtf(u);
    const float theta = uv.y * 2 * UM_PI;
    float sinTheta, cosTheta;
    dm_sincosf(theta, &sinTheta, &cosTheta);
    const float2 xz = float2{radius * cosTheta, radius * sinTheta};
    const float3 tangentSpaceDirection = f3(xz.x, sqrtf(1 - u), xz.y);
    return TangentToWorldSpace(tangentSpaceDirection, normal);
}

/* RT/Microfacet.cs:53-80 */
inline float RoughnessToAlpha(float roughness)
{
    roughness = um_max(roughness, 1e-3f);
    const float x = dm_logf(roughness);
    return 1.62142f +
           0.819955f * x +
           0.1734f * x * x +
           0.0171201f * x * x * x +
           0.000640711f * x * x * x * x;
}
inline float Lambda(float3 w, float3 normal, float roughness)
{
    const float cosTheta = um_dot(normal, w);
    const float sqCosTheta = cosTheta * cosTheta;
    const float sqSinTheta = um_max(0.0f, 1 - sqCosTheta);
    const float sinTheta = sqrtf(sqSinTheta);
    const float tanTheta = sinTheta / cosTheta;
    const float absTanTheta = fabsf(tanTheta);
    if (um_isinf(absTanTheta)) return 0;
    const float alpha = RoughnessToAlpha(roughness);
    const float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1 + alpha2Tan2Theta)) / 2;
}
/* RT/Microfacet.cs:9-12 */
inline float SmithMaskingShadowing(float3 w, float3 normal, float roughness)
{
    return 1 / (1 + Lambda(w, normal, roughness));
}

/* RT/Material.cs:16-218 */
struct Material {
    int Type;
    Texture Albedo, Glossiness, Emission, Metallic;
    float parameter;
    float IndexOfRefraction() const { return parameter; }
    float Density() const { return parameter; }

    static bool AlmostEquals(float lhs, float rhs) { return fabsf(rhs - lhs) < 1e-6f; } /* UTIL/MathExtensions.cs:24-27 */

    /* :181-196 */
    bool IsPerfectSpecular() const
    {
        switch (Type) {
            case RTOW_MATERIAL_DIELECTRIC: return true;
            case RTOW_MATERIAL_STANDARD:
                return Metallic.Type == RTOW_TEXTURE_CONSTANT &&
                       AlmostEquals(Metallic.MainColor.x, 1) && AlmostEquals(Metallic.MainColor.y, 1) && AlmostEquals(Metallic.MainColor.z, 1) &&
                       Glossiness.Type == RTOW_TEXTURE_CONSTANT &&
                       AlmostEquals(Glossiness.MainColor.x, 1) && AlmostEquals(Glossiness.MainColor.y, 1) && AlmostEquals(Glossiness.MainColor.z, 1);
        }
        return false;
    }
    /* :198-210 */
    static bool Refract(float3 v, float3 n, float niOverNt, float3* refracted)
    {
        const float dt = um_dot(v, n);
        const float discriminant = 1 - niOverNt * niOverNt * (1 - dt * dt);
        if (discriminant > 0) {
            *refracted = niOverNt * (v - n * dt) - n * sqrtf(discriminant);
            return true;
        }
        *refracted = f3(0);
        return false;
    }
    /* :212-217 */
    static float Schlick(float cosine, float refractiveIndex)
    {
        float r0 = (1 - refractiveIndex) / (1 + refractiveIndex);
        r0 *= r0;
        return r0 + (1 - r0) * dm_powf(1 - cosine, 5);
    }
    /* :176-179 */
    float3 Emit(float2 texCoords) const { return Emission.SampleColor(texCoords); }

    /* :49-65 */
    bool ProbabilisticHit(float* hitDistance, RandomSource& rng) const
    {
        if (Type != RTOW_MATERIAL_PROBABILISTIC_VOLUME) return false;
        rng.RandomEvents++;
        const float volumeHitDistance = -(1 / um_max(Density(), UM_EPSILON)) * dm_logf(rng.NextFloat());
        if (volumeHitDistance < *hitDistance) {
            *hitDistance = volumeHitDistance;
            return true;
        }
        return false;
    }

    /* :68-173 */
    void Scatter(const Ray& ray, const HitRecord& rec, RandomSource& rng, float3* reflectance, Ray* scattered) const
    {
        *reflectance = Albedo.SampleColor(rec.TexCoords);
        switch (Type) {
            case RTOW_MATERIAL_STANDARD: {
                const float metallic = Metallic.SampleScalar(rec.TexCoords);
                const float glossiness = Glossiness.SampleScalar(rec.TexCoords);

                const float roughness = dm_powf(1 - glossiness, 2);
                const float3 roughNormal = roughness > 0
                    ? um_normalize(um_lerp(rec.Normal, rng.OnCosineWeightedHemisphere(rec.Normal), roughness))
                    : rec.Normal;

                const float incidentCosine = -um_dot(ray.Direction, roughNormal);
                const float ior = um_lerp(1.5f /*PlasticIor*/, 1.1f /*MetalIor*/, metallic);
                const float fresnel = Schlick(incidentCosine, ior);
                const float maskingShadowing = SmithMaskingShadowing(ray.Direction, rec.Normal, roughness);
                const float reflectionChance = um_saturate(fresnel * glossiness * maskingShadowing);

                if (reflectionChance > 0 && rng.NextFloat() < reflectionChance) {
                    /* Glossy reflection (untinted!) */
                    *scattered = Ray(rec.Point, um_reflect(ray.Direction, roughNormal), ray.Time);
                    *reflectance = f3(1);
                } else {
                    if (metallic > 0 && rng.NextFloat() < metallic) {
                        /* Rough metal */
                        *scattered = Ray(rec.Point, um_reflect(ray.Direction, roughNormal), ray.Time);
                    } else {
                        /* Lambertian diffuse */
                        *scattered = Ray(rec.Point, rng.OnCosineWeightedHemisphere(rec.Normal), ray.Time);
                    }
                }

                /* Scatter type choices */
                if (reflectionChance > 0 && reflectionChance < 1) rng.RandomEvents++;
                if (metallic > 0 && metallic < 1) rng.RandomEvents++;

                /* Random lobe sizes */
                rng.RandomEvents += roughness * (reflectionChance + (1 - reflectionChance) * metallic);
                rng.RandomEvents += (1 - reflectionChance) * (1 - metallic);
                break;
            }
            case RTOW_MATERIAL_DIELECTRIC: {
                const float roughness = 1 - Glossiness.SampleScalar(rec.TexCoords);
                const float3 roughNormal = um_normalize(rec.Normal + roughness * rng.NextFloat3Direction());

                float niOverNt, cosine;
                float3 outwardRoughNormal;
                if (um_dot(ray.Direction, roughNormal) > 0) {
                    outwardRoughNormal = -roughNormal;
                    niOverNt = IndexOfRefraction();
                    cosine = IndexOfRefraction() * um_dot(ray.Direction, roughNormal);
                } else {
                    outwardRoughNormal = roughNormal;
                    niOverNt = 1 / IndexOfRefraction();
                    cosine = -um_dot(ray.Direction, roughNormal);
                }

                float3 scatterDirection;
                float3 refracted;
                if (Refract(ray.Direction, outwardRoughNormal, niOverNt, &refracted) &&
                    rng.NextFloat() > Schlick(cosine, IndexOfRefraction())) {
                    scatterDirection = refracted;
                } else {
                    scatterDirection = um_reflect(ray.Direction, roughNormal);
                    *reflectance = f3(1);
                }

                *scattered = Ray(rec.Point, scatterDirection, ray.Time);

                /* Scatter type choices */
                rng.RandomEvents++;
                /* Random lobe sizes */
                rng.RandomEvents += roughness;
                break;
            }
            case RTOW_MATERIAL_PROBABILISTIC_VOLUME:
                *scattered = Ray(rec.Point, rng.NextFloat3Direction()); /* Time := 0 (:164) */
                rng.RandomEvents += 2;
                break;
            default:
                *scattered = Ray();
                break;
        }
    }
};

/* RT/EntityTypes/Sphere.cs:6-24 */
struct Sphere {
    float SquaredRadius, Radius;
    AABB Bounds() const { const float a = fabsf(Radius); return AABB{f3(-a), f3(a)}; }
};

/* RT/HitTests.cs:9-21 */
inline bool HitAabb(const AABB& aabb, float3 rayOrigin, float3 rayInvDirection)
{
    const float3 t0 = (aabb.Min - rayOrigin) * rayInvDirection;
    const float3 t1 = (aabb.Max - rayOrigin) * rayInvDirection;
    const float tMin = um_max(0.0f, um_cmax(um_min(t0, t1)));
    const float tMax = um_cmin(um_max(t0, t1));
    return tMin < tMax;
}

/* RT/HitTests.cs:23-60 */
inline bool HitSphere(const Sphere& s, const Ray& r, float tMin, float tMax, float* distance, float3* normal)
{
    const float squaredRadius = s.SquaredRadius;
    const float radius = s.Radius;

    const float3 oc = r.Origin;
    const float a = um_dot(r.Direction, r.Direction);
    const float b = um_dot(oc, r.Direction);
    const float c = um_dot(oc, oc) - squaredRadius;
    const float discriminant = b * b - a * c;

    if (discriminant > 0) {
        const float sqrtDiscriminant = sqrtf(discriminant);
        float t = (-b - sqrtDiscriminant) / a;
        if (t < tMax && t > tMin) {
            *distance = t;
            *normal = r.GetPoint(t) / radius;
            return true;
        }
        t = (-b + sqrtDiscriminant) / a;
        if (t < tMax && t > tMin) {
            *distance = t;
            *normal = r.GetPoint(t) / radius;
            return true;
        }
    }
    *distance = 0;
    *normal = f3(0);
    return false;
}

/* RT/EntityTypes/Rect.cs:7-20: an axis-aligned rectangle in the XY plane */
struct Rect {
    float2 From, To;
    static Rect Make(float2 size) { return Rect{float2{-size.x / 2, -size.y / 2}, float2{size.x / 2, size.y / 2}}; }
    AABB Bounds() const { return AABB{f3(From.x, From.y, -0.001f), f3(To.x, To.y, 0.001f)}; }
};
/* RT/EntityTypes/Box.cs:6-18 */
struct Box {
    float3 InverseExtents, Extents;
    static Box Make(float3 size) { Box b; b.Extents = size / 2; b.InverseExtents = f3(1 / b.Extents.x, 1 / b.Extents.y, 1 / b.Extents.z); return b; }
    AABB Bounds() const { return AABB{-Extents, Extents}; }
};
/* RT/EntityTypes/Triangle.cs:6-51: Data = {v2 - v0, v1 - v0, v0}, per-vertex normals, per-vertex uv */
struct Triangle {
    float3 Data[3];
    float3 Normals[3];
    float2 TextureCoordinates[3];
    AABB Bounds() const                                                                   /* :37-49 */
    {
        const float3 vertices[3] = {Data[2], Data[1] + Data[2], Data[0] + Data[2]};
        float3 negativeOffset[3], positiveOffset[3];
        for (int i = 0; i < 3; i++) {
            const float3 absN = f3(fabsf(Normals[i].x), fabsf(Normals[i].y), fabsf(Normals[i].z));
            negativeOffset[i] = vertices[i] - absN * 0.001f;
            positiveOffset[i] = vertices[i] + absN * 0.001f;
        }
        return AABB{um_min(um_min(negativeOffset[0], negativeOffset[1]), negativeOffset[2]),
                    um_max(um_max(positiveOffset[0], positiveOffset[1]), positiveOffset[2])};
    }
};

inline float um_sign(float x) { return (x > 0.0f ? 1.0f : 0.0f) - (x < 0.0f ? 1.0f : 0.0f); } /* math.sign */

/* RT/HitTests.cs:62-78 */
inline bool HitRect(const Rect& rect, const Ray& r, float tMin, float tMax, float* distance, float3* normal)
{
    *distance = 0;
    *normal = f3(0);
    if (r.Direction.z >= 0) return false;
    const float t = -r.Origin.z / r.Direction.z;
    if (t < tMin || t > tMax) return false;
    const float2 xy = float2{r.Origin.x + t * r.Direction.x, r.Origin.y + t * r.Direction.y};
    if (xy.x < rect.From.x || xy.y < rect.From.y || xy.x > rect.To.x || xy.y > rect.To.y) return false;
    *distance = t;
    *normal = f3(0, 0, 1);
    return true;
}

/* RT/HitTests.cs:80-113 (Majercik et al. ray-box); ray direction is assumed to be normalized */
inline bool HitBox(const Box& box, Ray r, float tMin, float tMax, float* distance, float3* normal)
{
    r = Ray(r.Origin + r.Direction * tMin, r.Direction, r.Time);
    *distance = 0;
    *normal = f3(0);
    const float3 rayDirection = r.Direction;
    const float3 ao = f3(fabsf(r.Origin.x), fabsf(r.Origin.y), fabsf(r.Origin.z)) * box.InverseExtents;
    const float winding = um_cmax(ao) < 1 ? -1.0f : 1.0f;
    float3 sgn = -f3(um_sign(rayDirection.x), um_sign(rayDirection.y), um_sign(rayDirection.z));
    const float3 distanceToPlane = (box.Extents * winding * sgn - r.Origin) / rayDirection;
    const bool tx = distanceToPlane.x >= 0 &&
                    fabsf(r.Origin.y + rayDirection.y * distanceToPlane.x) < box.Extents.y && fabsf(r.Origin.z + rayDirection.z * distanceToPlane.x) < box.Extents.z;
    const bool ty = distanceToPlane.y >= 0 &&
                    fabsf(r.Origin.z + rayDirection.z * distanceToPlane.y) < box.Extents.z && fabsf(r.Origin.x + rayDirection.x * distanceToPlane.y) < box.Extents.x;
    const bool tz = distanceToPlane.z >= 0 &&
                    fabsf(r.Origin.x + rayDirection.x * distanceToPlane.z) < box.Extents.x && fabsf(r.Origin.y + rayDirection.y * distanceToPlane.z) < box.Extents.y;
    sgn = tx ? f3(sgn.x, 0, 0) : ty ? f3(0, sgn.y, 0) : f3(0, 0, tz ? sgn.z : 0);
    const bool nzx = sgn.x != 0, nzy = sgn.y != 0, nzz = sgn.z != 0;
    if (!(nzx || nzy || nzz)) return false;
    *distance = nzx ? distanceToPlane.x : nzy ? distanceToPlane.y : distanceToPlane.z;
    *distance += tMin;
    if (*distance > tMax) return false;
    *normal = sgn;
    return true;
}

/* RT/HitTests.cs:115-150 (Moeller-Trumbore, no back-face culling) */
inline bool HitTriangle(const Triangle& tri, const Ray& r, float tMin, float tMax, float* distance, float3* normal, float2* texCoord)
{
    *distance = 0;
    *normal = f3(0);
    *texCoord = float2{0, 0};
    const float3 pvec = um_cross(r.Direction, tri.Data[0]);
    const float det = um_dot(tri.Data[1], pvec);
    if (det == 0) return false;
    const float invDet = 1 / det;
    const float3 tvec = r.Origin - tri.Data[2];
    const float u = um_dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    const float3 qvec = um_cross(tvec, tri.Data[1]);
    const float v = um_dot(r.Direction, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    *distance = um_dot(tri.Data[0], qvec) * invDet;
    if (*distance < tMin || *distance > tMax) return false;
    const float3 bary = f3(1 - u - v, u, v);
    *normal = tri.Normals[0] * bary.x + tri.Normals[1] * bary.y + tri.Normals[2] * bary.z;
    const float2 t0 = tri.TextureCoordinates[0], t1 = tri.TextureCoordinates[1], t2 = tri.TextureCoordinates[2];
    *texCoord = float2{t0.x * bary.x + t1.x * bary.y + t2.x * bary.z, t0.y * bary.x + t1.y * bary.y + t2.y * bary.z};
    return true;
}

/* RT/Entity.cs:27-127 */
struct Entity {
    int Type;
    bool Moving;
    RigidTransform OriginTransform, InverseTransform;
    float3 DestinationOffset;
    float2 TimeRange;
    const Material* MaterialPtr;
    Sphere SphereContent;   /* void* Content (RT/Entity.cs:37), one of: */
    Rect RectContent;
    Box BoxContent;
    Triangle TriangleContent;
    int SourceIndex; /* oracle-only: index in the caller's entity array */

    /* :124-127 */
    RigidTransform TransformAtTime(float t) const
    {
        return RigidTransform{OriginTransform.rot,
                              OriginTransform.pos +
                              DestinationOffset * um_clamp(um_unlerp(TimeRange.x, TimeRange.y, t), 0.0f, 1.0f)};
    }
    /* :105-122 */
    bool HitContent(const Ray& r, float tMin, float tMax, float* distance, float3* normal, float2* texCoord) const
    {
        *texCoord = float2{0, 0};
        switch (Type) {
            case RTOW_ENTITY_SPHERE: return HitSphere(SphereContent, r, tMin, tMax, distance, normal);
            case RTOW_ENTITY_RECT: return HitRect(RectContent, r, tMin, tMax, distance, normal);
            case RTOW_ENTITY_BOX: return HitBox(BoxContent, r, tMin, tMax, distance, normal);
            case RTOW_ENTITY_TRIANGLE: return HitTriangle(TriangleContent, r, tMin, tMax, distance, normal, texCoord);
            default:
                *distance = 0;
                *normal = f3(0);
                return false;
        }
    }
    /* :74-103 */
    bool HitInternal(const Ray& ray, float tMin, float tMax, float* distance, float3* entitySpaceNormal, float2* texCoord,
                     RigidTransform* transformAtTime) const
    {
        RigidTransform inverseTransform;
        if (!Moving) {
            *transformAtTime = OriginTransform;
            inverseTransform = InverseTransform;
        } else {
            *transformAtTime = TransformAtTime(ray.Time);
            inverseTransform = um_inverse(*transformAtTime);
        }
        Ray entitySpaceRay;
        if (Type == RTOW_ENTITY_TRIANGLE)
            entitySpaceRay = ray;
        else
            entitySpaceRay = Ray(um_transform(inverseTransform, ray.Origin), um_rotate(inverseTransform, ray.Direction));
        return HitContent(entitySpaceRay, tMin, tMax, distance, entitySpaceNormal, texCoord);
    }
    /* :58-72 */
    bool Hit(const Ray& ray, float tMin, float tMax, HitRecord* rec) const
    {
        float distance;
        float3 entityLocalNormal;
        float2 texCoord;
        RigidTransform transformAtTime;
        if (HitInternal(ray, tMin, tMax, &distance, &entityLocalNormal, &texCoord, &transformAtTime)) {
            rec->Distance = distance;
            rec->Point = ray.GetPoint(distance);
            rec->Normal = um_normalize(um_rotate(transformAtTime, entityLocalNormal));
            rec->TexCoords = texCoord;
            rec->EntityPtr = nullptr;
            return true;
        }
        memset(rec, 0, sizeof(*rec));
        return false;
    }
};

/* RT/BvhNode.cs:5-22 (index-linked instead of pointer-linked) */
struct BvhNode {
    AABB Bounds;
    int Left, Right;       /* node indices, -1 = null */
    int EntitiesStart;     /* index into bvhEntities, -1 = null */
    int EntityCount;
    bool IsLeaf() const { return EntitiesStart >= 0; }
};

/* RT/View.cs:8-48 (fields only; the ctor runs on the host) */
struct View {
    float3 Origin, LowerLeftCorner, Horizontal, Vertical, Forward, Up, Right;
    float LensRadius;
    /* :38-48 */
    Ray GetRay(float2 normalizedCoordinates, RandomSource& rng) const
    {
        float2 rd;
        if (LensRadius == 0) rd = float2{0, 0};
        else { const float2 d = rng.InUnitDisk(); rd = float2{LensRadius * d.x, LensRadius * d.y}; }
        const float3 offset = Right * rd.x + Up * rd.y;
        const float3 origin = Origin + offset;
        const float3 direction = um_normalize(LowerLeftCorner - offset +
                                              normalizedCoordinates.x * Horizontal +
                                              normalizedCoordinates.y * Vertical);
        const float time = rng.NextFloat();
        return Ray(origin, direction, time);
    }
};

/* ===================================================================================================
 * Cubemap (RT/Texture.cs:141-211): point-sampled sky cube, faces +X -X +Y -Y +Z -Z contiguous in memory.
 * =================================================================================================== */
struct Cubemap {
    int halfFaceSizeX = 0, halfFaceSizeY = 0, faceSizeMinusOneX = 0, faceSizeMinusOneY = 0;
    int pixelStrideX = 0, pixelStrideY = 0;      /* pixelStrideVector = (pixelStride, pixelStride * faceSize.x) */
    int channelType = RTOW_CUBEMAP_SIGNED_HALF;
    int faceStride = 0;
    const uint8_t* dataPointer = nullptr;

    void Set(const RtowCubemapDesc& d, const uint8_t* data)                               /* ctor, :150-169 */
    {
        halfFaceSizeX = d.faceWidth / 2; halfFaceSizeY = d.faceHeight / 2;
        faceSizeMinusOneX = d.faceWidth - 1; faceSizeMinusOneY = d.faceHeight - 1;
        channelType = d.channelType;
        pixelStrideX = d.pixelStride; pixelStrideY = d.pixelStride * d.faceWidth;
        faceStride = d.pixelStride * d.faceWidth * d.faceHeight;
        dataPointer = data;
    }
    float3 Sample(float3 vector) const                                                    /* :171-210 */
    {
        if (dataPointer == nullptr) return f3(0);
        /* indexing math adapted from https://scalibq.wordpress.com/2013/06/23/cubemaps/ (reference comment) */
        const float absVector[4] = {fabsf(vector.x), fabsf(vector.y), fabsf(vector.z), 0.0f};
        const float maxDistance = um_max(um_max(um_max(absVector[0], absVector[1]), absVector[2]), absVector[3]);   /* cmax(float4) */
        int laneMask = 0;
        for (int i = 0; i < 4; i++) if (maxDistance == absVector[i]) laneMask |= 1 << i;                             /* bitmask(maxDistance == absVector) */
        if (laneMask == 0) return f3(0); /* NaN direction: tzcnt(0) = 32 indexes out of the vector in the reference (undefined); black here */
        int firstLane = 0;
        while (!((laneMask >> firstLane) & 1)) firstLane++;                                                           /* tzcnt */
        if (firstLane > 2) firstLane = 0; /* the zero vector: lane 3 can only be first when all are 0, where lane 0 is first anyway */
        const float v[3] = {vector.x, vector.y, vector.z};
        const bool positive = v[firstLane] >= 0;
        float2 uv;
        switch (firstLane) {
            case 0: uv = float2{positive ? -vector.z : vector.z, -vector.y}; break;       /* x */
            case 1: uv = float2{vector.x, positive ? vector.z : -vector.z}; break;        /* y */
            default: uv = float2{positive ? vector.x : -vector.x, -vector.y}; break;      /* z */
        }
        uv.x = uv.x / absVector[firstLane];
        uv.y = uv.y / absVector[firstLane];
        int cx = (int)((uv.x + 1) * (float)halfFaceSizeX), cy = (int)((uv.y + 1) * (float)halfFaceSizeY);
        cx = cx < faceSizeMinusOneX ? cx : faceSizeMinusOneX;                              /* min((int2) ..., faceSizeMinusOne) */
        cy = cy < faceSizeMinusOneY ? cy : faceSizeMinusOneY;
        const uint8_t* pFaceData = dataPointer + (size_t)(firstLane * 2 + (positive ? 0 : 1)) * (size_t)faceStride;
        pFaceData += cx * pixelStrideX + cy * pixelStrideY;                                /* dot(coords, pixelStrideVector) */
        switch (channelType) {
            case RTOW_CUBEMAP_UNSIGNED_BYTE:
                return f3((float)pFaceData[0], (float)pFaceData[1], (float)pFaceData[2]) / 255.0f;
            default: {
                uint16_t t[3];
                memcpy(t, pFaceData, 6);
                return f3(half_to_float(t[0]), half_to_float(t[1]), half_to_float(t[2]));
            }
        }
    }
};

/* ===================================================================================================
 * Scene: entity/material buffers + the reference's BVH builder
 * =================================================================================================== */
struct BvhBuildingEntity { int entity; AABB Bounds; };

struct OracleScene {
    std::vector<Material> materials;
    std::vector<Entity> entities;      /* entityBuffer */
    std::vector<Entity> bvhEntities;   /* entities re-ordered by the builder (UNITY/BvhNodeData.cs:157-160) */
    std::vector<BvhNode> nodes;        /* node 0 = root */
    int maxDepthSeen = 0;
    bool unsupported = false;
    struct ImageCopy { int width, height, pixelStride; std::vector<uint8_t> pixels; };
    std::vector<ImageCopy> images;     /* pixel data of Image textures (the host's Texture2D data) */
    Cubemap skyCubemap;                /* Environment.SkyCubemap (RT/Environment.cs:16); set by oracle_scene_set_cubemap */
    std::vector<uint8_t> skyCubemapData;
    /* the host's noise texture sets (UNITY/BlueNoiseData.cs, UNITY/SpatioTemporalBlueNoiseData.cs); set by oracle_scene_set_*_noise */
    uint32_t blueRowStride = 0, blueTextureCount = 0, stbRowStride = 0, stbTextureCount = 0;
    std::vector<uint16_t> blueTexels;
    std::vector<uint8_t> stbScalar, stbVector2, stbCosine, stbUnit2, stbUnit3;
    bool NoiseFor(int noiseColor, int textureIndex, NoiseTextures* out) const
    {
        *out = NoiseTextures();
        if (noiseColor == RTOW_NOISE_BLUE) {
            if (textureIndex < 0 || (uint32_t)textureIndex >= blueTextureCount) return false;
            out->blueRowStride = blueRowStride;
            out->blue = blueTexels.data() + (size_t)textureIndex * blueRowStride * blueRowStride * 4;
        } else if (noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) {
            if (textureIndex < 0 || (uint32_t)textureIndex >= stbTextureCount) return false;
            const size_t texels = (size_t)stbRowStride * stbRowStride, t = (size_t)textureIndex * texels;
            out->stbRowStride = stbRowStride;
            out->stbScalar = stbScalar.data() + t; out->stbVector2 = stbVector2.data() + t * 3; out->stbCosineUnitVector3 = stbCosine.data() + t * 4;
            out->stbUnitVector2 = stbUnit2.data() + t * 3; out->stbUnitVector3 = stbUnit3.data() + t * 3;
        }
        return true;
    }

    /* UNITY/BvhNodeData.cs:23-81 : world-space bounds of an entity (moving: union of start/end boxes) */
    static AABB EntityBounds(const Entity& e)
    {
        AABB Bounds = e.Type == RTOW_ENTITY_RECT ? e.RectContent.Bounds()
                    : e.Type == RTOW_ENTITY_BOX ? e.BoxContent.Bounds()
                    : e.Type == RTOW_ENTITY_TRIANGLE ? e.TriangleContent.Bounds()
                    : e.SphereContent.Bounds();                                     /* :32-39 */
        const float3 corners[8] = {
            f3(Bounds.Min.x, Bounds.Min.y, Bounds.Min.z), f3(Bounds.Min.x, Bounds.Min.y, Bounds.Max.z),
            f3(Bounds.Min.x, Bounds.Max.y, Bounds.Min.z), f3(Bounds.Max.x, Bounds.Min.y, Bounds.Min.z),
            f3(Bounds.Min.x, Bounds.Max.y, Bounds.Max.z), f3(Bounds.Max.x, Bounds.Max.y, Bounds.Min.z),
            f3(Bounds.Max.x, Bounds.Min.y, Bounds.Max.z), f3(Bounds.Max.x, Bounds.Max.y, Bounds.Max.z)};
        float3 minimum = f3(INFINITY), maximum = f3(-INFINITY);
        if (e.Moving) {
            const float3 destinationPosition = e.OriginTransform.pos + e.DestinationOffset;
            const RigidTransform minTransform{e.OriginTransform.rot, um_min(e.OriginTransform.pos, destinationPosition)};
            const RigidTransform maxTransform{e.OriginTransform.rot, um_max(e.OriginTransform.pos, destinationPosition)};
            for (int i = 0; i < 8; i++) {
                minimum = um_min(minimum, um_transform(minTransform, corners[i]));
                maximum = um_max(maximum, um_transform(maxTransform, corners[i]));
            }
        } else {
            for (int i = 0; i < 8; i++) {
                const float3 c = um_transform(e.OriginTransform, corners[i]);
                minimum = um_min(minimum, c);
                maximum = um_max(maximum, c);
            }
        }
        return AABB{minimum, maximum};
    }

    /* `new Entity(type, content, originTransform, material, moving, destinationOffset, timeRange)` (RT/Entity.cs:39-56)
     * with the Content struct built from the flat description. */
    static bool MakeEntity(const RtowEntity& s, const RtowTriangle* triangles, int triangleCount, Entity* out)
    {
        Entity e{};
        e.Type = s.type;
        e.Moving = s.moving != 0;
        e.OriginTransform = RigidTransform{f4(s.rotation), f3(s.position)};
        e.DestinationOffset = f3(s.destinationOffset);
        e.TimeRange = float2{s.timeRange.x, s.timeRange.y};
        bool ok = true;
        switch (s.type) {
            case RTOW_ENTITY_SPHERE: e.SphereContent = Sphere{s.size.x * s.size.x, s.size.x}; break;          /* Sphere.cs:10-14 */
            case RTOW_ENTITY_RECT: e.RectContent = Rect::Make(float2{s.size.x, s.size.y}); break;            /* Rect.cs:12-16 */
            case RTOW_ENTITY_BOX: e.BoxContent = Box::Make(f3(s.size)); break;                               /* Box.cs:11-15 */
            case RTOW_ENTITY_TRIANGLE:
                if (!triangles || s.contentIndex < 0 || s.contentIndex >= triangleCount) { ok = false; break; }
                for (int k = 0; k < 3; k++) {
                    e.TriangleContent.Data[k] = f3(triangles[s.contentIndex].data[k]);
                    e.TriangleContent.Normals[k] = f3(triangles[s.contentIndex].normals[k]);
                    e.TriangleContent.TextureCoordinates[k] = float2{triangles[s.contentIndex].textureCoordinates[k].x, triangles[s.contentIndex].textureCoordinates[k].y};
                }
                break;
            default: ok = false;
        }
        if (!e.Moving) e.InverseTransform = um_inverse(e.OriginTransform);                                  /* Entity.cs:51-52 */
        else e.InverseTransform = RigidTransform{float4{0, 0, 0, 0}, f3(0)};
        *out = e;
        return ok;
    }

    static float axisOf(float3 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

    /* UNITY/BvhNodeData.cs:122-213.  Returns the index of the node it filled.
     * NativeSlice.Sort is the unstable introsort restated above (unity_sort). */
    int BuildNode(std::vector<BvhBuildingEntity>& ents, int begin, int end, int maxDepth, int depth, int sortAxis)
    {
        const int self = (int)nodes.size();
        nodes.push_back(BvhNode{});
        maxDepthSeen = std::max(maxDepthSeen, depth);

        AABB entireBounds{f3(3.40282347e38f), f3(-3.40282347e38f)}; /* float.MaxValue / float.MinValue */
        for (int i = begin; i < end; i++) entireBounds = AABB::Enclose(entireBounds, ents[i].Bounds);

        int biggestPartition = -1;
        float biggestPartitionSize = -3.40282347e38f;
        const float3 entireSize = entireBounds.Size();
        for (int i = 0; i < 3; i++) {
            const float size = axisOf(entireSize, i);
            if (size > biggestPartitionSize) { biggestPartition = i; biggestPartitionSize = size; }
        }
        if (sortAxis != biggestPartition && biggestPartition >= 0) {
            const int ax = biggestPartition;
            unity_sort(ents.data() + begin, end - begin, [ax](const BvhBuildingEntity& l, const BvhBuildingEntity& r) {
                const float d = axisOf(l.Bounds.Min, ax) - axisOf(r.Bounds.Min, ax);       /* (int) sign(lhs - rhs), :246-249 */
                return d > 0 ? 1 : d < 0 ? -1 : 0;
            });
        }
        const int biggestAxis = biggestPartition;
        const int length = end - begin;

        BvhNode n{};
        if (depth == maxDepth || length <= 1) {
            n.EntitiesStart = (int)bvhEntities.size();
            for (int i = begin; i < end; i++) bvhEntities.push_back(entities[ents[i].entity]);
            if (length > 0) {
                n.Bounds = ents[begin].Bounds;
                for (int i = begin + 1; i < end; i++) n.Bounds = AABB::Enclose(n.Bounds, ents[i].Bounds);
            } else {
                n.Bounds = AABB{f3(0), f3(0)};
            }
            n.EntityCount = length;
            n.Left = n.Right = -1;
            nodes[self] = n;
        } else {
            n.EntitiesStart = -1;
            n.EntityCount = 0;
            int partitionLength = 0;
            const float partitionStart = axisOf(ents[begin].Bounds.Min, biggestAxis);
            for (int i = begin; i < end; i++) {
                partitionLength++;
                const AABB& bounds = ents[i].Bounds;
                if (axisOf(bounds.Min, biggestAxis) - partitionStart > biggestPartitionSize / 2 ||
                    axisOf(bounds.Size(), biggestAxis) > biggestPartitionSize / 2)
                    break;
            }
            if (partitionLength == length) partitionLength--;
            n.Left = BuildNode(ents, begin, begin + partitionLength, maxDepth, depth + 1, biggestPartition);
            n.Right = BuildNode(ents, begin + partitionLength, end, maxDepth, depth + 1, biggestPartition);
            n.Bounds = AABB::Enclose(nodes[n.Left].Bounds, nodes[n.Right].Bounds);
            nodes[self] = n;
        }
        return self;
    }

    void Build(const RtowSceneDesc* d)
    {
        images.clear();
        for (int i = 0; i < d->imageCount && d->images; i++) {
            const RtowImage& im = d->images[i];
            ImageCopy c{im.width, im.height, im.pixelStride, {}};
            if (im.width <= 0 || im.height <= 0 || im.pixelStride < 3 || !im.pixels) unsupported = true;
            else c.pixels.assign(im.pixels, im.pixels + (size_t)im.width * im.height * im.pixelStride);
            images.push_back(std::move(c));
        }
        materials.resize(d->materialCount);
        for (int i = 0; i < d->materialCount; i++) {
            const RtowMaterial& m = d->materials[i];
            auto tex = [this](const RtowTexture& t) {
                if (t.type != RTOW_TEXTURE_NONE && t.type != RTOW_TEXTURE_CONSTANT && t.type != RTOW_TEXTURE_CONSTANT_SCALAR && t.type != RTOW_TEXTURE_IMAGE)
                    unsupported = true;
                Texture r{t.type, f3(t.mainColor), t.parameter, t.scalarValueChannel};
                if (t.type == RTOW_TEXTURE_IMAGE && t.imageIndex >= 0) {
                    if (t.imageIndex >= (int)images.size()) unsupported = true;
                    else { const ImageCopy& im = images[t.imageIndex]; r.ImageSizeX = im.width; r.ImageSizeY = im.height; r.PixelStride = im.pixelStride; r.ImagePointer = im.pixels.data(); }
                }
                return r;
            };
            /* RT/Material.cs:28-46: `parameter` only stored for Dielectric / ProbabilisticVolume */
            float parameter = 0;
            if (m.type == RTOW_MATERIAL_DIELECTRIC || m.type == RTOW_MATERIAL_PROBABILISTIC_VOLUME) parameter = m.parameter;
            materials[i] = Material{m.type, tex(m.albedo), tex(m.glossiness), tex(m.emission), tex(m.metallic), parameter};
        }
        entities.resize(d->entityCount);
        for (int i = 0; i < d->entityCount; i++) {
            const RtowEntity& s = d->entities[i];
            Entity e{};
            if (!MakeEntity(s, d->triangles, d->triangleCount, &e)) unsupported = true;
            e.MaterialPtr = &materials[s.materialIndex];
            e.SourceIndex = i;
            entities[i] = e;
        }
        /* UNITY/Raytracer.cs:1306-1351 */
        std::vector<BvhBuildingEntity> building(entities.size());
        for (size_t i = 0; i < entities.size(); i++) building[i] = BvhBuildingEntity{(int)i, EntityBounds(entities[i])};
        nodes.clear();
        bvhEntities.clear();
        bvhEntities.reserve(entities.size());
        const int maxDepth = d->maxBvhDepth > 0 ? d->maxBvhDepth : 32; /* Assets/Prefabs/Raytracer.prefab default */
        BuildNode(building, 0, (int)building.size(), maxDepth, 0, -1);
    }
};

/* ===================================================================================================
 * SampleBatchJob (JOBS/SampleBatchJob.cs)
 * =================================================================================================== */
struct Diagnostics { float RayCount, BoundsHitCount, CandidateCount, SampleCountWeight; };

struct Counters {
    uint64_t rays = 0, boundsHit = 0, candidates = 0, nodesVisited = 0, hits = 0;
    uint32_t maxNodeStack = 0, maxCandidates = 0, maxHits = 0;
};

struct Job {
    NoiseTextures noise;               /* BlueNoise / StbNoise fields of the job (the texture of this batch) */
    mutable bool tracePixel = false;   /* ORACLE_TRACE_PIXEL=<index>: per-segment trace on stderr (debugging aid) */
    const OracleScene* scene;
    RtowSampleParams p;
    View view;
    const float *InputColor, *InputNormal, *InputAlbedo, *InputSampleCountWeight;
    float *OutputColor, *OutputNormal, *OutputAlbedo, *OutputSampleCountWeight;
    uint8_t* OutputDiagnostics;

    struct Scratch {
        std::vector<float3> emissionStack, attenuationStack;
        std::vector<int> nodeTraversalBuffer;              /* HybridPtrStack<BvhNode>, UTIL/HybridCollections.cs:8-52 */
        std::vector<const Entity*> hitCandidateBuffer;     /* HybridPtrStack<Entity> */
        std::vector<HitRecord> hitRecordBuffer;            /* HybridList<HitRecord>, :54-86 */
        Counters counters;
    };

    /* :403-448 */
    void FindHitCandidates(const Ray& ray, Scratch& s, Diagnostics& diagnostics) const
    {
        float3 rayInvDirection = f3(um_rcp(ray.Direction.x), um_rcp(ray.Direction.y), um_rcp(ray.Direction.z));
        /* Convert NaN to INFINITY (:411-412) */
        if (um_isnan(rayInvDirection.x)) rayInvDirection.x = INFINITY;
        if (um_isnan(rayInvDirection.y)) rayInvDirection.y = INFINITY;
        if (um_isnan(rayInvDirection.z)) rayInvDirection.z = INFINITY;

        s.nodeTraversalBuffer.clear();
        s.hitCandidateBuffer.clear();
        s.nodeTraversalBuffer.push_back(0);

        while (!s.nodeTraversalBuffer.empty()) {
            s.counters.maxNodeStack = std::max<uint32_t>(s.counters.maxNodeStack, (uint32_t)s.nodeTraversalBuffer.size());
            const BvhNode& node = scene->nodes[s.nodeTraversalBuffer.back()];
            s.nodeTraversalBuffer.pop_back();
            s.counters.nodesVisited++;

            if (!HitAabb(node.Bounds, ray.Origin, rayInvDirection)) continue;

            diagnostics.BoundsHitCount++;
            s.counters.boundsHit++;

            if (node.IsLeaf()) {
                for (int i = 0; i < node.EntityCount; i++) s.hitCandidateBuffer.push_back(&scene->bvhEntities[node.EntitiesStart + i]);
                diagnostics.CandidateCount += node.EntityCount;
                s.counters.candidates += node.EntityCount;
            } else {
                s.nodeTraversalBuffer.push_back(node.Left);
                s.nodeTraversalBuffer.push_back(node.Right);
            }
        }
        s.counters.maxCandidates = std::max<uint32_t>(s.counters.maxCandidates, (uint32_t)s.hitCandidateBuffer.size());
    }

    /* :450-475 */
    void FindHits(const Ray& ray, Scratch& s) const
    {
        s.hitRecordBuffer.clear();
        while (!s.hitCandidateBuffer.empty()) {
            const Entity* hitCandidate = s.hitCandidateBuffer.back();
            s.hitCandidateBuffer.pop_back();
            HitRecord thisRec;
            if (hitCandidate->Hit(ray, 0, INFINITY, &thisRec)) {
                thisRec.EntityPtr = hitCandidate;
                s.hitRecordBuffer.push_back(thisRec);

                /* Inject exit hits for probabilistic convex hulls (:462-469); IsConvexHull = Box || Sphere (RT/Entity.cs:22-25) */
                HitRecord exitRec;
                if (hitCandidate->MaterialPtr->Type == RTOW_MATERIAL_PROBABILISTIC_VOLUME &&
                    (hitCandidate->Type == RTOW_ENTITY_BOX || hitCandidate->Type == RTOW_ENTITY_SPHERE) &&
                    hitCandidate->Hit(ray, thisRec.Distance + 0.001f, INFINITY, &exitRec)) {
                    exitRec.EntityPtr = hitCandidate;
                    s.hitRecordBuffer.push_back(exitRec);
                }
            }
        }
        s.counters.hits += s.hitRecordBuffer.size();
        s.counters.maxHits = std::max<uint32_t>(s.counters.maxHits, (uint32_t)s.hitRecordBuffer.size());
        /* hitBuffer.Sort(new HitRecord.DistanceComparer()) (:474, RT/HitRecord.cs:22-25) */
        unity_sort(s.hitRecordBuffer.data(), (int)s.hitRecordBuffer.size(),
                   [](const HitRecord& x, const HitRecord& y) { return float_compare_to(x.Distance, y.Distance); });
    }

    /* :510-524 */
    static bool AnyBackwardsVolumeEntryHit(const Ray& backwardsRay, const Scratch& s)
    {
        for (size_t i = 0; i < s.hitCandidateBuffer.size(); i++) {
            const Entity* hitCandidate = s.hitCandidateBuffer[i];
            HitRecord hitRecord;
            if (hitCandidate->MaterialPtr->Type == RTOW_MATERIAL_PROBABILISTIC_VOLUME &&
                hitCandidate->Hit(backwardsRay, 0, INFINITY, &hitRecord) &&
                um_dot(hitRecord.Normal, backwardsRay.Direction) > 0)
                return true;
        }
        return false;
    }

    /* :477-508.  The backwards probe overwrites the node / candidate buffers; the hit list is untouched. */
    const Material* DetermineVolumeContainment(const Ray& ray, Scratch& s, Diagnostics& diagnostics) const
    {
        for (size_t i = 0; i < s.hitRecordBuffer.size(); i++) {
            const HitRecord& hit = s.hitRecordBuffer[i];
            if (hit.EntityPtr->MaterialPtr->Type == RTOW_MATERIAL_PROBABILISTIC_VOLUME) {
                /* Entry hit, early out */
                if (um_dot(hit.Normal, ray.Direction) < 0) break;
                /* Exit hit before an entry hit, we are likely inside this volume; throw a ray backwards to make sure */
                const Ray backwardsRay(ray.Origin, -ray.Direction, ray.Time);
                FindHitCandidates(backwardsRay, s, diagnostics);
                if (AnyBackwardsVolumeEntryHit(backwardsRay, s)) return hit.EntityPtr->MaterialPtr;
            }
        }
        return nullptr;
    }

    /* :166-401 */
    bool Sample(const Ray& eyeRay, RandomSource& rng, Scratch& s, float3* sampleColor, float3* sampleNormal, float3* sampleAlbedo,
                Diagnostics& diagnostics, float* randomEventsAcc) const
    {
        size_t cursor = 0; /* emissionCursor / attenuationCursor */
        float randomEventsLocalAcc = 0;
        int depth = 0;
        bool firstNonSpecularHit = false;
        *sampleColor = *sampleNormal = *sampleAlbedo = f3(0);
        const Material* currentProbabilisticVolumeMaterial = nullptr;                                /* :180 */
        const int TraceDepth = p.traceDepth;

        Ray ray = eyeRay;

        for (; depth < TraceDepth; depth++) {
            FindHitCandidates(ray, s, diagnostics);
            FindHits(ray, s);
            if (currentProbabilisticVolumeMaterial == nullptr)                                        /* :194-201 */
                currentProbabilisticVolumeMaterial = DetermineVolumeContainment(ray, s, diagnostics);

            diagnostics.RayCount++;
            s.counters.rays++;
            if (tracePixel) { for (size_t i = 0; i < s.hitRecordBuffer.size(); i++) fprintf(stderr, "[otrace]   hit %zu prim %d t %.9g entry %d\n", i, s.hitRecordBuffer[i].EntityPtr->SourceIndex, s.hitRecordBuffer[i].Distance, um_dot(s.hitRecordBuffer[i].Normal, ray.Direction) < 0 ? 1 : 0);
                fprintf(stderr, "[otrace]   contained %d o %.9g %.9g %.9g d %.9g %.9g %.9g\n", currentProbabilisticVolumeMaterial ? (int)(currentProbabilisticVolumeMaterial - &scene->materials[0]) : -1, ray.Origin.x, ray.Origin.y, ray.Origin.z, ray.Direction.x, ray.Direction.y, ray.Direction.z); }

            int hitIndex = 0;
            int hitCount = (int)s.hitRecordBuffer.size();
            while (hitIndex < hitCount) {
                HitRecord rec = s.hitRecordBuffer[hitIndex];
                const Material* material = rec.EntityPtr->MaterialPtr;

                if (currentProbabilisticVolumeMaterial != nullptr ||                                  /* Inside a volume */
                    material->Type == RTOW_MATERIAL_PROBABILISTIC_VOLUME) {                           /* Entering a volume (:212-303) */
                    const bool isEntryHit = currentProbabilisticVolumeMaterial == nullptr;
                    if (currentProbabilisticVolumeMaterial == nullptr) currentProbabilisticVolumeMaterial = material;

                    /* Look for an obstacle or an exit hit */
                    int exitHitIndex = hitIndex;
                    int lastExitIndex = -1;
                    int sameMaterialEntries = 0;
                    while (exitHitIndex < hitCount) {
                        const HitRecord& hit = s.hitRecordBuffer[exitHitIndex];
                        if (hit.EntityPtr->MaterialPtr == currentProbabilisticVolumeMaterial) {
                            if (um_dot(hit.Normal, ray.Direction) < 0)
                                sameMaterialEntries++;
                            else {
                                sameMaterialEntries--;
                                lastExitIndex = exitHitIndex;
                            }
                            if (sameMaterialEntries <= 0) break;
                        } else
                            break;
                        exitHitIndex++;
                    }
                    if (sameMaterialEntries > 0 && lastExitIndex != -1) exitHitIndex = lastExitIndex;

                    if (exitHitIndex < hitCount) {
                        const HitRecord exitHitRecord = s.hitRecordBuffer[exitHitIndex];
                        float distanceInProbabilisticVolume = exitHitRecord.Distance;
                        float probabilisticVolumeEntryDistance = 0;
                        if (isEntryHit) {
                            /* Factor in entry distance */
                            probabilisticVolumeEntryDistance = rec.Distance;
                            distanceInProbabilisticVolume -= rec.Distance;
                        }
                        if (currentProbabilisticVolumeMaterial->ProbabilisticHit(&distanceInProbabilisticVolume, rng)) {
                            /* We hit inside the volume; hijack the current hit record's distance and material */
                            const float totalDistance = probabilisticVolumeEntryDistance + distanceInProbabilisticVolume;
                            HitRecord inside{};
                            inside.Distance = totalDistance;
                            inside.Point = ray.GetPoint(totalDistance);
                            inside.Normal = -ray.Direction;
                            inside.TexCoords = float2{0, 0};
                            inside.EntityPtr = nullptr;
                            rec = inside;
                            material = currentProbabilisticVolumeMaterial;
                        } else {
                            /* No hit inside the volume, exit it */
                            currentProbabilisticVolumeMaterial = nullptr;
                            if (exitHitRecord.EntityPtr->MaterialPtr->Type == RTOW_MATERIAL_PROBABILISTIC_VOLUME &&
                                um_dot(exitHitRecord.Normal, ray.Direction) > 0) {
                                /* Volume exit, move to next hit */
                                hitIndex = exitHitIndex + 1;
                                continue;
                            }
                            /* Obstacle, continue */
                            rec = exitHitRecord;
                            material = rec.EntityPtr->MaterialPtr;
                        }
                    } else {
                        /* No more surfaces to hit (probabilistic volume has holes) */
                        s.hitRecordBuffer.clear();
                        hitCount = 0;
                        break;
                    }
                }

                if (tracePixel) fprintf(stderr, "[otrace] depth %d kind %d prim %d t %.9g curVol %d nHits %d rng %u\n", depth, rec.EntityPtr ? 0 : 1, rec.EntityPtr ? rec.EntityPtr->SourceIndex : 65535, rec.Distance,
                                        currentProbabilisticVolumeMaterial ? (int)(currentProbabilisticVolumeMaterial - &scene->materials[0]) : -1, hitCount, rng.whiteNoise.state);
                float3 albedo;
                Ray scatteredRay;
                material->Scatter(ray, rec, rng, &albedo, &scatteredRay);                    /* :308 */

                const float3 emission = material->Emit(rec.TexCoords);                                    /* :310 */
                s.emissionStack[cursor] = emission;                                          /* :311 */

                if (depth == 0) *sampleNormal = rec.Normal;                                  /* :313-314 */

                if (!firstNonSpecularHit) {                                                  /* :316-328 */
                    if (material->IsPerfectSpecular()) {
                    } else {
                        *sampleAlbedo = emission + albedo;
                        *sampleNormal = rec.Normal;
                        firstNonSpecularHit = true;
                    }
                }

                s.attenuationStack[cursor] = albedo;                                         /* :330 */
                cursor++;

                randomEventsLocalAcc += rng.RandomEvents / dm_powf(2, (float)depth);          /* :332 */
                rng.RandomEvents = 0;

                ray = scatteredRay;                                                          /* :335-336 */
                ray = ray.OffsetTowards(um_dot(scatteredRay.Direction, rec.Normal) >= 0 ? rec.Normal : -rec.Normal);
                break;
            }

            /* No hit? (:341-374) */
            if (hitIndex >= hitCount) {
                if (tracePixel) fprintf(stderr, "[otrace] depth %d kind 2 sky curVol %d rng %u\n", depth, currentProbabilisticVolumeMaterial ? (int)(currentProbabilisticVolumeMaterial - &scene->materials[0]) : -1, rng.whiteNoise.state);
                float3 hitSkyColor = f3(0);
                switch (p.environment.skyType) {
                    case RTOW_SKY_GRADIENT:
                        hitSkyColor = um_lerp(f3(p.environment.skyBottomColor), f3(p.environment.skyTopColor),
                                              0.5f * (ray.Direction.y + 1));
                        break;
                    case RTOW_SKY_CUBEMAP:                                                   /* :356-358 */
                        hitSkyColor = scene->skyCubemap.Sample(ray.Direction);
                        break;
                }
                s.emissionStack[cursor] = hitSkyColor;
                s.attenuationStack[cursor] = f3(1);
                cursor++;
                randomEventsLocalAcc += rng.RandomEvents / dm_powf(2, (float)depth);
                rng.RandomEvents = 0;

                if (!firstNonSpecularHit) {
                    *sampleAlbedo = hitSkyColor;
                    *sampleNormal = -ray.Direction;
                }
                break;
            }
        }

        *sampleColor = f3(0);

        /* Safety : fail this sample if the trace depth limit is reached (:379-381) */
        if (depth == TraceDepth) return false;

        /* Attenuate colors from the tail of the hit stack to the head (:384-396) */
        while (cursor != 0) {
            --cursor;
            const float3 a = s.attenuationStack[cursor];
            const float3 e = s.emissionStack[cursor];
            *sampleColor *= a;
            *sampleColor += e;
        }

        *randomEventsAcc += randomEventsLocalAcc;                                            /* :398 */
        return true;
    }

    /* :59-164 */
    void Execute(int index, Scratch& s) const
    {
        { static const int tracedPixel = getenv("ORACLE_TRACE_PIXEL") ? atoi(getenv("ORACLE_TRACE_PIXEL")) : -1; tracePixel = tracedPixel == index; }
        const int width = (int)p.size.x;
        const int cx = index % width; /* column */
        const int cy = index / width; /* row */

        if (cy % p.sliceDivider != p.sliceOffset) return;                                    /* :69-70 */

        const float* lastColor = InputColor + 4 * (size_t)index;                             /* :72-78 */
        float3 colorAcc = f3(lastColor[0], lastColor[1], lastColor[2]);
        float3 normalAcc = f3(InputNormal[3 * (size_t)index], InputNormal[3 * (size_t)index + 1], InputNormal[3 * (size_t)index + 2]);
        float3 albedoAcc = f3(InputAlbedo[3 * (size_t)index], InputAlbedo[3 * (size_t)index + 1], InputAlbedo[3 * (size_t)index + 2]);
        float sampleCountWeightAcc = InputSampleCountWeight[index];
        int sampleCount = (int)lastColor[3];

        /* :91  new Random((Seed * 0x8C4CA03Fu) ^ (uint)(index * 0x7383ED49u)) */
        RandomSource rng;
        rng.noiseColor = p.noiseColor;
        rng.tex = &noise;
        rng.whiteNoise.state = 0;                                                              /* `Random whiteNoise = default` (:80) */
        switch (p.noiseColor) {                                                                /* :81-93 */
            case RTOW_NOISE_BLUE:
            case RTOW_NOISE_SPATIOTEMPORAL_BLUE:
                rng.SetCoordinates(p.seed, (uint32_t)cx, (uint32_t)cy);                        /* seed = frameSeed = Seed (UNITY/Raytracer.cs:696-703) */
                break;
            default:
                rng.whiteNoise.Init((p.seed * 0x8C4CA03Fu) ^ ((uint32_t)index * 0x7383ED49u));
                break;
        }
        rng.RandomEvents = 0;
        rng.draws = 0;

        if (s.emissionStack.size() < (size_t)p.traceDepth + 1) {                              /* stackalloc :103-104 */
            s.emissionStack.resize((size_t)p.traceDepth + 1);
            s.attenuationStack.resize((size_t)p.traceDepth + 1);
        }

        float3 fallbackAlbedo = f3(0), fallbackNormal = f3(0);
        Diagnostics diagnostics{0, 0, 0, 0};

        /* :118-126 */
        uint32_t samplesToAccumulate;
        const float sampleCountWeight = sampleCountWeightAcc / (float)sampleCount;
        if (sampleCountWeight == 0)
            samplesToAccumulate = p.sampleCountRange[0];
        else {
            const float normalizedSampleCountWeight =
                um_saturate(um_unlerp(p.sampleCountWeightExtrema.x, p.sampleCountWeightExtrema.y, sampleCountWeight));
            samplesToAccumulate =
                (uint32_t)um_round(um_lerp((float)p.sampleCountRange[0], (float)p.sampleCountRange[1], normalizedSampleCountWeight));
        }
        diagnostics.SampleCountWeight = sampleCountWeight;                                   /* :128-130 */

        /* RTOW_RNG_PER_SAMPLE (include/rtow.h; NOT the reference): every sample has its own generator; groups of 16 samples are summed
         * from zero in sample order and the groups added to the accumulators in group order */
        const bool perSample = p.rngPolicy == RTOW_RNG_PER_SAMPLE || p.rngPolicy == RTOW_RNG_PER_SAMPLE_XOROSHIRO;
        float3 gColor = f3(0), gNormal = f3(0), gAlbedo = f3(0);
        float gWeight = 0;
        int gCount = 0;
        auto closeGroup = [&]() {
            if (gCount > 0) { colorAcc += gColor; normalAcc += gNormal; albedoAcc += gAlbedo; sampleCount += gCount; }
            sampleCountWeightAcc += gWeight;
            gColor = gNormal = gAlbedo = f3(0);
            gWeight = 0;
            gCount = 0;
        };

        for (uint32_t smp = 0; smp < samplesToAccumulate; smp++) {                            /* :132-157 */
            if (tracePixel) fprintf(stderr, "[otrace] sample %u\n", smp);
            if (perSample) {
                uint32_t state = ((p.seed * 0x8C4CA03Fu) ^ ((uint32_t)index * 0x7383ED49u)) ^ ((smp + 1u) * 0x9E3779B9u);
                if (p.rngPolicy == RTOW_RNG_PER_SAMPLE_XOROSHIRO) rng.whiteNoise.InitXoroshiro(state);
                else { if (state == 0) state = 0x9E3779B9u; rng.whiteNoise.Init(state); }
                rng.RandomEvents = 0;
            }
            float2 jitter;
            if (p.subPixelJitter) jitter = rng.NextFloat2();
            else jitter = float2{0.5f, 0.5f};
            const float2 normalizedCoordinates = float2{((float)cx + jitter.x) / p.size.x, ((float)cy + jitter.y) / p.size.y};
            const Ray eyeRay = view.GetRay(normalizedCoordinates, rng);

            float3 sampleColor, sampleNormal, sampleAlbedo;
            if (perSample) {
                if (Sample(eyeRay, rng, s, &sampleColor, &sampleNormal, &sampleAlbedo, diagnostics, &gWeight)) {
                    gColor += sampleColor;
                    gNormal += sampleNormal;
                    gAlbedo += sampleAlbedo;
                    gCount++;
                }
                if (smp % 16 == 15 || smp + 1 == samplesToAccumulate) closeGroup();
            } else if (Sample(eyeRay, rng, s, &sampleColor, &sampleNormal, &sampleAlbedo, diagnostics, &sampleCountWeightAcc)) {
                colorAcc += sampleColor;
                normalAcc += sampleNormal;
                albedoAcc += sampleAlbedo;
                sampleCount++;
            }
            if (smp == 0) {
                fallbackNormal = sampleNormal;
                fallbackAlbedo = sampleAlbedo;
            }
        }

        float* oc = OutputColor + 4 * (size_t)index;                                         /* :159-163 */
        oc[0] = colorAcc.x; oc[1] = colorAcc.y; oc[2] = colorAcc.z; oc[3] = (float)sampleCount;
        const float3 on = sampleCount == 0 ? fallbackNormal : normalAcc;
        const float3 oa = sampleCount == 0 ? fallbackAlbedo : albedoAcc;
        OutputNormal[3 * (size_t)index] = on.x; OutputNormal[3 * (size_t)index + 1] = on.y; OutputNormal[3 * (size_t)index + 2] = on.z;
        OutputAlbedo[3 * (size_t)index] = oa.x; OutputAlbedo[3 * (size_t)index + 1] = oa.y; OutputAlbedo[3 * (size_t)index + 2] = oa.z;
        OutputSampleCountWeight[index] = sampleCountWeightAcc;
        if (OutputDiagnostics) {
            if (p.diagnosticsStride >= 16) memcpy(OutputDiagnostics + (size_t)index * p.diagnosticsStride, &diagnostics, 16);
            else memcpy(OutputDiagnostics + (size_t)index * p.diagnosticsStride, &diagnostics.RayCount, 4);
        }
    }
};

View MakeView(const RtowView& v)
{
    View r;
    r.Origin = f3(v.origin); r.LowerLeftCorner = f3(v.lowerLeftCorner);
    r.Horizontal = f3(v.horizontal); r.Vertical = f3(v.vertical);
    r.Forward = f3(v.forward); r.Up = f3(v.up); r.Right = f3(v.right);
    r.LensRadius = v.lensRadius;
    return r;
}

/* UTIL/MathExtensions.cs:17-21 */
inline float LinearToGamma1(float value)
{
    value = um_max(value, 0.0f);
    return um_max(1.055f * dm_powf(value, 0.416666667f) - 0.055f, 0.0f);
}

} // namespace

/* ===================================================================================================
 * exported C entry points (ctypes)
 * =================================================================================================== */
struct OracleCountersOut {
    uint64_t rays, boundsHit, candidates, nodesVisited, hits;
    uint32_t maxNodeStack, maxCandidates, maxHits, bvhNodeCount, bvhDepth, threads;
};

ORACLE_API void* oracle_scene_create(const RtowSceneDesc* desc)
{
    if (!desc || desc->entityCount <= 0 || !desc->entities || !desc->materials) return nullptr;
    for (int i = 0; i < desc->entityCount; i++)
        if (desc->entities[i].materialIndex < 0 || desc->entities[i].materialIndex >= desc->materialCount) return nullptr;
    OracleScene* s = new OracleScene();
    s->Build(desc);
    if (s->unsupported) { delete s; return nullptr; }
    return s;
}
ORACLE_API void oracle_scene_destroy(void* scene) { delete (OracleScene*)scene; }
/* Environment.SkyCubemap = new Cubemap(...): copies the six faces (NULL drops the cubemap: Sample() then returns 0 like the
 * reference's null data pointer) */
ORACLE_API int oracle_scene_set_cubemap(void* scenePtr, const RtowCubemapDesc* d)
{
    OracleScene* sc = (OracleScene*)scenePtr;
    if (!sc) return 1;
    if (!d || !d->faces) { sc->skyCubemapData.clear(); sc->skyCubemap = Cubemap(); return 0; }
    if (d->faceWidth <= 0 || d->faceHeight <= 0 || d->pixelStride <= 0) return 1;
    const size_t bytes = (size_t)6 * d->faceWidth * d->faceHeight * d->pixelStride;
    sc->skyCubemapData.assign((const uint8_t*)d->faces, (const uint8_t*)d->faces + bytes);
    sc->skyCubemap.Set(*d, sc->skyCubemapData.data());
    return 0;
}
/* BlueNoiseData / SpatioTemporalBlueNoiseData: copies the texture sets (NULL drops them) */
ORACLE_API int oracle_scene_set_blue_noise(void* scenePtr, const RtowBlueNoiseDesc* d)
{
    OracleScene* sc = (OracleScene*)scenePtr;
    if (!sc) return 1;
    sc->blueTexels.clear(); sc->blueRowStride = sc->blueTextureCount = 0;
    if (!d || !d->texels) return 0;
    if (d->rowStride == 0 || d->textureCount == 0) return 1;
    const size_t n = (size_t)d->rowStride * d->rowStride * d->textureCount * 4;
    sc->blueTexels.assign((const uint16_t*)d->texels, (const uint16_t*)d->texels + n);
    sc->blueRowStride = d->rowStride; sc->blueTextureCount = d->textureCount;
    return 0;
}
ORACLE_API int oracle_scene_set_stb_noise(void* scenePtr, const RtowStbNoiseDesc* d)
{
    OracleScene* sc = (OracleScene*)scenePtr;
    if (!sc) return 1;
    sc->stbScalar.clear(); sc->stbVector2.clear(); sc->stbCosine.clear(); sc->stbUnit2.clear(); sc->stbUnit3.clear();
    sc->stbRowStride = sc->stbTextureCount = 0;
    if (!d) return 0;
    if (!d->scalar || !d->vector2 || !d->cosineUnitVector3 || !d->unitVector2 || !d->unitVector3 || d->rowStride == 0 || d->textureCount == 0) return 1;
    const size_t n = (size_t)d->rowStride * d->rowStride * d->textureCount;
    sc->stbScalar.assign((const uint8_t*)d->scalar, (const uint8_t*)d->scalar + n);
    sc->stbVector2.assign((const uint8_t*)d->vector2, (const uint8_t*)d->vector2 + n * 3);
    sc->stbCosine.assign((const uint8_t*)d->cosineUnitVector3, (const uint8_t*)d->cosineUnitVector3 + n * 4);
    sc->stbUnit2.assign((const uint8_t*)d->unitVector2, (const uint8_t*)d->unitVector2 + n * 3);
    sc->stbUnit3.assign((const uint8_t*)d->unitVector3, (const uint8_t*)d->unitVector3 + n * 3);
    sc->stbRowStride = d->rowStride; sc->stbTextureCount = d->textureCount;
    return 0;
}
/* bounds of the reference leaf every entity sits in (out[entity * 6 .. +5] = min.xyz, max.xyz): its own box, or the union in a forced leaf */
ORACLE_API int oracle_kat_leaf_boxes(const RtowSceneDesc* d, float* out)
{
    OracleScene sc;
    sc.Build(d);
    for (const BvhNode& n : sc.nodes)
        if (n.IsLeaf())
            for (int i = 0; i < n.EntityCount; i++) {
                float* o = out + (size_t)sc.bvhEntities[n.EntitiesStart + i].SourceIndex * 6;
                o[0] = n.Bounds.Min.x; o[1] = n.Bounds.Min.y; o[2] = n.Bounds.Min.z; o[3] = n.Bounds.Max.x; o[4] = n.Bounds.Max.y; o[5] = n.Bounds.Max.z;
            }
    return (int)sc.entities.size();
}
/* R2.Next(n) and the first `count` texel indices a PerPixelNoise(seed, (x, y), rowStride) visits */
ORACLE_API void oracle_kat_r2(uint32_t n, float* out) { const float2 r = R2Next(n); out[0] = r.x; out[1] = r.y; }
ORACLE_API void oracle_kat_per_pixel_noise(uint32_t seed, uint32_t x, uint32_t y, uint32_t rowStride, int count, uint32_t* out)
{
    PerPixelNoise p;
    p.Init(seed, x, y, rowStride);
    for (int i = 0; i < count; i++) out[i] = (uint32_t)p.Next();
}
ORACLE_API void oracle_kat_cubemap_sample(const RtowCubemapDesc* d, const float* dir, float* out)
{
    Cubemap c;
    c.Set(*d, (const uint8_t*)d->faces);
    const float3 r = c.Sample(f3(dir[0], dir[1], dir[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
ORACLE_API float oracle_kat_half_to_float(uint16_t h) { return half_to_float(h); }
ORACLE_API int oracle_scene_node_count(void* scene) { return (int)((OracleScene*)scene)->nodes.size(); }
ORACLE_API int oracle_scene_depth(void* scene) { return ((OracleScene*)scene)->maxDepthSeen; }

/* SampleBatchJob.Schedule(W*H, 1): one task per pixel index, dynamic hand-out (UNITY/Raytracer.cs:730). */
static int sample_impl(void* scenePtr, const RtowSampleParams* params,
                       const float* inColor, const float* inNormal, const float* inAlbedo, const float* inScw,
                       float* outColor, float* outNormal, float* outAlbedo, float* outScw,
                       void* diagnostics, int nthreads, OracleCountersOut* countersOut, const int* pixelIndices, int pixelCount);

ORACLE_API int oracle_sample_batch(void* scenePtr, const RtowSampleParams* params,
                                   const float* inColor, const float* inNormal, const float* inAlbedo, const float* inScw,
                                   float* outColor, float* outNormal, float* outAlbedo, float* outScw,
                                   void* diagnostics, int nthreads, OracleCountersOut* countersOut)
{
    return sample_impl(scenePtr, params, inColor, inNormal, inAlbedo, inScw, outColor, outNormal, outAlbedo, outScw, diagnostics, nthreads, countersOut, nullptr, 0);
}

/* SampleBatchJob.Execute for a chosen subset of pixel indices only (pixels are independent): lets tests check a sparse sample
 * of a full-size frame (BASELINE configs 2-5) in seconds.  Buffers are full-frame sized; other pixels are left untouched. */
ORACLE_API int oracle_sample_pixels(void* scenePtr, const RtowSampleParams* params,
                                    const float* inColor, const float* inNormal, const float* inAlbedo, const float* inScw,
                                    float* outColor, float* outNormal, float* outAlbedo, float* outScw,
                                    void* diagnostics, int nthreads, const int* pixelIndices, int pixelCount)
{
    if (!pixelIndices || pixelCount < 0) return 1;
    return sample_impl(scenePtr, params, inColor, inNormal, inAlbedo, inScw, outColor, outNormal, outAlbedo, outScw, diagnostics, nthreads, nullptr, pixelIndices, pixelCount);
}

static int sample_impl(void* scenePtr, const RtowSampleParams* params,
                       const float* inColor, const float* inNormal, const float* inAlbedo, const float* inScw,
                       float* outColor, float* outNormal, float* outAlbedo, float* outScw,
                       void* diagnostics, int nthreads, OracleCountersOut* countersOut, const int* pixelIndices, int pixelCount)
{
    if (!scenePtr || !params) return 1;
    if (params->noiseColor < RTOW_NOISE_WHITE || params->noiseColor > RTOW_NOISE_SPATIOTEMPORAL_BLUE) return 1;
    if (params->rngPolicy != RTOW_RNG_REFERENCE && !((params->rngPolicy == RTOW_RNG_PER_SAMPLE || params->rngPolicy == RTOW_RNG_PER_SAMPLE_XOROSHIRO) && params->noiseColor == RTOW_NOISE_WHITE)) return 1;
    if (params->sliceDivider < 1 || params->traceDepth < 0) return 1;
    const OracleScene* scene = (const OracleScene*)scenePtr;
    Job job;
    job.scene = scene;
    job.p = *params;
    job.view = MakeView(params->view);
    if (!scene->NoiseFor(params->noiseColor, params->noiseTextureIndex, &job.noise)) return 1;   /* no such noise texture */
    job.InputColor = inColor; job.InputNormal = inNormal; job.InputAlbedo = inAlbedo; job.InputSampleCountWeight = inScw;
    job.OutputColor = outColor; job.OutputNormal = outNormal; job.OutputAlbedo = outAlbedo; job.OutputSampleCountWeight = outScw;
    job.OutputDiagnostics = (uint8_t*)diagnostics;

    const int total = pixelIndices ? pixelCount : (int)params->size.x * (int)params->size.y;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    std::atomic<int> next{0};
    std::vector<Counters> perThread(nthreads);
    auto worker = [&](int tid) {
        Job::Scratch scratch;
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= total) break;
            job.Execute(pixelIndices ? pixelIndices[i] : i, scratch);
        }
        perThread[tid] = scratch.counters;
    };
    if (nthreads == 1) worker(0);
    else {
        std::vector<std::thread> threads;
        for (int t = 0; t < nthreads; t++) threads.emplace_back(worker, t);
        for (auto& t : threads) t.join();
    }
    if (countersOut) {
        OracleCountersOut o{};
        for (const Counters& c : perThread) {
            o.rays += c.rays; o.boundsHit += c.boundsHit; o.candidates += c.candidates; o.nodesVisited += c.nodesVisited; o.hits += c.hits;
            o.maxNodeStack = std::max(o.maxNodeStack, c.maxNodeStack);
            o.maxCandidates = std::max(o.maxCandidates, c.maxCandidates);
            o.maxHits = std::max(o.maxHits, c.maxHits);
        }
        o.bvhNodeCount = (uint32_t)scene->nodes.size();
        o.bvhDepth = (uint32_t)scene->maxDepthSeen;
        o.threads = (uint32_t)nthreads;
        *countersOut = o;
    }
    return 0;
}

/* ---- post passes ---- */

/* JOBS/CombineJob.cs:29-71 */
ORACLE_API void oracle_combine(int width, int height, int debugMode, int ldrAlbedo,
                               const float* inColor4, const float* inNormal, const float* inAlbedo,
                               float* outColor, float* outNormal, float* outAlbedo)
{
    const int n = width * height;
    for (int index = 0; index < n; index++) {
        float4 inputColor{inColor4[4 * index], inColor4[4 * index + 1], inColor4[4 * index + 2], inColor4[4 * index + 3]};
        int realSampleCount = (int)inputColor.w;
        float3 finalColor;
        auto anyNan = [](float4 c) { return um_isnan(c.x) || um_isnan(c.y) || um_isnan(c.z) || um_isnan(c.w); };
        if (!debugMode) {
            if (realSampleCount == 0) {
                int tentativeIndex = index;
                while (realSampleCount == 0 && (tentativeIndex -= width) >= 0) {
                    inputColor = float4{inColor4[4 * tentativeIndex], inColor4[4 * tentativeIndex + 1], inColor4[4 * tentativeIndex + 2], inColor4[4 * tentativeIndex + 3]};
                    realSampleCount = (int)inputColor.w;
                }
            }
            if (realSampleCount == 0) finalColor = f3(0);
            else if (anyNan(inputColor)) finalColor = f3(0);
            else finalColor = f3(inputColor.x, inputColor.y, inputColor.z) / (float)realSampleCount;
        } else {
            if (realSampleCount == 0) finalColor = f3(1, 0, 1);
            else if (anyNan(inputColor)) finalColor = f3(0, 1, 1);
            else finalColor = f3(inputColor.x, inputColor.y, inputColor.z) / (float)realSampleCount;
        }
        const float denom = (float)std::max(realSampleCount, 1);
        float3 finalAlbedo = f3(inAlbedo[3 * index], inAlbedo[3 * index + 1], inAlbedo[3 * index + 2]) / denom;
        if (ldrAlbedo) finalAlbedo = um_min(finalAlbedo, f3(1));
        const float3 nrm = um_normalizesafe(f3(inNormal[3 * index], inNormal[3 * index + 1], inNormal[3 * index + 2]) / denom);
        outColor[3 * index] = finalColor.x; outColor[3 * index + 1] = finalColor.y; outColor[3 * index + 2] = finalColor.z;
        outNormal[3 * index] = nrm.x; outNormal[3 * index + 1] = nrm.y; outNormal[3 * index + 2] = nrm.z;
        outAlbedo[3 * index] = finalAlbedo.x; outAlbedo[3 * index + 1] = finalAlbedo.y; outAlbedo[3 * index + 2] = finalAlbedo.z;
    }
}

/* JOBS/FinalizeTexturesJob.cs:23-55 */
ORACLE_API void oracle_finalize(int n, const float* inColor, const float* inNormal, const float* inAlbedo,
                                uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo)
{
    for (int index = 0; index < n; index++) {
        for (int c = 0; c < 3; c++) {
            const float oc = um_saturate(LinearToGamma1(inColor[3 * index + c])) * 255;
            outColor[4 * index + c] = (uint8_t)oc;
            const float on = um_saturate(LinearToGamma1(inNormal[3 * index + c] * 0.5f + 0.5f)) * 255;
            outNormal[4 * index + c] = (uint8_t)on;
            const float oa = um_saturate(LinearToGamma1(inAlbedo[3 * index + c])) * 255;
            outAlbedo[4 * index + c] = (uint8_t)oa;
        }
        outColor[4 * index + 3] = 255; outNormal[4 * index + 3] = 255; outAlbedo[4 * index + 3] = 255;
    }
}

/* JOBS/ReduceMetricsJob.cs:22-45 */
ORACLE_API void oracle_reduce_metrics(int n, const void* diagnostics, int diagnosticsStride, const float* color4, const float* scw,
                                      RtowMetrics* out)
{
    int totalRayCount = 0;
    float minSampleCountWeight = INFINITY, maxSampleCountWeight = -INFINITY;
    float minSamples = INFINITY, maxSamples = -INFINITY;
    int totalSamples = 0;
    int64_t rays64 = 0, samples64 = 0;
    for (int i = 0; i < n; i++) {
        float rayCount;
        memcpy(&rayCount, (const uint8_t*)diagnostics + (size_t)i * diagnosticsStride, 4);
        totalRayCount = (int)((uint32_t)totalRayCount + (uint32_t)(int)rayCount);
        rays64 += (int)rayCount;
        const int sampleCount = (int)color4[4 * (size_t)i + 3];
        totalSamples = (int)((uint32_t)totalSamples + (uint32_t)sampleCount);
        samples64 += sampleCount;
        const float sampleCountWeight = scw[i] / (float)sampleCount;
        minSampleCountWeight = um_min(minSampleCountWeight, sampleCountWeight);
        maxSampleCountWeight = um_max(maxSampleCountWeight, sampleCountWeight);
        minSamples = um_min(minSamples, (float)sampleCount);
        maxSamples = um_max(maxSamples, (float)sampleCount);
    }
    out->totalRayCount = totalRayCount;
    out->totalSamples = totalSamples;
    out->sampleCountWeightExtrema = RtowFloat2{minSampleCountWeight, maxSampleCountWeight};
    out->sampleCountExtrema[0] = (int)minSamples;
    out->sampleCountExtrema[1] = (int)maxSamples;
    out->totalRayCount64 = rays64;
    out->totalSamples64 = samples64;
}

/* ---- known-answer-test hooks ---- */
ORACLE_API void oracle_kat_rng(uint32_t seed, int n, uint32_t* states, float* floats)
{
    UmRandom a; a.Init(seed);
    UmRandom b; b.Init(seed);
    for (int i = 0; i < n; i++) { states[i] = a.NextState(); floats[i] = b.NextFloat(); }
}
ORACLE_API void oracle_kat_xoroshiro(uint32_t seed, int n, uint32_t* outputs, float* floats)
{
    UmRandom a; a.InitXoroshiro(seed);
    UmRandom b; b.InitXoroshiro(seed);
    for (int i = 0; i < n; i++) { outputs[i] = a.NextState(); floats[i] = b.NextFloat(); }
}
ORACLE_API uint32_t oracle_kat_pixel_seed(uint32_t seed, int index) { return (seed * 0x8C4CA03Fu) ^ ((uint32_t)index * 0x7383ED49u); }
ORACLE_API void oracle_kat_sincos(int n, const float* x, float* s, float* c) { for (int i = 0; i < n; i++) dm_sincosf(x[i], &s[i], &c[i]); }
ORACLE_API void oracle_kat_log(int n, const float* x, float* y) { for (int i = 0; i < n; i++) y[i] = dm_logf(x[i]); }
ORACLE_API void oracle_kat_pow(int n, const float* x, const float* e, float* y) { for (int i = 0; i < n; i++) y[i] = dm_powf(x[i], e[i]); }
ORACLE_API float oracle_kat_schlick(float cosine, float ior) { return Material::Schlick(cosine, ior); }
ORACLE_API int oracle_kat_refract(const float* v, const float* n, float niOverNt, float* out)
{
    float3 r;
    const bool ok = Material::Refract(f3(v[0], v[1], v[2]), f3(n[0], n[1], n[2]), niOverNt, &r);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return ok ? 1 : 0;
}
ORACLE_API float oracle_kat_lambda(const float* w, const float* n, float roughness) { return Lambda(f3(w[0], w[1], w[2]), f3(n[0], n[1], n[2]), roughness); }
ORACLE_API float oracle_kat_roughness_to_alpha(float roughness) { return RoughnessToAlpha(roughness); }
ORACLE_API float oracle_kat_linear_to_gamma(float v) { return LinearToGamma1(v); }
ORACLE_API void oracle_kat_basis(const float* n, float* tangent, float* bitangent)
{
    float3 t, b;
    GetOrthonormalBasis(f3(n[0], n[1], n[2]), &t, &b);
    tangent[0] = t.x; tangent[1] = t.y; tangent[2] = t.z;
    bitangent[0] = b.x; bitangent[1] = b.y; bitangent[2] = b.z;
}
ORACLE_API int oracle_kat_aabb_hit(const float* mn, const float* mx, const float* ro, const float* rd)
{
    float3 inv = f3(um_rcp(rd[0]), um_rcp(rd[1]), um_rcp(rd[2]));
    if (um_isnan(inv.x)) inv.x = INFINITY;
    if (um_isnan(inv.y)) inv.y = INFINITY;
    if (um_isnan(inv.z)) inv.z = INFINITY;
    return HitAabb(AABB{f3(mn[0], mn[1], mn[2]), f3(mx[0], mx[1], mx[2])}, f3(ro[0], ro[1], ro[2]), inv) ? 1 : 0;
}
/* unity_sort on (key, id) pairs with the float comparer; ids carry the permutation out */
ORACLE_API void oracle_kat_unity_sort(float* keys, int* ids, int n)
{
    struct KV { float k; int id; };
    std::vector<KV> v(n);
    for (int i = 0; i < n; i++) v[i] = KV{keys[i], ids[i]};
    unity_sort(v.data(), n, [](const KV& x, const KV& y) { return float_compare_to(x.k, y.k); });
    for (int i = 0; i < n; i++) { keys[i] = v[i].k; ids[i] = v[i].id; }
}
/* HeapSort calls since the last query: lets a test prove that an input drove the introsort to its depth limit */
ORACLE_API int oracle_kat_unity_sort_heapsorts(void) { return g_heapSortCalls.exchange(0); }
/* An input of n distinct keys that defeats THIS introsort's median-of-three partitions (M. D. McIlroy, "A Killer Adversary for Quicksort",
 * 1999: the comparator decides the keys while the sort runs), so that sorting it again reaches the 2*floor(log2(n)) depth limit. */
ORACLE_API void oracle_kat_unity_sort_killer(int n, float* keysOut)
{
    std::vector<int> val(n, n), items(n);
    int solid = 0, candidate = 0;
    for (int i = 0; i < n; i++) items[i] = i;
    auto cmp = [&](int x, int y) {
        if (val[x] == n && val[y] == n) { if (x == candidate) val[x] = solid++; else val[y] = solid++; }
        if (val[x] == n) candidate = x; else if (val[y] == n) candidate = y;
        return val[x] - val[y];
    };
    unity_sort(items.data(), n, cmp);
    for (int i = 0; i < n; i++) keysOut[i] = (float)(val[i] == n ? solid++ : val[i]);
    (void)g_heapSortCalls.exchange(0);
}
/* the order in which hits of equal distance start out: source indices of the re-ordered entity array, leaf by leaf, each
 * leaf back to front (candidates are pushed front to back and popped from the end) */
ORACLE_API int oracle_kat_hit_tie_order(const RtowSceneDesc* d, int* out)
{
    OracleScene sc;
    sc.Build(d);
    int k = 0;
    /* leaves in left-to-right order == ascending EntitiesStart */
    std::vector<const BvhNode*> leaves;
    for (const BvhNode& n : sc.nodes) if (n.IsLeaf()) leaves.push_back(&n);
    std::sort(leaves.begin(), leaves.end(), [](const BvhNode* a, const BvhNode* b) { return a->EntitiesStart < b->EntitiesStart; });
    for (const BvhNode* n : leaves)
        for (int i = n->EntityCount - 1; i >= 0; i--) out[k++] = sc.bvhEntities[n->EntitiesStart + i].SourceIndex;
    return k;
}
/* Entity.Hit for one RtowEntity, world-space ray; out = {distance, point[3], normal[3], uv[2]} */
ORACLE_API int oracle_kat_entity_hit(const RtowEntity* ent, const RtowTriangle* triangles, int triangleCount, const float* ro, const float* rd,
                                     float time, float tMin, float tMax, float* out)
{
    Entity e{};
    if (!OracleScene::MakeEntity(*ent, triangles, triangleCount, &e)) return -1;
    HitRecord rec;
    const bool hit = e.Hit(Ray(f3(ro[0], ro[1], ro[2]), f3(rd[0], rd[1], rd[2]), time), tMin, tMax, &rec);
    out[0] = rec.Distance;
    out[1] = rec.Point.x; out[2] = rec.Point.y; out[3] = rec.Point.z;
    out[4] = rec.Normal.x; out[5] = rec.Normal.y; out[6] = rec.Normal.z;
    out[7] = rec.TexCoords.x; out[8] = rec.TexCoords.y;
    return hit ? 1 : 0;
}
/* world-space bounds of one entity (BvhBuildingEntity ctor, UNITY/BvhNodeData.cs:23-81); out = {min[3], max[3]} */
ORACLE_API int oracle_kat_entity_bounds(const RtowEntity* ent, const RtowTriangle* triangles, int triangleCount, float* out)
{
    Entity e{};
    if (!OracleScene::MakeEntity(*ent, triangles, triangleCount, &e)) return -1;
    const AABB b = OracleScene::EntityBounds(e);
    out[0] = b.Min.x; out[1] = b.Min.y; out[2] = b.Min.z; out[3] = b.Max.x; out[4] = b.Max.y; out[5] = b.Max.z;
    return 0;
}
/* One Material.Scatter call. io: rngState (in/out). out = {reflectance[3], origin[3], dir[3], time, randomEvents, draws, isPerfectSpecular, emission[3]} */
ORACLE_API void oracle_kat_scatter(const RtowMaterial* m, const float* ro, const float* rd, float time,
                                   const float* point, const float* normal, float distance, uint32_t* rngState, float* out)
{
    auto tex = [](const RtowTexture& t) { return Texture{t.type, f3(t.mainColor), t.parameter, t.scalarValueChannel}; };
    float parameter = 0;
    if (m->type == RTOW_MATERIAL_DIELECTRIC || m->type == RTOW_MATERIAL_PROBABILISTIC_VOLUME) parameter = m->parameter;
    const Material mat{m->type, tex(m->albedo), tex(m->glossiness), tex(m->emission), tex(m->metallic), parameter};
    RandomSource rng;
    rng.whiteNoise.state = *rngState;
    rng.RandomEvents = 0;
    rng.draws = 0;
    HitRecord rec{};
    rec.Distance = distance;
    rec.Point = f3(point[0], point[1], point[2]);
    rec.Normal = f3(normal[0], normal[1], normal[2]);
    float3 reflectance;
    Ray scattered;
    mat.Scatter(Ray(f3(ro[0], ro[1], ro[2]), f3(rd[0], rd[1], rd[2]), time), rec, rng, &reflectance, &scattered);
    const float3 em = mat.Emit(rec.TexCoords);
    *rngState = rng.whiteNoise.state;
    out[0] = reflectance.x; out[1] = reflectance.y; out[2] = reflectance.z;
    out[3] = scattered.Origin.x; out[4] = scattered.Origin.y; out[5] = scattered.Origin.z;
    out[6] = scattered.Direction.x; out[7] = scattered.Direction.y; out[8] = scattered.Direction.z;
    out[9] = scattered.Time;
    out[10] = rng.RandomEvents;
    out[11] = (float)rng.draws;
    out[12] = mat.IsPerfectSpecular() ? 1.0f : 0.0f;
    out[13] = em.x; out[14] = em.y; out[15] = em.z;
}
/* View.GetRay for one normalized coordinate. out = {origin[3], dir[3], time, draws} */
ORACLE_API void oracle_kat_get_ray(const RtowView* v, float u, float w, uint32_t* rngState, float* out)
{
    const View view = MakeView(*v);
    RandomSource rng;
    rng.whiteNoise.state = *rngState;
    rng.RandomEvents = 0;
    rng.draws = 0;
    const Ray r = view.GetRay(float2{u, w}, rng);
    *rngState = rng.whiteNoise.state;
    out[0] = r.Origin.x; out[1] = r.Origin.y; out[2] = r.Origin.z;
    out[3] = r.Direction.x; out[4] = r.Direction.y; out[5] = r.Direction.z;
    out[6] = r.Time;
    out[7] = (float)rng.draws;
}
/* Nearest hit along a ray through the reference-shaped pipeline (FindHitCandidates + FindHits, element 0);
 * used for the auto-focus probe (UNITY/Raytracer.cs:608-609) and first-hit KATs.
 * out = {distance, point[3], normal[3], entityIndex}; returns the number of hits found. */
ORACLE_API int oracle_kat_nearest_hit(void* scenePtr, const float* ro, const float* rd, float time, float* out)
{
    const OracleScene* scene = (const OracleScene*)scenePtr;
    Job job;
    job.scene = scene;
    Job::Scratch s;
    Diagnostics d{0, 0, 0, 0};
    const Ray ray(f3(ro[0], ro[1], ro[2]), f3(rd[0], rd[1], rd[2]), time);
    job.FindHitCandidates(ray, s, d);
    job.FindHits(ray, s);
    if (s.hitRecordBuffer.empty()) return 0;
    const HitRecord& r = s.hitRecordBuffer[0];
    out[0] = r.Distance;
    out[1] = r.Point.x; out[2] = r.Point.y; out[3] = r.Point.z;
    out[4] = r.Normal.x; out[5] = r.Normal.y; out[6] = r.Normal.z;
    out[7] = (float)r.EntityPtr->SourceIndex;
    return (int)s.hitRecordBuffer.size();
}

/* HitTests.Hit(this BvhNode n, Ray r, float tMin, float tMax, out HitRecord rec) (RT/HitTests.cs:152-196): the recursion behind Raytracer.HitWorld
 * (UNITY/Raytracer.cs:1353), the auto-focus probe of ScheduleSample (:608-609).  Not the job's FindHitCandidates / FindHits pair: the reciprocal
 * direction is taken as it comes (no NaN -> INFINITY step), a leaf keeps its FIRST entity on equal distances (:167 `thisRec.Distance < rec.Distance`),
 * an inner node its RIGHT child's hit (:185 `leftRecord.Distance < rightRecord.Distance ? leftRecord : rightRecord`). */
static bool HitBvhNode(const OracleScene* scene, int nodeIndex, const Ray& r, float tMin, float tMax, HitRecord* rec)
{
    const BvhNode& n = scene->nodes[nodeIndex];
    memset(rec, 0, sizeof(*rec));                                                                  /* rec = default */
    const float3 rayInvDirection = f3(um_rcp(r.Direction.x), um_rcp(r.Direction.y), um_rcp(r.Direction.z));
    if (!HitAabb(n.Bounds, r.Origin, rayInvDirection)) return false;
    if (n.IsLeaf()) {
        bool anyHit = false;
        for (int i = 0; i < n.EntityCount; i++) {
            HitRecord thisRec;
            const bool thisHit = scene->bvhEntities[n.EntitiesStart + i].Hit(r, tMin, tMax, &thisRec);
            if (thisHit && (!anyHit || thisRec.Distance < rec->Distance)) {
                anyHit = true;
                *rec = thisRec;
                rec->EntityPtr = &scene->bvhEntities[n.EntitiesStart + i];
            }
        }
        return anyHit;
    }
    HitRecord leftRecord, rightRecord;
    const bool hitLeft = HitBvhNode(scene, n.Left, r, tMin, tMax, &leftRecord);
    const bool hitRight = HitBvhNode(scene, n.Right, r, tMin, tMax, &rightRecord);
    if (!hitLeft && !hitRight) return false;
    if (hitLeft && hitRight) { *rec = leftRecord.Distance < rightRecord.Distance ? leftRecord : rightRecord; return true; }
    if (hitLeft) { *rec = leftRecord; return true; }
    *rec = rightRecord;
    return true;
}
/* Raytracer.HitWorld: BvhRoot->Hit(r, 0, float.PositiveInfinity, out hitRec).  out = {distance, point[3], normal[3], entityIndex}; returns 1 on a hit. */
ORACLE_API int oracle_hit_world(void* scenePtr, const float* ro, const float* rd, float time, float* out)
{
    const OracleScene* scene = (const OracleScene*)scenePtr;
    const Ray ray(f3(ro[0], ro[1], ro[2]), f3(rd[0], rd[1], rd[2]), time);
    HitRecord r;
    if (!HitBvhNode(scene, 0, ray, 0.0f, INFINITY, &r)) return 0;
    out[0] = r.Distance;
    out[1] = r.Point.x; out[2] = r.Point.y; out[3] = r.Point.z;
    out[4] = r.Normal.x; out[5] = r.Normal.y; out[6] = r.Normal.z;
    out[7] = (float)r.EntityPtr->SourceIndex;
    return 1;
}

/* sizeof() of every boundary struct as the C++ compiler sees include/rtow.h; tests compare them with the ctypes mirror. */
ORACLE_API void oracle_abi_sizes(int* out)
{
    out[0] = (int)sizeof(RtowTexture); out[1] = (int)sizeof(RtowMaterial); out[2] = (int)sizeof(RtowEntity);
    out[3] = (int)sizeof(RtowSceneDesc); out[4] = (int)sizeof(RtowSceneInfo); out[5] = (int)sizeof(RtowView);
    out[6] = (int)sizeof(RtowEnvironment); out[7] = (int)sizeof(RtowSampleParams); out[8] = (int)sizeof(RtowAccumBuffers);
    out[9] = (int)sizeof(RtowContextOptions); out[10] = (int)sizeof(RtowMetrics); out[11] = (int)sizeof(RtowCombineParams);
    out[12] = (int)sizeof(RtowTriangle); out[13] = (int)sizeof(RtowCubemapDesc);
    out[14] = (int)sizeof(RtowBlueNoiseDesc); out[15] = (int)sizeof(RtowStbNoiseDesc); out[16] = (int)sizeof(RtowImage);
    out[17] = (int)sizeof(RtowCommId);
}
