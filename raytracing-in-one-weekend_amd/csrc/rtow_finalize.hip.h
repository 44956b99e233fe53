// rtow_finalize.hip.h - the float -> byte conversion of FinalizeTexturesJob (JOBS/FinalizeTexturesJob.cs:23-55), two ways.
//
// to_byte_exact is the specification: LinearToGamma (UTIL/MathExtensions.cs:17-21: max(1.055 * pow(v, 0.416666667) - 0.055, 0)) on
// max(v, 0), saturate, * 255, (byte) - with DESIGN.md's deterministic pow (det_log + det_exp2, ~70 instructions).  Nine of them per pixel made
// the finalize kernel compute bound at 2.9 TB/s of a 5.4 TB/s stream (profiles/r03z_post_passes.json).
//
// to_byte_table gives the same byte for every one of the 2^32 float operands (tests/native/finalize_parity.hip sweeps them all on the
// device) from ~20 instructions: the conversion is a step function of v with 255 steps, so a byte is described by the 255 thresholds
// T[k] = the float from which the byte stays >= k (plus the one stretch of floats where the exact form wobbles).  A hardware v_log_f32 / v_exp_f32 estimate lands within one step of the answer, and two
// comparisons against the table move it onto it.  The table is built on the device, from to_byte_exact itself, when a context is created.
#pragma once
#include <hip/hip_runtime.h>

#include "rtow_detmath.hip.h"

namespace rtow {

constexpr int kByteZones = 1;                     // mixed zones the table can describe (the deterministic pow has ONE, see below; more: everything takes the exact form)
constexpr int kByteThresholdFloats = 257 + 2 * kByteZones;
// T[0] = -inf, T[1..255] = the steps, T[256] = NaN (no float reaches it), T[257 + 2z], T[258 + 2z] = mixed zone z as [first, end) (NaN, NaN = none)

__device__ __forceinline__ float fin_max(float x, float y) { return (y != y || x > y) ? x : y; }      // math.max: a NaN second operand is skipped
__device__ __forceinline__ float fin_min(float x, float y) { return (y != y || x < y) ? x : y; }

__device__ __forceinline__ unsigned to_byte_exact(float v)
{
    v = fin_max(v, 0.0f);
    const float g = fin_max(1.055f * det_pow(v, 0.416666667f) - 0.055f, 0.0f);
    return (unsigned)(fin_max(0.0f, fin_min(1.0f, g)) * 255);
}

// One thread per step k: bisection over the bit patterns of [+0, +inf] (ordered like the floats they encode) finds A place where the byte
// reaches k.  The polynomial pow is not monotone to the last ulp - at 0x3eef815b the byte goes 123, 124, 123, 124 over four neighbouring
// floats - so each thread then walks kScan floats either side: T[k] = the float from which the byte STAYS >= k, and if some float below it
// already reached k the stretch between the two is a mixed zone, which to_byte_table hands to the exact form.  More zones than the table
// holds: one zone covering everything (all exact, still correct).  Whether a window of kScan floats is enough is not argued, it is
// enumerated: tests/native/finalize_parity.hip compares the two forms on all 2^32 operands.
__global__ void __launch_bounds__(256) build_byte_thresholds_kernel(float* __restrict__ T)
{
    constexpr unsigned kScan = 256u;
    __shared__ unsigned zones;
    const unsigned k = threadIdx.x;
    if (k == 0) {
        zones = 0u;
        T[0] = -__builtin_inff(); T[256] = __builtin_nanf("");
        for (int z = 0; z < 2 * kByteZones; z++) T[257 + z] = __builtin_nanf("");
    }
    __syncthreads();
    if (k != 0) {
        unsigned lo = 0u, hi = 0x7f800000u;           // byte(+0) = 0 < k <= 255 = byte(+inf)
        while (hi - lo > 1u) {
            const unsigned mid = lo + (hi - lo) / 2u;
            if (to_byte_exact(__uint_as_float(mid)) >= k) hi = mid; else lo = mid;
        }
        const unsigned from = hi > kScan ? hi - kScan : 0u, to = hi < 0x7f800000u - kScan ? hi + kScan : 0x7f800000u;
        unsigned first = hi, stays = hi;
        for (unsigned i = from; i <= to; i++) {
            const bool reached = to_byte_exact(__uint_as_float(i)) >= k;
            if (reached && i < first) first = i;
            if (!reached && i >= stays) stays = i + 1u;
        }
        T[k] = __uint_as_float(stays);
        if (first < stays) {
            const unsigned z = atomicAdd(&zones, 1u);
            if (z < (unsigned)kByteZones) { T[257 + 2 * z] = __uint_as_float(first); T[258 + 2 * z] = __uint_as_float(stays); }
        }
    }
    __syncthreads();
    // more zones than the table holds: one zone covering every operand (all exact, still correct; inf itself is 255 on both paths)
    if (k == 0 && zones > (unsigned)kByteZones) { T[257] = 0.0f; T[258] = __builtin_inff(); }
}

// T: the table above (LDS).  NaN and negative operands give 0, like the exact form (math.max(NaN, 0) is 0).
// The conversion is written for N operands at once (a pixel's nine channels): N independent chains of estimate -> ONE paired LDS read of
// T[k], T[k + 1] -> fix-up, with the (one) mixed zone handled after all of them by a wave-rare branch into an out-of-line copy of the exact
// form.  Nine inlined copies of "estimate, read, wait, read, wait, or else the exact form" serialised 18 LDS round trips per pixel behind
// branches and held the kernel at 4.4 TB/s (profiles/r03zb_post_passes.json).
// A mixed zone [first, end) of non-negative floats as a range of bit patterns (ordered like the floats): x is inside iff bits(x) - first < end - first
// as unsigned numbers - a subtraction and one comparison per operand.  No zone (NaN, NaN in the table): length 0.
struct ByteZones { unsigned first[kByteZones], length[kByteZones]; };
__device__ __forceinline__ ByteZones load_byte_zones(const float* T)
{
    ByteZones Z;
    for (int z = 0; z < kByteZones; z++) {
        const float lo = T[257 + 2 * z], hi = T[258 + 2 * z];
        const bool none = lo != lo || hi != hi;
        Z.first[z] = none ? 0u : __float_as_uint(lo);
        Z.length[z] = none ? 0u : __float_as_uint(hi) - __float_as_uint(lo);
    }
    return Z;
}
__device__ __noinline__ unsigned to_byte_exact_outlined(float v) { return to_byte_exact(v); }

template <int N>
__device__ __forceinline__ void to_bytes_table(const float (&v)[N], unsigned (&out)[N], const float* T, const ByteZones& Z)
{
    bool mixed[N];
    bool any = false;
#pragma unroll
    for (int c = 0; c < N; c++) {
        // max(v, 0) with NaN -> 0, on the bit pattern: every float with the sign bit set (negative numbers, -0, negative NaNs) is a negative
        // integer.  A positive NaN stays: it fails every comparison below, its estimate clamps to 0, and it lies in no zone - byte 0 all the same.
        const int xb = __float_as_int(v[c]);
        const float x = __int_as_float(xb > 0 ? xb : 0);
        bool m = false;
        for (int z = 0; z < kByteZones; z++) m = m | (__float_as_uint(x) - Z.first[z] < Z.length[z]);      // |: no branches between the nine chains
        mixed[c] = m;
        any = any | m;
        // estimate: exp2(log2(x) / 2.4) on the transcendental unit (x = 0: log2 = -inf, exp2 = 0); any error below one step is repaired next
        const float g = 1.055f * __builtin_amdgcn_exp2f(0.416666667f * __builtin_amdgcn_logf(x)) - 0.055f;
        const float s = __builtin_amdgcn_fmed3f(g, 0.0f, 1.0f);   // clamp; a NaN estimate (NaN operand) gives 0 or NaN here and k = 0 either way
        const int k = (int)(s * 255.0f);
        const float lo = T[k], hi = T[k + 1];                     // one ds_read2_b32; T[256] = NaN never compares
        // at most one of the two holds (T[k] < T[k + 1]); written as a sum so that both thresholds are read up front, not one behind a branch
        out[c] = (unsigned)(k + (x >= hi ? 1 : 0) - (x < lo ? 1 : 0));
    }
    if (any) {
#pragma unroll
        for (int c = 0; c < N; c++) if (mixed[c]) out[c] = to_byte_exact_outlined(v[c]);
    }
}

__device__ __forceinline__ unsigned to_byte_table(float v, const float* T)
{
    const float in[1] = {v};
    unsigned out[1];
    to_bytes_table<1>(in, out, T, load_byte_zones(T));
    return out[0];
}

} // namespace rtow
