// valu_calib.hip - issue-rate calibration of the gfx950 vector ALU, LDS and the scheduler idioms the sample kernel is made of.
//
// Purpose: two rocprofv3 figures disagreed about how busy the VALU is under sample_batch_kernel -
// SQ_INSTS_VALU x 2 cycles (0.47) against SQ_ACTIVE_INST_VALU (0.96).  This program runs pure instruction streams with a known
// instruction count per wave, times them with the shader clock, and is profiled with the same counters, so that
//   * cycles per wave64 instruction are known per opcode class (full rate, packed, transcendental, integer multiply, LDS), and
//   * the counters' units are pinned (what one v_fma_f32 adds to SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU).
// Stand-alone test infrastructure: not linked into librtow_hip.so.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_calib valu_calib.hip && ./valu_calib [wavesPerSimd=4] > calib.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
// eight independent destinations per group, four groups per loop trip = 32 instructions of the class per trip
#define DEF_KERNEL_F32(NAME, ASM)                                                                                       \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long* out, int iters, float seed)                      \
    {                                                                                                                   \
        float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 2, a3 = a0 * 0.7f, a4 = a0 + 4, a5 = a0 * 0.3f, a6 = a0 + 6, a7 = a0 * 0.9f; \
        float b = seed * 1.0001f + 1.0f, c = seed * 0.5f + 0.25f;                                                        \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                                                        \
        for (int i = 0; i < iters; i++) {                                                                               \
            asm volatile(REP4(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                                               \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                                                        \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[1 << 20] = 1;                                      \
        if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }               \
    }

#define I8(op, tail) \
    op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"

DEF_KERNEL_F32(k_fma, I8("v_fma_f32", ", %8, %9"))
DEF_KERNEL_F32(k_mul, I8("v_mul_f32", ", %8"))
DEF_KERNEL_F32(k_add, I8("v_add_f32", ", %8"))
DEF_KERNEL_F32(k_min, I8("v_min_f32", ", %8"))
DEF_KERNEL_F32(k_sub, I8("v_sub_f32", ", %8"))
DEF_KERNEL_F32(k_fmac, I8("v_fmac_f32", ", %8"))
DEF_KERNEL_F32(k_and, I8("v_and_b32", ", %8"))
DEF_KERNEL_F32(k_or, I8("v_or_b32", ", %8"))
DEF_KERNEL_F32(k_add_u32, I8("v_add_u32", ", %8"))
DEF_KERNEL_F32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %8\nv_lshl_add_u32 %1, %1, 3, %8\nv_lshl_add_u32 %2, %2, 3, %8\nv_lshl_add_u32 %3, %3, 3, %8\nv_lshl_add_u32 %4, %4, 3, %8\nv_lshl_add_u32 %5, %5, 3, %8\nv_lshl_add_u32 %6, %6, 3, %8\nv_lshl_add_u32 %7, %7, 3, %8\n")
DEF_KERNEL_F32(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %8\nv_mov_b32 %3, %9\nv_mov_b32 %4, %8\nv_mov_b32 %5, %9\nv_mov_b32 %6, %8\nv_mov_b32 %7, %9\n")
DEF_KERNEL_F32(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n")
DEF_KERNEL_F32(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\nv_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n")
DEF_KERNEL_F32(k_med3, I8("v_med3_f32", ", %8, %9"))
DEF_KERNEL_F32(k_mul_then_min, "v_mul_f32 %0, %0, %8\nv_min_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_min_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_min_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_min_f32 %7, %7, %8\n")
DEF_KERNEL_F32(k_max3, I8("v_max3_f32", ", %8, %9"))
DEF_KERNEL_F32(k_rcp, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n")
DEF_KERNEL_F32(k_sqrt, "v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\nv_sqrt_f32 %4, %4\nv_sqrt_f32 %5, %5\nv_sqrt_f32 %6, %6\nv_sqrt_f32 %7, %7\n")
DEF_KERNEL_F32(k_rsq, "v_rsq_f32 %0, %0\nv_rsq_f32 %1, %1\nv_rsq_f32 %2, %2\nv_rsq_f32 %3, %3\nv_rsq_f32 %4, %4\nv_rsq_f32 %5, %5\nv_rsq_f32 %6, %6\nv_rsq_f32 %7, %7\n")
DEF_KERNEL_F32(k_xor, I8("v_xor_b32", ", %8"))
DEF_KERNEL_F32(k_lshl, "v_lshlrev_b32 %0, 13, %0\nv_lshlrev_b32 %1, 13, %1\nv_lshlrev_b32 %2, 13, %2\nv_lshlrev_b32 %3, 13, %3\nv_lshlrev_b32 %4, 13, %4\nv_lshlrev_b32 %5, 13, %5\nv_lshlrev_b32 %6, 13, %6\nv_lshlrev_b32 %7, 13, %7\n")
DEF_KERNEL_F32(k_lshl_xor, "v_lshl_or_b32 %0, %0, 13, %0\nv_lshl_or_b32 %1, %1, 13, %1\nv_lshl_or_b32 %2, %2, 13, %2\nv_lshl_or_b32 %3, %3, 13, %3\nv_lshl_or_b32 %4, %4, 13, %4\nv_lshl_or_b32 %5, %5, 13, %5\nv_lshl_or_b32 %6, %6, 13, %6\nv_lshl_or_b32 %7, %7, 13, %7\n")
DEF_KERNEL_F32(k_mul_lo_u32, I8("v_mul_lo_u32", ", %8"))
DEF_KERNEL_F32(k_mad_u32_u24, I8("v_mad_u32_u24", ", %8, %9"))
DEF_KERNEL_F32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\nv_cvt_f32_u32 %4, %4\nv_cvt_f32_u32 %5, %5\nv_cvt_f32_u32 %6, %6\nv_cvt_f32_u32 %7, %7\n")
DEF_KERNEL_F32(k_cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %9, vcc\n")
DEF_KERNEL_F32(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %8, %0\nv_div_scale_f32 %1, vcc, %1, %8, %1\nv_div_scale_f32 %2, vcc, %2, %8, %2\nv_div_scale_f32 %3, vcc, %3, %8, %3\nv_div_scale_f32 %4, vcc, %4, %8, %4\nv_div_scale_f32 %5, vcc, %5, %8, %5\nv_div_scale_f32 %6, vcc, %6, %8, %6\nv_div_scale_f32 %7, vcc, %7, %8, %7\n")
DEF_KERNEL_F32(k_div_fixup, I8("v_div_fixup_f32", ", %8, %9"))
DEF_KERNEL_F32(k_div_fmas, I8("v_div_fmas_f32", ", %8, %9"))
// scheduler idiom: ballot (v_cmp into an SGPR pair) + popcount, as run ~8 times per trip of the stage loop
__global__ void __launch_bounds__(1024) k_ballot_popc(unsigned long long* out, int iters, float seed)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 + 2, a3 = a0 * 7u;
    const unsigned b = (unsigned)seed;
    unsigned acc = 0;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc += (unsigned)__popcll(__ballot(a0 == b + (unsigned)k)) + (unsigned)__popcll(__ballot(a1 == b + (unsigned)i)) +
                   (unsigned)__popcll(__ballot(a2 == b + (unsigned)(k + i))) + (unsigned)__popcll(__ballot(a3 == b + (unsigned)(k ^ i)));
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (acc == 0x12345678u) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }
}
// dependent chain (latency, one wave's view): every instruction consumes the previous result
DEF_KERNEL_F32(k_fma_dependent, "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n")
DEF_KERNEL_F32(k_rcp_dependent, "v_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\n")

// half the lanes masked off: is a VALU instruction cheaper with fewer active lanes?  (it is not: EXEC only gates the write)
__global__ void __launch_bounds__(1024) k_fma_half_exec(unsigned long long* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 2, a3 = a0 * 0.7f, a4 = a0 + 4, a5 = a0 * 0.3f, a6 = a0 + 6, a7 = a0 * 0.9f;
    float b = seed * 1.0001f + 1.0f, c = seed * 0.5f + 0.25f;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    if (threadIdx.x & 1) {
        for (int i = 0; i < iters; i++) {
            asm volatile(REP4(I8("v_fma_f32", ", %8, %9")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }
}

// packed fp32 (two floats per lane per instruction): the box walk's slab arithmetic
#define DEF_KERNEL_PK(NAME, ASM)                                                                                        \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long* out, int iters, float seed)                      \
    {                                                                                                                   \
        const float s = seed + threadIdx.x;                                                                             \
        f2 a0 = {s, s + 1}, a1 = {s * 1.5f, s}, a2 = {s + 2, s}, a3 = {s * 0.7f, s}, a4 = {s + 4, s}, a5 = {s * 0.3f, s}, a6 = {s + 6, s}, a7 = {s * 0.9f, s}; \
        f2 b = {seed * 1.0001f + 1.0f, seed + 3.0f}, c = {seed * 0.5f + 0.25f, seed};                                    \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                                                        \
        for (int i = 0; i < iters; i++) {                                                                               \
            asm volatile(REP4(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                                               \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                                                        \
        if (a0.x + a1.x + a2.y + a3.x + a4.y + a5.x + a6.x + a7.y == 12345.678f) out[1 << 20] = 1;                      \
        if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }               \
    }
DEF_KERNEL_PK(k_pk_mul, I8("v_pk_mul_f32", ", %8"))
DEF_KERNEL_PK(k_pk_add, I8("v_pk_add_f32", ", %8"))
DEF_KERNEL_PK(k_pk_fma, I8("v_pk_fma_f32", ", %8, %9"))

// LDS: the walk's node fetch (3 x ds_read_b128 + ds_read_b64 per visit), the [level][lane] 16-bit stack / candidate slots
template <int KIND>
__global__ void __launch_bounds__(1024) k_lds(unsigned long long* out, int iters, float seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (unsigned i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    unsigned idx = (threadIdx.x * 97u + (unsigned)seed) & 511u;   // divergent node index, like lanes at different tree nodes
    float acc = 0;
    unsigned short* st = reinterpret_cast<unsigned short*>(smem + 32768) + (threadIdx.x & ~63u) + ((threadIdx.x & 31u) << 1) + ((threadIdx.x >> 5) & 1u);
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) {          // one 64-byte node: 3 x b128 + 1 x b64 (4 LDS instructions)
                const float4* p = reinterpret_cast<const float4*>(smem + idx * 64u);
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                const int2 c = *reinterpret_cast<const int2*>(smem + idx * 64u + 48);
                acc += q0.x + q1.y + q2.z;
                idx = ((unsigned)c.x ^ (unsigned)c.y ^ __float_as_uint(q0.w)) & 511u;
            } else if (KIND == 1) {   // 16-bit stack slot read + write at a lane-dependent level
                const unsigned lvl = idx & 15u;
                const unsigned v = st[lvl * 1024u];
                st[((lvl + 1u) & 15u) * 1024u] = (unsigned short)(v + 1u);
                idx = (idx + v) & 511u;
            } else {                  // 16-byte sphere record
                const float4 s4 = *reinterpret_cast<const float4*>(smem + (idx & 511u) * 16u);
                acc += s4.x;
                idx = (__float_as_uint(s4.w) >> 3) & 511u;
            }
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (acc == 12345.678f) out[1 << 20] = idx;
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }
}

// what the compiler makes of the path's IEEE operations (hipcc default: correctly rounded divide / sqrt), per operation
template <int KIND>
__global__ void __launch_bounds__(1024) k_ieee(unsigned long long* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x + 1.0f, a1 = a0 * 1.5f, a2 = a0 + 2, a3 = a0 * 0.7f;
    const float b = seed * 1.0001f + 1.5f;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) { a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; a3 = a3 / b; }
            else if (KIND == 1) { a0 = __builtin_sqrtf(a0) + b; a1 = __builtin_sqrtf(a1) + b; a2 = __builtin_sqrtf(a2) + b; a3 = __builtin_sqrtf(a3) + b; }
            else { a0 = 1.0f / __builtin_sqrtf(a0 + b); a1 = 1.0f / __builtin_sqrtf(a1 + b); a2 = 1.0f / __builtin_sqrtf(a2 + b); a3 = 1.0f / __builtin_sqrtf(a3 + b); }
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (a0 + a1 + a2 + a3 == 12345.678f) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; out[65536 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0; }
}

struct Case { const char* name; void (*fn)(unsigned long long*, int, float); int perTrip; size_t lds; const char* what; };

int main(int argc, char** argv)
{
    int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 4;
    if (wavesPerSimd < 1 || wavesPerSimd > 4) wavesPerSimd = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    unsigned long long* d = nullptr;
    hipMalloc(&d, ((1 << 20) + 16) * sizeof(unsigned long long));
    const Case cases[] = {
        {"v_fma_f32", k_fma, 32, 0, "full-rate fp32"},
        {"v_mul_f32", k_mul, 32, 0, "full-rate fp32"},
        {"v_add_f32", k_add, 32, 0, "full-rate fp32"},
        {"v_min_f32", k_min, 32, 0, "slab test"},
        {"v_sub_f32", k_sub, 32, 0, "full-rate fp32?"},
        {"v_fmac_f32", k_fmac, 32, 0, "full-rate fp32?"},
        {"v_med3_f32", k_med3, 32, 0, "slab test alternative"},
        {"v_mul+v_min alternating", k_mul_then_min, 32, 0, "does a half-rate op hold the pipe? (per instruction)"},
        {"v_and_b32", k_and, 32, 0, "bit ops"},
        {"v_or_b32", k_or, 32, 0, "bit ops"},
        {"v_add_u32", k_add_u32, 32, 0, "integer add"},
        {"v_lshl_add_u32", k_lshl_add, 32, 0, "LDS addressing"},
        {"v_mov_b32", k_mov, 32, 0, "moves"},
        {"v_cndmask_b32 (vcc)", k_cndmask, 32, 0, "select only"},
        {"v_cmp_lt_f32 (vcc)", k_cmp, 32, 0, "compare only"},
        {"v_max3_f32", k_max3, 32, 0, "slab test"},
        {"v_pk_mul_f32", k_pk_mul, 32, 0, "packed fp32 (2 floats / lane)"},
        {"v_pk_add_f32", k_pk_add, 32, 0, "packed fp32"},
        {"v_pk_fma_f32", k_pk_fma, 32, 0, "packed fp32"},
        {"v_rcp_f32", k_rcp, 32, 0, "transcendental"},
        {"v_sqrt_f32", k_sqrt, 32, 0, "transcendental"},
        {"v_rsq_f32", k_rsq, 32, 0, "transcendental"},
        {"v_xor_b32", k_xor, 32, 0, "xorshift32"},
        {"v_lshlrev_b32", k_lshl, 32, 0, "xorshift32"},
        {"v_lshl_or_b32", k_lshl_xor, 32, 0, "shift+or fused (v_lshl_or_b32)"},
        {"v_mul_lo_u32", k_mul_lo_u32, 32, 0, "integer multiply"},
        {"v_mad_u32_u24", k_mad_u32_u24, 32, 0, "24-bit multiply-add (LDS addressing)"},
        {"v_cvt_f32_u32", k_cvt_f32_u32, 32, 0, "conversion"},
        {"v_cmp+v_cndmask", k_cmp_cndmask, 32, 0, "select (2 instructions counted as 2)"},
        {"v_div_scale_f32", k_div_scale, 32, 0, "IEEE division expansion"},
        {"v_div_fmas_f32", k_div_fmas, 32, 0, "IEEE division expansion"},
        {"v_div_fixup_f32", k_div_fixup, 32, 0, "IEEE division expansion"},
        {"ballot+popcount", k_ballot_popc, 32, 0, "__popcll(__ballot(x == y)): v_cmp to an SGPR pair + s_bcnt1 (+ add), per ballot"},
        {"v_fma_f32 dependent", k_fma_dependent, 32, 0, "latency chain"},
        {"v_rcp_f32 dependent", k_rcp_dependent, 32, 0, "latency chain"},
        {"v_fma_f32 half EXEC", k_fma_half_exec, 32, 0, "32 of 64 lanes active"},
        {"lds node fetch (3xb128+b64)", k_lds<0>, 8, 65536, "per 64-byte node, divergent addresses"},
        {"lds u16 stack read+write", k_lds<1>, 8, 65536, "per read+write pair"},
        {"lds b128 sphere record", k_lds<2>, 8, 65536, "per record"},
        {"IEEE a/b (compiler expansion)", k_ieee<0>, 32, 0, "per division, 4 independent chains"},
        {"IEEE sqrtf (compiler expansion)", k_ieee<1>, 32, 0, "per sqrt (+1 add)"},
        {"IEEE 1/sqrtf(x+b)", k_ieee<2>, 32, 0, "per normalize-style rsqrt (+1 add)"},
    };
    const int iters = 4096;
    const int block = 256 * wavesPerSimd;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_simd\": %d, \"cases\": [\n", prop.gcnArchName, cus, prop.clockRate / 1000, wavesPerSimd);
    const int nCases = (int)(sizeof(cases) / sizeof(cases[0]));
    for (int ci = 0; ci < nCases; ci++) {
        const Case& c = cases[ci];
        if (c.lds) hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        const int waves = cus * (block / 64);
        std::vector<unsigned long long> h(waves), hw(waves);
        double best = 1e30, bestWall = 1e30;
        float bestMs = 0;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(block), c.lds, 0, d, iters, 1.0f + rep);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            hipMemcpy(hw.data(), d + 65536, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum = 0, sumW = 0;
            for (unsigned long long v : h) sum += (double)v;
            for (unsigned long long v : hw) sumW += (double)v;
            if (sumW / waves < bestWall) { best = sum / waves; bestWall = sumW / waves; bestMs = ms; }
        }
        const double ops = (double)iters * c.perTrip;                 // per wave
        // per wave and per instruction: s_memtime ticks (clock64) and nanoseconds (wall_clock64 = s_memrealtime, 100 MHz) inside the timed loop;
        // per SIMD: the same divided by the waves that share the SIMD (they interleave).  cycles_at_nominal converts time with
        // hipDeviceProp.clockRate; the part may clock lower under load, so ratios between rows are the robust figures.
        const double nsWave = bestWall * 10.0;
        printf("  {\"name\": \"%s\", \"what\": \"%s\", \"ops_per_wave\": %.0f, \"memtime_ticks_per_op_per_simd\": %.4f, \"ns_per_op_per_simd\": %.4f, \"kernel_ms\": %.4f, "
               "\"cycles_per_op_per_simd_at_nominal_clock\": %.3f}%s\n",
               c.name, c.what, ops, best / ops / wavesPerSimd, nsWave / ops / wavesPerSimd, bestMs, nsWave * 1e-9 * (double)prop.clockRate * 1e3 / (ops * wavesPerSimd), ci + 1 < nCases ? "," : "");
        fflush(stdout);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("]}\n");
    hipFree(d);
    return 0;
}
