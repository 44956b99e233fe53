// xcd_affinity_probe.hip - litmus for handing data between workgroups on the SAME XCD inside one running kernel (gfx950: eight XCDs, one L2 each),
// and a census of where a persistent launch's workgroups land.  Question behind it (DESIGN.md 4.1 "Roofline"): a chained launch makes every
// accumulator access write-through (sc1) because batch b + 1 of a pixel chunk may run on another XCD; if a chunk's batches all stayed on one
// XCD, would plain (write-back) stores plus L1-bypassing (sc0) loads be enough?
//   census   XCC_ID (s_getreg_b32 hwreg(HW_REG_XCC_ID)) of every workgroup of a 256 x 1024-lane launch with 140 KB of LDS each
//   litmus   writer fills 1 KiB with the round number by PLAIN stores, s_waitcnt vmcnt(0), publishes a flag (relaxed agent-scope atomic);
//            the reader - which read the same lines in the previous round, so its L1 and L2 hold them - reads them back with
//              protocol 0: plain loads    1: sc0 loads (workgroup scope)    2: sc1 loads (agent scope: miss the CU's L1; the XCD's L2 serves its own dirty lines)
//            for pairs on the same XCD (workgroups b and b + 8) and for pairs on different XCDs (b and b + 1).  Stores and loads are inline assembly:
//            a C++ `volatile` access is compiled to sc0 sc1 (system scope) and would measure something else.
// Stand-alone test infrastructure: not linked into librtow_hip.so.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

__global__ void __launch_bounds__(1024) census(unsigned* out)
{
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) { lds[0] = 1; out[blockIdx.x] = xcc_id(); }
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int PROTO>
__global__ void __launch_bounds__(64) probe(unsigned* data, unsigned* flag, unsigned* ack, unsigned long long* result, unsigned* where, int rounds, int stride)
{
    // pairs: (b, b + stride) inside groups of 2 * stride consecutive workgroups
    const int group = blockIdx.x / (2 * stride), within = blockIdx.x % (2 * stride);
    const bool writer = within < stride;
    const int pair = group * stride + (within % stride);
    if (threadIdx.x == 0) where[blockIdx.x] = xcc_id();
    unsigned* d = data + (size_t)pair * 256 + threadIdx.x * 4;
    unsigned stale = 0;
    for (int r = 1; r <= rounds; r++) {
        if (writer) {
            const u4 v = u4{(unsigned)r, (unsigned)r, (unsigned)r, (unsigned)r};
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1\n\ts_waitcnt vmcnt(0)" : : "v"(d), "v"(v) : "memory");      // plain (write-back) store
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (threadIdx.x == 0) __hip_atomic_store(flag + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) while (__hip_atomic_load(ack + pair * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            if (threadIdx.x == 0) while (__hip_atomic_load(flag + pair * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            u4 q;
            if (PROTO == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(d) : "memory");
            else if (PROTO == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(d) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(d) : "memory");
            stale += (q.x != (unsigned)r) + (q.y != (unsigned)r) + (q.z != (unsigned)r) + (q.w != (unsigned)r);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (threadIdx.x == 0) __hip_atomic_store(ack + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!writer) atomicAdd(&result[0], (unsigned long long)stale);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount, rounds = 20000;
    unsigned *data, *flag, *ack, *where;
    unsigned long long* result;
    hipMalloc(&data, (size_t)cus * 1024); hipMalloc(&flag, cus * 128); hipMalloc(&ack, cus * 128); hipMalloc(&result, 64); hipMalloc(&where, cus * 4);
    // census of a persistent launch like the sample kernel's
    hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(census, dim3(cus), dim3(1024), 140 * 1024, 0, where);
    hipDeviceSynchronize();
    std::vector<unsigned> w(cus);
    hipMemcpy(w.data(), where, cus * 4, hipMemcpyDeviceToHost);
    int count[16] = {0}, roundRobin = 1;
    for (int b = 0; b < cus; b++) { count[w[b] & 15]++; if (w[b] != w[b % 8]) roundRobin = 0; }
    printf("{\"workgroups\": %d, \"per_xcc\": [", cus);
    for (int x = 0; x < 16; x++) printf("%d%s", count[x], x < 15 ? ", " : "");
    printf("], \"xcc_of_first_8\": [");
    for (int b = 0; b < 8; b++) printf("%u%s", w[b], b < 7 ? ", " : "");
    printf("], \"xcc_is_a_function_of_workgroup_index_mod_8\": %s,\n \"litmus\": [\n", roundRobin ? "true" : "false");
    for (int stride = 8; stride >= 1; stride -= 7) {                 // 8: same XCD under round-robin dispatch; 1: neighbouring XCDs
        for (int proto = 0; proto < 3; proto++) {
            hipMemset(data, 0, (size_t)cus * 1024); hipMemset(flag, 0, cus * 128); hipMemset(ack, 0, cus * 128); hipMemset(result, 0, 64);
            if (proto == 0) hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(64), 0, 0, data, flag, ack, result, where, rounds, stride);
            else if (proto == 1) hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(64), 0, 0, data, flag, ack, result, where, rounds, stride);
            else hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(64), 0, 0, data, flag, ack, result, where, rounds, stride);
            hipDeviceSynchronize();
            unsigned long long h = 0;
            hipMemcpy(&h, result, 8, hipMemcpyDeviceToHost);
            hipMemcpy(w.data(), where, cus * 4, hipMemcpyDeviceToHost);
            int same = 0, pairs = cus / 2;
            for (int b = 0; b < cus; b++) { const int within = b % (2 * stride); if (within < stride && w[b] == w[b + stride]) same++; }
            printf("  {\"pair_stride\": %d, \"pairs\": %d, \"pairs_on_one_xcd\": %d, \"reader_loads\": \"%s\", \"stale_dwords\": %llu, \"of\": %.0f}%s\n", stride, pairs, same,
                   proto == 0 ? "plain" : proto == 1 ? "sc0" : "sc1", h, (double)pairs * rounds * 256, (stride == 1 && proto == 2) ? "" : ",");
            fflush(stdout);
        }
    }
    printf("]}\n");
    return 0;
}
