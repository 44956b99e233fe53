// coherence_probe.hip - litmus test for handing data between workgroups on DIFFERENT XCDs inside one running kernel (gfx950: eight XCDs,
// one L2 each).  The chained sample batches (rtowSampleBatchChainDevice) let a lane start batch b + 1 of a pixel chunk as soon as batch b
// of that chunk is stored, possibly by a workgroup on another XCD; this probe pins which access protocol makes that hand-off correct:
//   0  plain stores / plain loads, flag by relaxed agent-scope atomics, NO fences           (control: expected to read stale lines)
//   1  plain stores, release fence (agent) + atomic flag | atomic flag load, acquire fence (agent), plain loads   (buffer_wbl2 sc1 / buffer_inv sc1)
//   2  agent-scope relaxed atomic stores (sc1), workgroup release fence (s_waitcnt) + atomic flag | atomic flag load, agent-scope relaxed
//      atomic loads (sc1): no cache-wide write-back or invalidate
// Writer i and reader i are consecutive workgroups (round-robin dispatch puts them on different XCDs).  Every round the writer fills 1 KiB
// with the round number and publishes it; the reader, whose L2 still holds the previous round's lines, must see the new values.
// Stand-alone test infrastructure: not linked into librtow_hip.so.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PROTO>
__global__ void __launch_bounds__(64) probe(unsigned* data, unsigned* flag, unsigned* ack, unsigned long long* result, int rounds)
{
    const int pair = blockIdx.x >> 1;
    const bool writer = (blockIdx.x & 1) == 0;
    unsigned* d = data + (size_t)pair * 256 + threadIdx.x * 4;      // 64 lanes x 16 B = 1 KiB per pair
    unsigned stale = 0;
    unsigned long long cycles = 0;
    for (int r = 1; r <= rounds; r++) {
        if (writer) {
            const unsigned long long t0 = wall_clock64();
            if (PROTO == 2) {
                for (int k = 0; k < 4; k++) __hip_atomic_store(d + k, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            } else {
                *reinterpret_cast<uint4*>(d) = make_uint4(r, r, r, r);
                if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            if (threadIdx.x == 0) __hip_atomic_store(flag + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cycles += wall_clock64() - t0;
            if (threadIdx.x == 0) while (__hip_atomic_load(ack + pair * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            if (threadIdx.x == 0) while (__hip_atomic_load(flag + pair * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (wave-level: lane 0's loop exit orders the other lanes' loads after it)
            const unsigned long long t0 = wall_clock64();
            unsigned v[4];
            if (PROTO == 2) {
                for (int k = 0; k < 4; k++) v[k] = __hip_atomic_load(d + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 q = *reinterpret_cast<volatile u4*>(d);              // volatile: re-read every round, a plain (non-atomic) 16-byte load
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            }
            cycles += wall_clock64() - t0;
            for (int k = 0; k < 4; k++) stale += v[k] != (unsigned)r ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (threadIdx.x == 0) __hip_atomic_store(ack + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    atomicAdd(&result[writer ? 1 : 0], writer ? cycles : (unsigned long long)stale);
    if (!writer) atomicAdd(&result[2], cycles);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int pairs = prop.multiProcessorCount / 2, rounds = 20000;
    unsigned *data, *flag, *ack;
    unsigned long long* result;
    hipMalloc(&data, (size_t)pairs * 1024); hipMalloc(&flag, pairs * 128); hipMalloc(&ack, pairs * 128); hipMalloc(&result, 64);
    printf("{\"pairs\": %d, \"rounds\": %d, \"protocols\": [\n", pairs, rounds);
    for (int proto = 0; proto < 3; proto++) {
        hipMemset(data, 0, (size_t)pairs * 1024); hipMemset(flag, 0, pairs * 128); hipMemset(ack, 0, pairs * 128); hipMemset(result, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (proto == 0) hipLaunchKernelGGL(probe<0>, dim3(pairs * 2), dim3(64), 0, 0, data, flag, ack, result, rounds);
        if (proto == 1) hipLaunchKernelGGL(probe<1>, dim3(pairs * 2), dim3(64), 0, 0, data, flag, ack, result, rounds);
        if (proto == 2) hipLaunchKernelGGL(probe<2>, dim3(pairs * 2), dim3(64), 0, 0, data, flag, ack, result, rounds);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3];
        hipMemcpy(h, result, sizeof(h), hipMemcpyDeviceToHost);
        const double per = (double)pairs * rounds * 64;
        printf("  {\"protocol\": %d, \"stale_dwords\": %llu, \"of\": %.0f, \"writer_ns_per_round\": %.1f, \"reader_ns_per_round\": %.1f, \"kernel_ms\": %.2f}%s\n",
               proto, h[0], per * 4, (double)h[1] * 10.0 / per, (double)h[2] * 10.0 / per, ms, proto < 2 ? "," : "");
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}
