// Test-only library: the product's hit-list sorts (sort_hit_list / sort_hit_list_spilled, raytracing-in-one-weekend_amd/csrc/rtow_sample_kernel.hip.h) run
// on the device over caller-supplied key lists, one list per lane, with the first 24 entries in the lane's own arrays and the rest in its column of
// a spill area laid out like the product's ([entry][lane]).  tests/test_gpu_hitsort.py compares the permutation each lane ends with against the
// oracle's NativeSortExtension restatement (oracle_kat_unity_sort), including inputs that drive the introsort to its heap-sort fallback.
#include "../../raytracing-in-one-weekend_amd/csrc/rtow_sample_kernel.hip.h"

namespace rtow {
namespace {

__global__ void sort_lists_kernel(const float* keys, int n, int lists, int* idsOut, uint4* spillArea, unsigned stride, const unsigned* rank)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lists) return;
    float hitT[kLocalHits], hitTmin0[kLocalHits];
    unsigned hitCode[kLocalHits];
    const HitSpill spill{spillArea + lane, stride, n > kLocalHits ? (unsigned)(n - kLocalHits) : 0u};
    for (int i = 0; i < n; i++) hit_set(hitT, hitTmin0, hitCode, spill, i, HitRec{keys[(size_t)lane * n + i], (float)i, (unsigned)i});
    if (n > kLocalHits) sort_hit_list_spilled(hitT, hitTmin0, hitCode, spill, n, rank);
    else if (n > 1) sort_hit_list(hitT, hitTmin0, hitCode, n, rank);
    for (int i = 0; i < n; i++) {
        const HitRec r = hit_get(hitT, hitTmin0, hitCode, spill, i);
        // the three fields of an entry must travel together
        idsOut[(size_t)lane * n + i] = (r.tmin0 == (float)r.code && r.t == keys[(size_t)lane * n + r.code]) ? (int)r.code : -1;
    }
}

} // namespace
} // namespace rtow

// keys: lists x n floats (list after list); idsOut: the source index of every sorted position, -1 where an entry came apart.
// The lists start in index order (rank = identity), like a hit list that is already in leaf order.
extern "C" __attribute__((visibility("default"))) int hitsort_run(const float* keys, int n, int lists, int* idsOut)
{
    if (n < 1 || n > 65536 || lists < 1) return -1;
    float* dKeys = nullptr; int* dIds = nullptr; uint4* dSpill = nullptr; unsigned* dRank = nullptr;
    const size_t count = (size_t)n * lists;
    const unsigned stride = (unsigned)((lists + 63) / 64 * 64);
    std::vector<unsigned> rank(65536);
    for (unsigned i = 0; i < 65536; i++) rank[i] = i;
    bool ok = hipMalloc(&dKeys, count * 4) == hipSuccess && hipMalloc(&dIds, count * 4) == hipSuccess && hipMalloc(&dRank, 65536 * 4) == hipSuccess;
    if (ok && n > rtow::kLocalHits) ok = hipMalloc(&dSpill, (size_t)(n - rtow::kLocalHits) * stride * sizeof(uint4)) == hipSuccess;
    ok = ok && hipMemcpy(dKeys, keys, count * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dRank, rank.data(), 65536 * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        rtow::sort_lists_kernel<<<(lists + 63) / 64, 64>>>(dKeys, n, lists, dIds, dSpill, stride, dRank);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(idsOut, dIds, count * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(dKeys); (void)hipFree(dIds); (void)hipFree(dSpill); (void)hipFree(dRank);
    return ok ? 0 : -2;
}
