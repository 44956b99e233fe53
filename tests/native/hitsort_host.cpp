// Test-only: the product's hit-list sorts compiled for the HOST.  tests/test_hitsort_host.py cuts the text between the
// "[hit-list sorts: begin]" / "[hit-list sorts: end]" markers out of raytracing-in-one-weekend_amd/csrc/rtow_sample_kernel.hip.h into
// tests/build/hitsort_extracted.inc; this file supplies the few device names that text uses, so the very same source runs under
// g++ -fsanitize=undefined and is compared with the oracle's NativeSortExtension restatement (and, on the GPU, tests/test_gpu_hitsort.py
// compares the device build of it).  Same list layout as the product: 24 entries in local arrays, the rest in a strided spill column.
#include <cstdint>
#include <cstring>
#include <vector>

#define __device__
#define __forceinline__ inline
#define __noinline__
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
constexpr int kLocalHitEntries = 24;
constexpr int kBlockThreads = 1024;
struct SampleKernelArgs { uint4* hitSpill; uint32_t hitSpillStride, hitSpillEntries; };
static struct { unsigned x; } blockIdx, threadIdx;

#include "../build/hitsort_extracted.inc"

extern "C" int hitsort_host_run(const float* keys, int n, int* idsOut)
{
    if (n < 1 || n > 65536) return -1;
    float hitT[kLocalHits], hitTmin0[kLocalHits];
    unsigned hitCode[kLocalHits];
    const unsigned stride = 3;
    std::vector<uint4> column(n > kLocalHits ? (size_t)(n - kLocalHits) * stride : 1);
    const HitSpill spill{column.data(), stride, n > kLocalHits ? (uint32_t)(n - kLocalHits) : 0u};
    std::vector<unsigned> rank(65536);
    for (unsigned i = 0; i < 65536; i++) rank[i] = i;
    for (int i = 0; i < n; i++) hit_set(hitT, hitTmin0, hitCode, spill, i, HitRec{keys[i], (float)i, (unsigned)i});
    if (n > kLocalHits) sort_hit_list_spilled(hitT, hitTmin0, hitCode, spill, n, rank.data());
    else if (n > 1) sort_hit_list(hitT, hitTmin0, hitCode, n, rank.data());
    for (int i = 0; i < n; i++) {
        const HitRec r = hit_get(hitT, hitTmin0, hitCode, spill, i);
        idsOut[i] = (r.tmin0 == (float)r.code && r.code < (unsigned)n && r.t == keys[r.code]) ? (int)r.code : -1;
    }
    return 0;
}
