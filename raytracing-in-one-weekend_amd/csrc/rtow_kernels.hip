// rtow_kernels.hip - the small kernels around the sample-batch megakernel (rtow_sample_kernel.hip.h): per-sample record fold, camera-ray
// node lists, chunk ordering, scene preparation and the post passes (CombineJob, FinalizeTexturesJob, ReduceMetricsJob), and the host
// launchers of all of them.
#include "rtow_sample_kernel.hip.h"
#include "rtow_finalize.hip.h"

namespace rtow {

namespace {

// The float3 streams of the post passes (normal, albedo, combined colour: 12 B per pixel, tightly packed) move as ONE 12-byte access per lane
// (global_load_dwordx3 / global_store_dwordx3: lane stride 12 B, a wave's instruction covers 768 contiguous bytes) instead of three 4-byte
// accesses with a 12-byte lane stride, which cost three times the address / tag work for the same bytes (rocprofv3, profiles/r03_post_passes.json).
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ V3 load3(const float* p, size_t index) { const F3 v = reinterpret_cast<const F3*>(p)[index]; return v3(v.x, v.y, v.z); }
__device__ __forceinline__ void store3(float* p, size_t index, V3 v) { F3 o; o.x = v.x; o.y = v.y; o.z = v.z; reinterpret_cast<F3*>(p)[index] = o; }

// Launch order of the 64-pixel ticket chunks: most expensive first (longest-processing-time-first), from the per-chunk ray
// counts of the previous launch.  Counting sort on a 1024-bucket quantisation of the cost; the order inside a bucket is
// arbitrary - it only changes which lane renders which pixel, never a result.
constexpr int kOrderBuckets = 1024;
constexpr unsigned kOrderSingleBlockChunks = 65536u;      // beyond this many chunks the order is built by several workgroups (order_*_kernel)
static_assert(kChunkOrderScratchWords >= kOrderBuckets + 1, "histogram + maximum");
// cost[0..n) = ray count per chunk, cost[n..2n) = ray count of the chunk's most expensive pixel.  byMax: order by the most
// expensive pixel first (what decides when the last lanes retire), ties by the chunk total; otherwise by the total (used for
// the 1-sample probe, whose per-pixel counts are too noisy).
__device__ __forceinline__ float chunk_key(const unsigned* cost, unsigned n, unsigned i, int byMax)
{
    return byMax ? (float)cost[n + i] * 64.0f + (float)cost[i] * (1.0f / 64.0f) : (float)cost[i];
}
// ------------------------------------------------------------------------------------------------------------
// RTOW_RNG_PER_SAMPLE: the sample kernel's units (pixel, group of 16 samples) leave one 64-byte record each; this kernel adds a
// pixel's records to its accumulators IN GROUP ORDER (so the result does not depend on which lane ran which unit) and writes the
// outputs and diagnostics of SampleBatchJob.Execute (JOBS/SampleBatchJob.cs:159-163).
//   record = {colour.xyz, successes | normal.xyz, rays | albedo.xyz, sampleCountWeight | boundsHits, candidates, -, -};
//   a group-0 record with no success carries the AOVs of the batch's sample 0 (the fallback, :152-156) instead of sums.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_unit_records_kernel(SampleKernelArgs A)
{
    const unsigned ticket = blockIdx.x * blockDim.x + threadIdx.x;       // owned pixel
    const unsigned pixels = A.totalWork / A.groupsPerPixel;
    if (ticket >= pixels) return;
    // a cancelled batch leaves units that were never pulled with stale records: like the reference, whose cancelled Execute returns before
    // any write (JOBS/SampleBatchJob.cs:61-62), nothing is folded then (the flag stays set until the next batch is enqueued)
    if (A.cancelFlag && *A.cancelFlag != 0u) return;
    int cx, ownedRow;
    owned_pixel_xy(ticket, (unsigned)A.width, A.tilesPerRow, A.tiledPixels, cx, ownedRow);      // the sample kernel's numbering of the owned pixels
    const int cy = A.sliceOffset + ownedRow * A.sliceDivider;
    const size_t pix = (size_t)cy * (size_t)A.width + (size_t)cx;
    const float4 last = reinterpret_cast<const float4*>(A.inColor)[pix];
    V3 color = v3(last.x, last.y, last.z);
    V3 normal = load3(A.inNormal, pix);
    V3 albedo = load3(A.inAlbedo, pix);
    float scw = A.inScw[pix];
    int count = (int)last.w;
    const float weight = scw / (float)count;                              // Diagnostics.SampleCountWeight (:128-130)
    float rays = 0, bounds = 0, cands = 0;
    V3 fbNormal = v3(0, 0, 0), fbAlbedo = v3(0, 0, 0);
    const float4* rec = reinterpret_cast<const float4*>(A.unitRecords) + (size_t)ticket * A.groupsPerPixel * 4u;
    for (unsigned g = 0; g < A.groupsPerPixel; g++, rec += 4) {
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        if (r0.w > 0) {
            color = add(color, v3(r0.x, r0.y, r0.z));
            normal = add(normal, v3(r1.x, r1.y, r1.z));
            albedo = add(albedo, v3(r2.x, r2.y, r2.z));
            count += (int)r0.w;
        } else if (g == 0) {
            fbNormal = v3(r1.x, r1.y, r1.z);
            fbAlbedo = v3(r2.x, r2.y, r2.z);
        }
        scw += r2.w;
        rays += r1.w;
        bounds += r3.x;
        cands += r3.y;
    }
    reinterpret_cast<float4*>(A.outColor)[pix] = make_float4(color.x, color.y, color.z, (float)count);
    const V3 on = count == 0 ? fbNormal : normal, oa = count == 0 ? fbAlbedo : albedo;
    store3(A.outNormal, pix, on);
    store3(A.outAlbedo, pix, oa);
    A.outScw[pix] = scw;
    if (A.diagnostics) {
        if (A.diagnosticsStride >= 16) *reinterpret_cast<float4*>(A.diagnostics + pix * 16u) = make_float4(rays, bounds, cands, weight);
        else *reinterpret_cast<float*>(A.diagnostics + pix * 4u) = rays;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Camera-ray node lists.  All camera rays of one pixel (every sample: jitter inside the pixel, origin inside the
// lens disk) stay inside a thin beam around the pixel's centre ray, and 40 % of all rays are camera rays: one
// CONSERVATIVE walk of the beam per pixel finds every primitive any of them can hit, and the sample kernel then skips the
// box walk for depth-0 rays.  One thread per owned pixel, neighbouring pixels walk the same nodes (fully coherent).
//
// Bound.  A camera ray is p(s) = o + off + s * u with |off| <= R (lens), u = normalize(G - off), G = LLC + u*H + v*V inside the
// pixel.  With dc = normalize(G_centre):  |u - dc| <= |normalize(G) - dc| + |normalize(G - off) - normalize(G)| <= chordPix + R / (|G| - R),
// so the point q(s) = o + s * dc of the centre ray is within  R + s * (chordPix + chordLens)  of p(s).  A hit inside box B happens
// at s <= far(B) + R (far = distance from o to B's farthest corner), hence the centre ray passes through B grown by
// delta(B) = R + (far + R) * (chordPix + chordLens) on every side - tested with a slab test, tMin = 0.  chordPix is exact at the
// pixel's corners (a cone cut by the image plane is convex).  Lists longer than 4 fall back to the normal walk (kNoPrimaryList).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) primary_candidates_kernel(SampleKernelArgs A, uint2* __restrict__ out)
{
    const unsigned ticket = blockIdx.x * blockDim.x + threadIdx.x;
    if (ticket >= A.totalWork) return;
    const int ownedRow = (int)(ticket / (unsigned)A.width);
    const int cx = (int)(ticket - (unsigned)ownedRow * (unsigned)A.width);
    const int cy = A.sliceOffset + ownedRow * A.sliceDivider;
    const int pix = cy * A.width + cx;
    const SceneLayout& L = A.layout;

    const V3 o = v3(A.view.origin);
    const V3 llc = v3(A.view.lowerLeftCorner), hv = v3(A.view.horizontal), vv = v3(A.view.vertical);
    auto towards = [&](float u, float v) { return v3(llc.x + u * hv.x + v * vv.x, llc.y + u * hv.y + v * vv.y, llc.z + u * hv.z + v * vv.z); };
    const float u0 = A.subPixelJitter ? (float)cx / A.sizeX : ((float)cx + 0.5f) / A.sizeX, u1 = A.subPixelJitter ? ((float)cx + 1.0f) / A.sizeX : u0;
    const float v0 = A.subPixelJitter ? (float)cy / A.sizeY : ((float)cy + 0.5f) / A.sizeY, v1 = A.subPixelJitter ? ((float)cy + 1.0f) / A.sizeY : v0;
    const V3 gc = towards(0.5f * (u0 + u1), 0.5f * (v0 + v1));
    const V3 dc = normalize(gc);
    float chord = 0, gmin = __builtin_sqrtf(dot(gc, gc));
    for (int k = 0; k < 4; k++) {
        const V3 g = towards((k & 1) ? u1 : u0, (k & 2) ? v1 : v0);
        const V3 d = sub(normalize(g), dc);
        chord = fmaxf(chord, __builtin_sqrtf(dot(d, d)));
        gmin = fminf(gmin, __builtin_sqrtf(dot(g, g)));
    }
    const V3 right = v3(A.view.right), up = v3(A.view.up);
    const float R = A.view.lensRadius == 0 ? 0.0f : __builtin_fabsf(A.view.lensRadius) * (__builtin_sqrtf(dot(right, right)) + __builtin_sqrtf(dot(up, up))) * 1.001f;
    bool ok = gmin > 4.0f * R && chord == chord && gmin == gmin;               // degenerate views: walk normally
    const float chordLens = R > 0 ? R / (gmin - R) : 0.0f;
    const float spread = (chord + chordLens) * 1.02f + 2e-6f;                   // + slack for the fp32 evaluation of the directions
    auto clampInv = [](float d) { const float r = 1.0f / d; return fminf(fmaxf(r, -1e30f), 1e30f); };   // no inf: 0 * inf would be NaN
    const V3 inv = v3(clampInv(dc.x), clampInv(dc.y), clampInv(dc.z));

    auto beamHitsBox = [&](float lx, float ly, float lz, float hx, float hy, float hz) {
        const float fx = fmaxf(__builtin_fabsf(lx - o.x), __builtin_fabsf(hx - o.x));
        const float fy = fmaxf(__builtin_fabsf(ly - o.y), __builtin_fabsf(hy - o.y));
        const float fz = fmaxf(__builtin_fabsf(lz - o.z), __builtin_fabsf(hz - o.z));
        const float far = __builtin_sqrtf(fx * fx + fy * fy + fz * fz);
        const float delta = (R + (far + R) * spread) * 1.01f + 1e-5f * (1.0f + far);
        const float t0x = (lx - delta - o.x) * inv.x, t1x = (hx + delta - o.x) * inv.x;
        const float t0y = (ly - delta - o.y) * inv.y, t1y = (hy + delta - o.y) * inv.y;
        const float t0z = (lz - delta - o.z) * inv.z, t1z = (hz + delta - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
        const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
        return !(tn > tf);                                                      // NaN (cannot happen with finite inv) would count as a hit
    };

    const unsigned none = A.wideCodes ? 0xffffffffu : 0xffffu;
    const int capacity = (A.wideCodes || L.sceneKind > SCENE_KIND_SPHERES_MOTION) ? 4 : 8;      // what one uint4 per pixel holds - and what the scene kind's kernels keep of it (sphere kinds: all eight)
    unsigned list[8] = {none, none, none, none, none, none, none, none};
    int count = 0;
    int stack[RTOW_STACK_CAPACITY + 1];
    int sp = 0, cur = 0;
    const GpuNode* nodes = reinterpret_cast<const GpuNode*>(A.sceneBlob + L.nodeOffset);
    while (ok && cur >= 0) {
        const int node = cur;
        const GpuNode n = nodes[node];
        cur = -1;
        bool leafChildInBeam = false;
        for (int side = 0; side < 2; side++) {
            if (side == 1 && L.sphereCount < 2) break;                          // single-entity scene: the second child is a placeholder
            const int child = side ? n.child1 : n.child0;
            if (!beamHitsBox(n.lox[side], n.loy[side], n.loz[side], n.hix[side], n.hiy[side], n.hiz[side])) continue;
            if (child < 0) leafChildInBeam = true;
            else if (cur < 0) cur = child;
            else if (sp <= RTOW_STACK_CAPACITY) stack[sp++] = child;
            else ok = false;
        }
        if (leafChildInBeam) {                                                  // the NODE goes on the list: its leaf boxes are re-tested per ray
            if (count == capacity) ok = false;
            for (int k = 0; k < 8; k++) if (k == count) list[k] = (unsigned)node;
            count++;
        }
        if (cur < 0 && sp > 0) cur = stack[--sp];
    }
    if (A.wideCodes) reinterpret_cast<uint4*>(out)[pix] = ok ? make_uint4(list[0], list[1], list[2], list[3]) : make_uint4(0xffffffffu, 0u, 0u, 0u);   // first slot empty, second not: no list
    else reinterpret_cast<uint4*>(out)[pix] = ok ? make_uint4(list[0] | (list[1] << 16), list[2] | (list[3] << 16), list[4] | (list[5] << 16), list[6] | (list[7] << 16)) : make_uint4(kNoPrimaryList, 0u, 0u, 0u);
}

// pixelCost[64 * chunk .. +63] -> cost[chunk] = sum, cost[n + chunk] = max; one wave per chunk
__global__ void __launch_bounds__(256) reduce_chunk_cost_kernel(const unsigned short* __restrict__ pixelCost, unsigned n, unsigned* __restrict__ cost)
{
    const unsigned chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (chunk >= n) return;
    unsigned v = pixelCost[(size_t)chunk * 64u + (threadIdx.x & 63u)];
    unsigned sum = v, mx = v;
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        const unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63u) == 0) { cost[chunk] = sum; cost[n + chunk] = mx; }
}
__global__ void __launch_bounds__(1024) build_chunk_order_kernel(const unsigned* __restrict__ cost, unsigned n, unsigned* __restrict__ order, int byMax)
{
    __shared__ unsigned hist[kOrderBuckets];
    __shared__ unsigned maxKeyBits;
    const unsigned t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) maxKeyBits = __float_as_uint(1.0f);
    __syncthreads();
    float m = 0;
    for (unsigned i = t; i < n; i += 1024) m = fmaxf(m, chunk_key(cost, n, i, byMax));
    atomicMax(&maxKeyBits, __float_as_uint(m));                     // non-negative floats order like their bit patterns
    __syncthreads();
    const float scale = (float)(kOrderBuckets - 1) / __uint_as_float(maxKeyBits);
    for (unsigned i = t; i < n; i += 1024) atomicAdd(&hist[(kOrderBuckets - 1) - (unsigned)(chunk_key(cost, n, i, byMax) * scale)], 1u);   // bucket 0 = most expensive
    __syncthreads();
    if (t == 0) {
        unsigned run = 0;
        for (int b = 0; b < kOrderBuckets; b++) { const unsigned c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (unsigned i = t; i < n; i += 1024) order[atomicAdd(&hist[(kOrderBuckets - 1) - (unsigned)(chunk_key(cost, n, i, byMax) * scale)], 1u)] = i;
}

// The same order over several workgroups, for launches of hundreds of thousands of chunks (the per-sample policies' units at 1080p: 518 400 chunks, 0.67 ms in the one
// workgroup above = 1.1 % of a batch): maximum, histogram, prefix and scatter as four small launches over a 1 025-word scratch behind the cost array
// (scratch[0 .. 1023] = histogram / cursors, scratch[1024] = bits of the largest key).  Same buckets, same (unspecified) order inside a bucket.
__global__ void __launch_bounds__(256) order_max_kernel(const unsigned* __restrict__ cost, unsigned n, int byMax, unsigned* __restrict__ scratch)
{
    float m = 1.0f;                                                  // (the single-workgroup kernel starts from 1 too)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) m = fmaxf(m, chunk_key(cost, n, i, byMax));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63u) == 0) atomicMax(scratch + kOrderBuckets, __float_as_uint(m));
}
// (a workgroup counts its 2 048 chunks in LDS and touches every global bucket once: the keys of a launch crowd into a few buckets - per-sample units cost alike - and half a
// million global atomics on a handful of addresses took longer than the single workgroup they were meant to replace)
constexpr unsigned kOrderChunksPerBlock = 2048u;
__global__ void __launch_bounds__(256) order_hist_kernel(const unsigned* __restrict__ cost, unsigned n, int byMax, unsigned* __restrict__ scratch)
{
    __shared__ unsigned hist[kOrderBuckets];
    for (unsigned b = threadIdx.x; b < (unsigned)kOrderBuckets; b += 256u) hist[b] = 0u;
    __syncthreads();
    const float scale = (float)(kOrderBuckets - 1) / __uint_as_float(scratch[kOrderBuckets]);
    const unsigned first = blockIdx.x * kOrderChunksPerBlock, last = first + kOrderChunksPerBlock < n ? first + kOrderChunksPerBlock : n;
    for (unsigned i = first + threadIdx.x; i < last; i += 256u) atomicAdd(&hist[(kOrderBuckets - 1) - (unsigned)(chunk_key(cost, n, i, byMax) * scale)], 1u);
    __syncthreads();
    for (unsigned b = threadIdx.x; b < (unsigned)kOrderBuckets; b += 256u) if (hist[b]) atomicAdd(&scratch[b], hist[b]);
}
__global__ void __launch_bounds__(1024) order_prefix_kernel(unsigned* __restrict__ scratch)
{
    __shared__ unsigned sums[kOrderBuckets];
    const unsigned t = threadIdx.x, c = scratch[t];
    sums[t] = c;
    __syncthreads();
    for (unsigned d = 1; d < (unsigned)kOrderBuckets; d <<= 1) {      // inclusive scan
        const unsigned v = t >= d ? sums[t - d] : 0u;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    scratch[t] = sums[t] - c;                                          // exclusive: where bucket t starts
}
__global__ void __launch_bounds__(256) order_scatter_kernel(const unsigned* __restrict__ cost, unsigned n, int byMax, unsigned* __restrict__ scratch, unsigned* __restrict__ order)
{
    __shared__ unsigned hist[kOrderBuckets];       // this workgroup's count per bucket, then its cursor inside the range it reserved in that bucket
    for (unsigned b = threadIdx.x; b < (unsigned)kOrderBuckets; b += 256u) hist[b] = 0u;
    __syncthreads();
    const float scale = (float)(kOrderBuckets - 1) / __uint_as_float(scratch[kOrderBuckets]);
    const unsigned first = blockIdx.x * kOrderChunksPerBlock, last = first + kOrderChunksPerBlock < n ? first + kOrderChunksPerBlock : n;
    for (unsigned i = first + threadIdx.x; i < last; i += 256u) atomicAdd(&hist[(kOrderBuckets - 1) - (unsigned)(chunk_key(cost, n, i, byMax) * scale)], 1u);
    __syncthreads();
    for (unsigned b = threadIdx.x; b < (unsigned)kOrderBuckets; b += 256u) if (hist[b]) hist[b] = atomicAdd(&scratch[b], hist[b]);      // reserve; hist[b] = where this workgroup's share starts
    __syncthreads();
    for (unsigned i = first + threadIdx.x; i < last; i += 256u) order[atomicAdd(&hist[(kOrderBuckets - 1) - (unsigned)(chunk_key(cost, n, i, byMax) * scale)], 1u)] = i;
}

// ------------------------------------------------------------------------------------------------------------
// Which pixels share a wave (SampleKernelArgs.ticketMap).  A chunk's 64 tickets go to the 64 lanes of one wave; numbered in tiles they are an 8 x 8 block of the image - but a tile on
// a sphere's edge holds sky pixels (one ray per sample), ground pixels and glass pixels, whose lanes wait for each other's stages.  The reference hands pixels out in no
// defined order (Schedule(W * H, 1), UNITY/Raytracer.cs:730), so the numbering is free: per super-tile of side x side tiles this kernel sorts the super-tile's pixels by the ray
// count the last launch measured for them - most expensive first; equal counts (sky: exactly one ray per sample) keep their tile order - and deals them out to the super-tile's own
// chunks 64 at a time.  A super-tile's accumulator lines stay with the few CUs that take its chunks.  One workgroup per super-tile, bitonic sort in LDS of
// (cost + 1) << 32 | owned-pixel number (0 = padding).  In place: ticketMap and pixelCost (permuted along, so that it stays in ticket order under the new map).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) init_ticket_map_kernel(unsigned* __restrict__ map, unsigned n)
{
    const unsigned i = blockIdx.x * 1024u + threadIdx.x;
    if (i < n) map[i] = i;
}
__global__ void __launch_bounds__(1024) regroup_tickets_kernel(unsigned short* __restrict__ pixelCost, unsigned* __restrict__ ticketMap, unsigned tilesPerRow, unsigned tileRows, unsigned side,
                                                              unsigned t1, unsigned t2, unsigned t3)
{
    extern __shared__ unsigned long long regroupKeys[];
    const unsigned perRow = (tilesPerRow + side - 1u) / side;
    const unsigned sty = blockIdx.x / perRow, stx = blockIdx.x - sty * perRow;
    const unsigned tx0 = stx * side, ty0 = sty * side;
    const unsigned w = tilesPerRow - tx0 < side ? tilesPerRow - tx0 : side, h = tileRows - ty0 < side ? tileRows - ty0 : side;
    const unsigned n = w * h * 64u;
    unsigned N = 64u;
    while (N < n) N <<= 1;
    auto ticketOf = [&](unsigned e) { const unsigned k = e >> 6, ky = k / w, kx = k - ky * w; return ((ty0 + ky) * tilesPerRow + tx0 + kx) * 64u + (e & 63u); };
    for (unsigned e = threadIdx.x; e < N; e += 1024u) {
        unsigned long long key = 0ull;
        if (e < n) {
            const unsigned t = ticketOf(e);
            const unsigned cost = pixelCost[t];
            // key = primary << 48 | owned-pixel number << 16 | cost: the primary key is the ray count itself (t1 == 0) or its class (cost classes t1 < t2 < t3: sky only / ... ),
            // equal primaries keep their tile order, the ray count travels along
            const unsigned primary = t1 == 0u ? cost + 1u : 1u + (cost > t1 ? 1u : 0u) + (cost > t2 ? 1u : 0u) + (cost > t3 ? 1u : 0u);
            // (the sort is descending: the pixel number goes in complemented, so that equal primaries come out in ASCENDING pixel order - their tile order)
            key = ((unsigned long long)(primary > 0xffffu ? 0xffffu : primary) << 48) | ((unsigned long long)(0xffffffffu - ticketMap[t]) << 16) | (unsigned long long)cost;
        }
        regroupKeys[e] = key;
    }
    __syncthreads();
    for (unsigned k = 2u; k <= N; k <<= 1) {
        for (unsigned j = k >> 1; j > 0u; j >>= 1) {
            for (unsigned e = threadIdx.x; e < N; e += 1024u) {
                const unsigned p = e ^ j;
                if (p > e) {
                    const unsigned long long a = regroupKeys[e], b = regroupKeys[p];
                    const bool descending = (e & k) == 0u;                       // the whole array ends up descending: padding (0) last
                    if (descending ? a < b : a > b) { regroupKeys[e] = b; regroupKeys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (unsigned e = threadIdx.x; e < n; e += 1024u) {
        const unsigned long long key = regroupKeys[e];
        const unsigned t = ticketOf(e);
        ticketMap[t] = 0xffffffffu - (unsigned)(key >> 16);
        pixelCost[t] = (unsigned short)(key & 0xffffull);
    }
}

// The same map with the tiles left as they are: only the ORDER of a tile's 64 tickets changes - most expensive pixel first.  A wave's lanes take the tickets of its chunk one by one
// as they finish their previous pixels (the first tickets go within microseconds, the last ones when the slowest lanes of the previous chunk are done, a pixel-time later), so a chunk's
// expensive pixel that happens to be its 60th ticket starts tens of milliseconds after the chunk was handed out - and ends the launch.  The wave still traces the same 64 neighbours.
// One wave per tile: 64 keys (cost + 1) << 32 | owned-pixel number, bitonic sort through cross-lane moves, descending.
__global__ void __launch_bounds__(256) order_tile_tickets_kernel(unsigned short* __restrict__ pixelCost, unsigned* __restrict__ ticketMap, unsigned tiles, unsigned levels)
{
    const unsigned tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= tiles) return;
    const unsigned t = tile * 64u + lane;
    const unsigned cost = pixelCost[t], pixel = ticketMap[t];
    // primary key: the ray count itself (levels == 0), or its level among `levels` equal parts of the tile's range [0, max] - pixels of one level keep the order they had, so
    // that a tile of like pixels (all sky; a patch of ground) is still taken row by row and its accumulator lines are touched once, and only the pixels that stand out move to the front
    unsigned top = cost;
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)top, off, 64); top = o > top ? o : top; }
    const unsigned primary = levels ? (cost * levels) / (top + 1u) : cost;
    unsigned key = (primary << 12) | ((63u - lane) << 6) | lane;                 // descending: higher level first, then the earlier place; the low six bits say where the element came from
    for (unsigned k = 2u; k <= 64u; k <<= 1) {
        for (unsigned j = k >> 1; j > 0u; j >>= 1) {
            const unsigned other = (unsigned)__shfl_xor((int)key, (int)j, 64);
            const bool upper = (lane & j) != 0u;                                 // this lane holds the later element of the pair
            const bool descending = (lane & k) == 0u;
            const bool keepLarger = descending != upper;                         // the earlier place of a descending run keeps the larger key
            key = keepLarger ? (key > other ? key : other) : (key < other ? key : other);
        }
    }
    const int from = (int)(key & 63u);
    ticketMap[t] = (unsigned)__shfl((int)pixel, from, 64);
    pixelCost[t] = (unsigned short)__shfl((int)cost, from, 64);
}

// Derived per-entity transform data for SCENE_KIND_GENERAL, on the device: InverseTransform = inverse(OriginTransform)
// (RT/Entity.cs:51-52; math.inverse(RigidTransform): invRot = inverse(rot), invTranslation = mul(invRot, -pos);
// math.inverse(quaternion q) = rcp(dot(q, q)) * q * float4(-1, -1, -1, 1)).
__global__ void prepare_entities_kernel(uint8_t* blob, SceneLayout L)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.sphereCount) return;
    const unsigned type = reinterpret_cast<const unsigned*>(blob + L.matIndexOffset)[i] >> kPrimTypeShift;
    if (type == RTOW_ENTITY_TRIANGLE) return;
    float4* p = reinterpret_cast<float4*>(blob + L.primOffset + (size_t)i * 128u);
    const float4 q = p[0];
    const float r = 1.0f / (q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float4 inv = make_float4(r * q.x * -1.0f, r * q.y * -1.0f, r * q.z * -1.0f, r * q.w * 1.0f);
    p[1] = inv;
    const float4 q2 = p[2];
    const V3 it = rotate(inv, v3(-q2.x, -q2.y, -q2.z));
    float4 q4 = p[4];
    q4.y = it.x; q4.z = it.y; q4.w = it.z;
    p[4] = q4;
}

// Derived per-material constants, computed ON THE DEVICE with the same float program the per-hit code would run
// (pow(1 - glossiness, 2), RoughnessToAlpha, lerp(PlasticIor, MetalIor, metallic), Schlick's r0, 1 / ior), so hoisting
// them out of the bounce loop cannot change a bit of the result.
__global__ void prepare_materials_kernel(uint8_t* blob, SceneLayout L)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.materialCount) return;
    GpuMaterial* m = reinterpret_cast<GpuMaterial*>(blob + L.materialOffset) + i;
    float roughness, ior, invIor = 0.0f, alpha = 0.0f;
    if (m->type == RTOW_MATERIAL_STANDARD) {
        roughness = det_sq(1 - m->glossiness);                  // RT/Material.cs:80
        ior = 1.5f + m->metallic * (1.1f - 1.5f);               // :84 lerp(PlasticIor, MetalIor, metallic)
        alpha = roughness_to_alpha(roughness);                  // RT/Microfacet.cs:72 (inside Lambda)
    } else {
        roughness = 1 - m->glossiness;                          // RT/Material.cs:123
        ior = m->parameter;
        invIor = 1 / ior;                                       // :137
    }
    float r0 = (1 - ior) / (1 + ior);                           // Schlick, :214-215
    r0 *= r0;
    m->roughness = roughness;
    m->alpha = alpha;
    m->ior = ior;
    m->r0 = r0;
    m->invIor = invIor;
}

// ------------------------------------------------------------------------------------------------------------
// post passes
// ------------------------------------------------------------------------------------------------------------

// CombineJob.Execute (JOBS/CombineJob.cs:29-71) for one pixel: sum -> mean, interlace look-around, NaN / zero-sample handling
__device__ __forceinline__ void combine_pixel(const RtowCombineParams& p, const float4* __restrict__ inColor, int index, float4 c, V3 nIn, V3 aIn, V3& finalColor, V3& nn, V3& alb)
{
    int count = (int)c.w;
    if (!p.debugMode && count == 0) {
        int tentative = index;
        while (count == 0 && (tentative -= p.width) >= 0) { // look-around for interlaced buffers (:40-50)
            c = inColor[tentative];
            count = (int)c.w;
        }
    }
    const bool anyNan = (c.x != c.x) || (c.y != c.y) || (c.z != c.z) || (c.w != c.w);
    if (count == 0) finalColor = p.debugMode ? v3(1, 0, 1) : v3(0, 0, 0);
    else if (anyNan) finalColor = p.debugMode ? v3(0, 1, 1) : v3(0, 0, 0);
    else finalColor = v3(c.x / (float)count, c.y / (float)count, c.z / (float)count);

    const float denom = (float)(count > 1 ? count : 1);
    alb = v3(aIn.x / denom, aIn.y / denom, aIn.z / denom);
    if (p.ldrAlbedo) alb = v3(um_min(alb.x, 1.0f), um_min(alb.y, 1.0f), um_min(alb.z, 1.0f));
    const V3 nv = v3(nIn.x / denom, nIn.y / denom, nIn.z / denom);
    const float len = dot(nv, nv);
    nn = v3(0, 0, 0);
    if (len > 1.175494351e-38f) { const float r = 1.0f / __builtin_sqrtf(len); nn = v3(nv.x * r, nv.y * r, nv.z * r); } // normalizesafe
}

__global__ void __launch_bounds__(256) combine_kernel(RtowCombineParams p, const float4* __restrict__ inColor, const float* __restrict__ inNormal,
                                                      const float* __restrict__ inAlbedo, float* __restrict__ outColor, float* __restrict__ outNormal,
                                                      float* __restrict__ outAlbedo)
{
    const int n = p.width * p.height;
    for (int index = (int)(blockIdx.x * blockDim.x + threadIdx.x); index < n; index += (int)(gridDim.x * blockDim.x)) {
        V3 finalColor, nn, alb;
        combine_pixel(p, inColor, index, inColor[index], load3(inNormal, (size_t)index), load3(inAlbedo, (size_t)index), finalColor, nn, alb);
        store3(outColor, (size_t)index, finalColor);
        store3(outNormal, (size_t)index, nn);
        store3(outAlbedo, (size_t)index, alb);
    }
}

// Grid of the streaming post passes: one block of 256 lanes per 256 elements, dispatched in order, streams through memory front to back.  The 2 048
// resident blocks of rounds 2 / 3 each walked the arrays with a stride of 2 048 x 256 elements - every block on the same few channels at the same time:
// at 3840 x 2160 combine 5.4 -> 5.9 - 6.2 TB/s, add 5.2 -> 5.6 - 5.7, finalize 4.9 -> 5.1 (5.3 -> 5.9 at 1080p); 1 024 / 4 096 / 8 192 blocks in between
// (gpurun_out/r03bp, r03bq).  The caps below only matter beyond 268 M elements; the kernels keep their grid-stride loops for that case.
#ifndef RTOW_COMBINE_BLOCKS
#define RTOW_COMBINE_BLOCKS 1048576
#endif
#ifndef RTOW_FINALIZE_BLOCKS
#define RTOW_FINALIZE_BLOCKS 1048576
#endif
// FinalizeTexturesJob.Execute (JOBS/FinalizeTexturesJob.cs:23-55).  The nine float -> byte conversions per pixel go through the step table
// (rtow_finalize.hip.h: same byte as the deterministic-pow form for every float operand, a fifth of its instructions), staged in LDS.
__global__ void __launch_bounds__(256, 8) finalize_kernel(int n, const float* __restrict__ inColor, const float* __restrict__ inNormal,
                                                       const float* __restrict__ inAlbedo, uchar4* __restrict__ outColor, uchar4* __restrict__ outNormal,
                                                       uchar4* __restrict__ outAlbedo, const float* __restrict__ thresholds)
{
    __shared__ float T[kByteThresholdFloats];
    for (int i = (int)threadIdx.x; i < kByteThresholdFloats; i += (int)blockDim.x) T[i] = thresholds[i];
    __syncthreads();
    const ByteZones Z = load_byte_zones(T);
    // software pipelined: the next pixel's three 12-byte loads are issued before this pixel's conversions, so a wave always has a pixel in flight
    const int stride = (int)(gridDim.x * blockDim.x);
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    V3 c = v3(0, 0, 0), nm = v3(0, 0, 0), al = v3(0, 0, 0);
    if (i < n) { c = load3(inColor, (size_t)i); nm = load3(inNormal, (size_t)i); al = load3(inAlbedo, (size_t)i); }
    while (i < n) {
        const int j = i + stride;
        V3 c2 = v3(0, 0, 0), nm2 = v3(0, 0, 0), al2 = v3(0, 0, 0);
        if (j < n) { c2 = load3(inColor, (size_t)j); nm2 = load3(inNormal, (size_t)j); al2 = load3(inAlbedo, (size_t)j); }
        // nine independent conversions (rtow_finalize.hip.h): their LDS reads are in flight together
        const float v[9] = {c.x, c.y, c.z, nm.x * 0.5f + 0.5f, nm.y * 0.5f + 0.5f, nm.z * 0.5f + 0.5f, al.x, al.y, al.z};
        unsigned b[9];
        to_bytes_table<9>(v, b, T, Z);
        outColor[i] = make_uchar4((unsigned char)b[0], (unsigned char)b[1], (unsigned char)b[2], 255);
        outNormal[i] = make_uchar4((unsigned char)b[3], (unsigned char)b[4], (unsigned char)b[5], 255);
        outAlbedo[i] = make_uchar4((unsigned char)b[6], (unsigned char)b[7], (unsigned char)b[8], 255);
        c = c2; nm = nm2; al = al2;
        i = j;
    }
}

// ReduceMetricsJob.Execute (JOBS/ReduceMetricsJob.cs:22-45) as a two-level reduction; integer sums and float min/max are
// order independent (math.min/max skip a NaN second operand), so the result equals the reference's serial loop.
__global__ void __launch_bounds__(256) reduce_metrics_kernel(int n, const uint8_t* __restrict__ diag, int stride, const float4* __restrict__ color,
                                                             const float* __restrict__ scw, MetricsPartial* __restrict__ partials)
{
    long long rays = 0, samples = 0;
    float minW = __builtin_inff(), maxW = -__builtin_inff(), minS = __builtin_inff(), maxS = -__builtin_inff();
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
        const float rc = *reinterpret_cast<const float*>(diag + (size_t)i * (size_t)stride);
        rays += (int)rc;
        const int sc = (int)color[i].w;
        samples += sc;
        const float w = scw[i] / (float)sc;
        minW = um_min(minW, w); maxW = um_max(maxW, w);
        minS = um_min(minS, (float)sc); maxS = um_max(maxS, (float)sc);
    }
    __shared__ MetricsPartial sh[256];
    MetricsPartial mine;
    mine.rays = rays; mine.samples = samples; mine.minW = minW; mine.maxW = maxW; mine.minS = minS; mine.maxS = maxS;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            MetricsPartial a = sh[threadIdx.x];
            const MetricsPartial b = sh[threadIdx.x + s];
            a.rays += b.rays; a.samples += b.samples;
            a.minW = um_min(a.minW, b.minW); a.maxW = um_max(a.maxW, b.maxW);
            a.minS = um_min(a.minS, b.minS); a.maxS = um_max(a.maxS, b.maxS);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// CombineJob + FinalizeTexturesJob in one pass (the reference's default chain with denoiseMode 0, Assets/Prefabs/Raytracer.prefab:391: combine -> finalize back to
// back, UNITY/Raytracer.cs:806-807): 40 B read and 12 B written per pixel instead of 80 + 48 through the float3 intermediates.  Same float program as the two
// kernels one after the other (combine_pixel, to_bytes_table), so the bytes equal oracle.combine -> oracle.finalize.
__global__ void __launch_bounds__(256, 8) combine_finalize_kernel(RtowCombineParams p, const float4* __restrict__ inColor, const float* __restrict__ inNormal,
                                                                  const float* __restrict__ inAlbedo, uchar4* __restrict__ outColor, uchar4* __restrict__ outNormal,
                                                                  uchar4* __restrict__ outAlbedo, const float* __restrict__ thresholds)
{
    __shared__ float T[kByteThresholdFloats];
    for (int i = (int)threadIdx.x; i < kByteThresholdFloats; i += (int)blockDim.x) T[i] = thresholds[i];
    __syncthreads();
    const ByteZones Z = load_byte_zones(T);
    const int n = p.width * p.height;
    const int stride = (int)(gridDim.x * blockDim.x);
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    float4 c4 = make_float4(0, 0, 0, 0);
    V3 nm = v3(0, 0, 0), al = v3(0, 0, 0);
    if (i < n) { c4 = inColor[i]; nm = load3(inNormal, (size_t)i); al = load3(inAlbedo, (size_t)i); }
    while (i < n) {
        const int j = i + stride;
        float4 c42 = make_float4(0, 0, 0, 0);
        V3 nm2 = v3(0, 0, 0), al2 = v3(0, 0, 0);
        if (j < n) { c42 = inColor[j]; nm2 = load3(inNormal, (size_t)j); al2 = load3(inAlbedo, (size_t)j); }      // the next pixel's loads are in flight during this pixel's conversions
        V3 c, nn, alb;
        combine_pixel(p, inColor, i, c4, nm, al, c, nn, alb);
        const float v[9] = {c.x, c.y, c.z, nn.x * 0.5f + 0.5f, nn.y * 0.5f + 0.5f, nn.z * 0.5f + 0.5f, alb.x, alb.y, alb.z};
        unsigned b[9];
        to_bytes_table<9>(v, b, T, Z);
        outColor[i] = make_uchar4((unsigned char)b[0], (unsigned char)b[1], (unsigned char)b[2], 255);
        outNormal[i] = make_uchar4((unsigned char)b[3], (unsigned char)b[4], (unsigned char)b[5], 255);
        outAlbedo[i] = make_uchar4((unsigned char)b[6], (unsigned char)b[7], (unsigned char)b[8], 255);
        c4 = c42; nm = nm2; al = al2;
        i = j;
    }
}

// rtowAddAccumDevice: the four accumulators in ONE launch (round 3: four).  Blocks [0, b0) add colour, [b0, b1) normal, ... - each range a flat stream of
// 16-byte (or, unaligned / the tail, 4-byte) units, dispatched front to back.
struct AddSpan { float* dst; const float* src; size_t floats; unsigned firstBlock; unsigned wide; };
struct AddSpans { AddSpan s[4]; };
__global__ void __launch_bounds__(256) add_accum_kernel(AddSpans a)
{
    int k = 0;
    if (blockIdx.x >= a.s[1].firstBlock) k = 1;
    if (blockIdx.x >= a.s[2].firstBlock) k = 2;
    if (blockIdx.x >= a.s[3].firstBlock) k = 3;
    const AddSpan sp = a.s[k];
    const size_t i = (size_t)(blockIdx.x - sp.firstBlock) * blockDim.x + threadIdx.x;
    if (sp.wide) {
        const size_t n4 = sp.floats / 4;
        if (i < n4) {
            float4 x = reinterpret_cast<float4*>(sp.dst)[i];
            const float4 y = reinterpret_cast<const float4*>(sp.src)[i];
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
            reinterpret_cast<float4*>(sp.dst)[i] = x;
        } else {
            const size_t t = n4 * 4 + (i - n4);          // the (at most three) floats behind the last whole unit
            if (t < sp.floats) sp.dst[t] += sp.src[t];
        }
    } else if (i < sp.floats) {
        sp.dst[i] += sp.src[i];
    }
}

// second stage of ReduceMetricsJob for the asynchronous form: the per-block partials -> one RtowMetrics record (integer sums and min / max: any order gives
// the host-side fold's result)
__global__ void __launch_bounds__(256) fold_metrics_kernel(const MetricsPartial* __restrict__ parts, int count, RtowMetrics* __restrict__ out)
{
    MetricsPartial mine;
    mine.rays = 0; mine.samples = 0; mine.minW = __builtin_inff(); mine.maxW = -__builtin_inff(); mine.minS = __builtin_inff(); mine.maxS = -__builtin_inff();
    for (int i = (int)threadIdx.x; i < count; i += (int)blockDim.x) {
        const MetricsPartial p = parts[i];
        mine.rays += p.rays; mine.samples += p.samples;
        mine.minW = um_min(mine.minW, p.minW); mine.maxW = um_max(mine.maxW, p.maxW);
        mine.minS = um_min(mine.minS, p.minS); mine.maxS = um_max(mine.maxS, p.maxS);
    }
    __shared__ MetricsPartial sh[256];
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            MetricsPartial a = sh[threadIdx.x];
            const MetricsPartial b = sh[threadIdx.x + s];
            a.rays += b.rays; a.samples += b.samples;
            a.minW = um_min(a.minW, b.minW); a.maxW = um_max(a.maxW, b.maxW);
            a.minS = um_min(a.minS, b.minS); a.maxS = um_max(a.maxS, b.maxS);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const MetricsPartial t = sh[0];
        RtowMetrics m;
        m.totalRayCount = (int32_t)(uint32_t)(uint64_t)t.rays;         // the reference accumulates in int32 (wraps)
        m.totalSamples = (int32_t)(uint32_t)(uint64_t)t.samples;
        m.sampleCountWeightExtrema = RtowFloat2{t.minW, t.maxW};
        m.sampleCountExtrema[0] = (int32_t)t.minS;
        m.sampleCountExtrema[1] = (int32_t)t.maxS;
        m.totalRayCount64 = t.rays;
        m.totalSamples64 = t.samples;
        *out = m;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Same-XCD hand-over litmus (run once per context, rtowCreateContext).  A chained launch hands a pixel chunk's accumulators from batch b to batch b + 1 inside
// the running kernel with PLAIN stores + s_waitcnt vmcnt(0) on the writing side and sc1 loads on the reading side, both on the XCD that owns the chunk
// (rtow_sample_kernel.hip.h, "Chained batches").  That those loads see those stores is measured behaviour of this part (profiles/calib/xcd_affinity_probe.hip),
// not an architectural guarantee - a partition mode, firmware or memory-type change could break it silently.  So the context measures it on the device it
// runs on: workgroups pair up by the XCD they land on (s_getreg XCC_ID), the writer of a pair fills 1 KiB with the round number exactly like the sample kernel
// stores a pixel, publishes a flag; the reader - which read the same lines a round earlier, so its L1 and the L2 hold them - reads them back exactly like the
// sample kernel loads a pixel.  Any stale dword, or a pair that does not finish, and the context runs chains batch by batch instead (rtow_api.hip).
// Every wait is bounded: a litmus must not be able to hang the device.
// state: [0] registered workgroups, [1..16] workgroups per XCD, [17] stale dwords, [18] timeouts, [19] pairs that ran
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) xcd_coherence_litmus_kernel(unsigned* state, unsigned* flag, unsigned* ack, unsigned* data, int rounds, unsigned pairsPerXcd)
{
    constexpr int kSpin = 1 << 18;
    __shared__ unsigned shRole, shXcd, shGo;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 15u;
        shXcd = x;
        shRole = atomicAdd(&state[1 + x], 1u);
        __threadfence();
        atomicAdd(&state[0], 1u);
        int spins = 0;
        while (__hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < kSpin) __builtin_amdgcn_s_sleep(4);
        const unsigned total = __hip_atomic_load(&state[1 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned go = spins < kSpin ? 1u : 0u;
        if (!go) atomicAdd(&state[18], 1u);
        if ((shRole | 1u) >= total || (shRole >> 1) >= pairsPerXcd) go = 0u;           // no partner on this XCD (odd count) or beyond the buffers
        shGo = go;
    }
    __syncthreads();
    if (!shGo) return;
    const bool writer = (shRole & 1u) == 0u;
    const unsigned pair = shXcd * pairsPerXcd + (shRole >> 1);
    unsigned* d = data + (size_t)pair * 256u + threadIdx.x * 4u;
    unsigned stale = 0;
    bool timedOut = false;
    for (int r = 1; r <= rounds && !timedOut; r++) {
        if (writer) {
            d[0] = (unsigned)r; d[1] = (unsigned)r; d[2] = (unsigned)r; d[3] = (unsigned)r;              // plain (write-back) stores, like a pixel's accumulators
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");                                          // coherent_flush()
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(flag + pair * 32u, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ack + pair * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r && ++spins < kSpin) __builtin_amdgcn_s_sleep(2);
                shGo = spins < kSpin ? 1u : 0u;
            }
            __syncthreads();
            timedOut = shGo == 0u;
        } else {
            if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag + pair * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r && ++spins < kSpin) __builtin_amdgcn_s_sleep(2);
                shGo = spins < kSpin ? 1u : 0u;
            }
            __syncthreads();
            timedOut = shGo == 0u;
            if (!timedOut) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 q;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(d) : "memory");      // coherent_load4()
                stale += (q.x != (unsigned)r) + (q.y != (unsigned)r) + (q.z != (unsigned)r) + (q.w != (unsigned)r);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(ack + pair * 32u, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
    if (timedOut && threadIdx.x == 0) atomicAdd(&state[18], 1u);
    if (!writer) {
        if (stale) atomicAdd(&state[17], stale);
        if (threadIdx.x == 0 && !timedOut) atomicAdd(&state[19], 1u);
    }
}

// the tie watch's bitmap -> the fix-up launch's list (rtow_kernels.h: SampleKernelArgs.tieRedo); almost always all zero
__global__ void __launch_bounds__(256) collect_tied_pixels_kernel(const unsigned* __restrict__ bits, unsigned words, unsigned* __restrict__ redo, unsigned capacity, unsigned batches,
                                                                  uint32_t* overflowFlag, unsigned busyAt)
{
    const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    unsigned v = bits[w];
    while (v) {
        const unsigned b = (unsigned)__builtin_ctz(v);
        v &= v - 1u;
        const unsigned k = atomicAdd(redo, batches);
        if (k + batches > busyAt) overflowFlag[1] = 1u;           // correct, but many: the host sends an all-triangle scene that ties this often to its exact-tie kernels (rtow_api.hip kTieWatchBusy)
        for (unsigned q = 0; q < batches; q++) {
            if (k + q < capacity) redo[4u + k + q] = (q << 27) | (w * 32u + b);
            else *overflowFlag = 1u;                              // more tied pixel-batches than the list holds: RTOW_ERROR_CAPACITY on the host side
        }
    }
}

} // namespace

hipError_t launchCollectTiedPixels(const unsigned* tieBits, unsigned words, unsigned* tieRedo, unsigned capacity, unsigned batches, uint32_t* overflowFlag, unsigned busyAt, hipStream_t stream)
{
    hipLaunchKernelGGL(collect_tied_pixels_kernel, dim3((words + 255u) / 256u), dim3(256), 0, stream, tieBits, words, tieRedo, capacity, batches, overflowFlag, busyAt);
    return hipGetLastError();
}

hipError_t runXcdCoherenceLitmus(int cuCount, hipStream_t stream, unsigned* outPairs, unsigned* outStale, unsigned* outTimeouts)
{
    const unsigned grid = (unsigned)(cuCount * 2 < 64 ? 64 : (cuCount * 2 > 1024 ? 1024 : cuCount * 2));
    const unsigned pairsPerXcd = grid / 2;
    const int rounds = 48;
    unsigned *state = nullptr, *flag = nullptr, *ack = nullptr, *data = nullptr;
    const size_t pairs = (size_t)kMaxXcds * pairsPerXcd;
    hipError_t e = hipMalloc(&state, 32 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&flag, pairs * 32 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&ack, pairs * 32 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&data, pairs * 256 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(state, 0, 32 * sizeof(unsigned), stream);
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, pairs * 32 * sizeof(unsigned), stream);
    if (e == hipSuccess) e = hipMemsetAsync(ack, 0, pairs * 32 * sizeof(unsigned), stream);
    if (e == hipSuccess) e = hipMemsetAsync(data, 0, pairs * 256 * sizeof(unsigned), stream);
    unsigned host[32] = {0};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(xcd_coherence_litmus_kernel, dim3(grid), dim3(64), 0, stream, state, flag, ack, data, rounds, pairsPerXcd);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, state, sizeof(host), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (state) (void)hipFree(state);
    if (flag) (void)hipFree(flag);
    if (ack) (void)hipFree(ack);
    if (data) (void)hipFree(data);
    *outPairs = host[19]; *outStale = host[17]; *outTimeouts = host[18];
    return e;
}

hipError_t launchCombineFinalize(const RtowCombineParams& p, const float* inColor, const float* inNormal, const float* inAlbedo,
                                 uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, const float* thresholds, hipStream_t stream)
{
    const int n = p.width * p.height;
    const int blocks = n < 256 * RTOW_FINALIZE_BLOCKS ? (n + 255) / 256 : RTOW_FINALIZE_BLOCKS;
    hipLaunchKernelGGL(combine_finalize_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, p, reinterpret_cast<const float4*>(inColor), inNormal, inAlbedo,
                       reinterpret_cast<uchar4*>(outColor), reinterpret_cast<uchar4*>(outNormal), reinterpret_cast<uchar4*>(outAlbedo), thresholds);
    return hipGetLastError();
}

hipError_t launchAddAccum(size_t pixels, float* const dst[4], const float* const src[4], hipStream_t stream)
{
    static const size_t comps[4] = {4, 3, 3, 1};
    AddSpans a;
    unsigned blocks = 0;
    for (int k = 0; k < 4; k++) {
        const size_t floats = pixels * comps[k];
        const bool wide = ((reinterpret_cast<uintptr_t>(dst[k]) | reinterpret_cast<uintptr_t>(src[k])) & 15u) == 0;
        const size_t units = wide ? floats / 4 + (floats & 3) : floats;
        a.s[k] = AddSpan{dst[k], src[k], floats, blocks, wide ? 1u : 0u};
        blocks += (unsigned)((units + 255) / 256);
    }
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(add_accum_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launchFoldMetrics(const MetricsPartial* partials, RtowMetrics* out, hipStream_t stream)
{
    hipLaunchKernelGGL(fold_metrics_kernel, dim3(1), dim3(256), 0, stream, partials, kMetricsBlocks, out);
    return hipGetLastError();
}

hipError_t launchSampleBatch(const SampleKernelArgs& args, int numBlocks, hipStream_t stream)
{
    const bool allLds = args.ldsSceneBytes == args.layout.totalBytes;
    switch (args.layout.sceneKind) {
        case SCENE_KIND_SPHERES: return args.layout.exactTies ? launchSampleSpheresTies(args, numBlocks, stream, allLds) : launchSampleSpheres(args, numBlocks, stream, allLds);
        case SCENE_KIND_SPHERES_MOTION: return args.layout.exactTies ? launchSampleSpheresMotionTies(args, numBlocks, stream, allLds) : launchSampleSpheresMotion(args, numBlocks, stream, allLds);
        case SCENE_KIND_VOLUMES: return launchSampleVolumes(args, numBlocks, stream, allLds);
        case SCENE_KIND_TEXTURED: return args.layout.exactTies ? launchSampleTexturedTies(args, numBlocks, stream, allLds) : launchSampleTextured(args, numBlocks, stream, allLds);
        case SCENE_KIND_VOLUMES_TEXTURED: return launchSampleVolumesTextured(args, numBlocks, stream, allLds);
        case SCENE_KIND_TRIANGLES_TEXTURED: return args.layout.exactTies ? launchSampleTrianglesTexturedTies(args, numBlocks, stream, allLds) : launchSampleTrianglesTextured(args, numBlocks, stream, allLds);
        case SCENE_KIND_TRIANGLES: return args.layout.exactTies ? launchSampleTrianglesTies(args, numBlocks, stream, allLds) : launchSampleTriangles(args, numBlocks, stream, allLds);
        default: return args.layout.exactTies ? launchSampleGeneralTies(args, numBlocks, stream, allLds) : launchSampleGeneral(args, numBlocks, stream, allLds);
    }
}

// ------------------------------------------------------------------------------------------------------------
// rtowGatherRowsDevice: rows first, first + step, ... of a full-frame buffer <-> one contiguous block (what travels over xGMI).
// HBM bound, 4 B read + 4 B written per float; at most 11 floats per owned pixel per batch.
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) copy_rows_kernel(T* frame, T* packed, unsigned rowUnits, unsigned rows, unsigned first, unsigned step, int toFrame)
{
    // one row per blockIdx.y slice, grid-stride inside the row: no division per element
    for (unsigned k = blockIdx.y; k < rows; k += gridDim.y) {
        T* f = frame + ((size_t)first + (size_t)k * step) * rowUnits;
        T* q = packed + (size_t)k * rowUnits;
        for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < rowUnits; j += gridDim.x * blockDim.x) {
            if (toFrame) f[j] = q[j];
            else q[j] = f[j];
        }
    }
}

hipError_t launchCopyRows(float* frame, float* packed, unsigned rowFloats, unsigned rows, unsigned first, unsigned step, bool toFrame, hipStream_t stream)
{
    if ((size_t)rows * rowFloats == 0) return hipSuccess;
    // rows travel as 16-byte units when every row starts on a 16-byte boundary in both buffers
    const bool wide = (rowFloats & 3u) == 0u && ((reinterpret_cast<uintptr_t>(frame) | reinterpret_cast<uintptr_t>(packed)) & 15u) == 0u;
    const unsigned units = wide ? rowFloats / 4u : rowFloats;
    const unsigned bx = units < 256u * 8u ? (units + 255u) / 256u : 8u;
    const unsigned by = rows < 4096u ? rows : 4096u;
    if (wide) hipLaunchKernelGGL(copy_rows_kernel<float4>, dim3(bx, by), dim3(256), 0, stream, reinterpret_cast<float4*>(frame), reinterpret_cast<float4*>(packed), units, rows, first, step, toFrame ? 1 : 0);
    else hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(bx, by), dim3(256), 0, stream, frame, packed, units, rows, first, step, toFrame ? 1 : 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// rtowExchangeAccumDevice: accum[row] += src_0[row]; accum[row] += src_1[row]; ... for the rows first, first + step, ... of a full-frame buffer,
// in group order with one rounding per addition - what `groups` successive add passes compute - in ONE pass that reads and writes accum once.
// Source g is the rank's own partial sum (frame layout, in place) for g == own, else region g of the receive block (packed rows).
// ------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T add_units(T a, T b);
template <> __device__ __forceinline__ float add_units<float>(float a, float b) { return a + b; }
template <> __device__ __forceinline__ float4 add_units<float4>(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <typename T>
__global__ void __launch_bounds__(256) fold_rows_kernel(T* __restrict__ accum, const T* __restrict__ ownPartial, const T* __restrict__ recv, size_t regionUnits, unsigned rowUnits,
                                                        unsigned rows, unsigned first, unsigned step, unsigned groups, unsigned own)
{
    for (unsigned k = blockIdx.y; k < rows; k += gridDim.y) {
        const size_t frameRow = ((size_t)first + (size_t)k * step) * rowUnits, packedRow = (size_t)k * rowUnits;
        for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < rowUnits; j += gridDim.x * blockDim.x) {
            T a = accum[frameRow + j];
            for (unsigned g = 0; g < groups; g++) a = add_units<T>(a, g == own ? ownPartial[frameRow + j] : recv[(size_t)g * regionUnits + packedRow + j]);
            accum[frameRow + j] = a;
        }
    }
}

hipError_t launchFoldRows(float* accum, const float* ownPartial, const float* recv, size_t regionFloats, unsigned rowFloats, unsigned rows, unsigned first, unsigned step,
                          unsigned groups, unsigned own, hipStream_t stream)
{
    if ((size_t)rows * rowFloats == 0 || groups == 0) return hipSuccess;
    const bool wide = (rowFloats & 3u) == 0u && (regionFloats & 3u) == 0u &&
                      ((reinterpret_cast<uintptr_t>(accum) | reinterpret_cast<uintptr_t>(ownPartial) | reinterpret_cast<uintptr_t>(recv)) & 15u) == 0u;
    const unsigned units = wide ? rowFloats / 4u : rowFloats;
    const unsigned bx = units < 256u * 8u ? (units + 255u) / 256u : 8u;
    const unsigned by = rows < 4096u ? rows : 4096u;
    if (wide) hipLaunchKernelGGL(fold_rows_kernel<float4>, dim3(bx, by), dim3(256), 0, stream, reinterpret_cast<float4*>(accum), reinterpret_cast<const float4*>(ownPartial),
                                 reinterpret_cast<const float4*>(recv), regionFloats / 4u, units, rows, first, step, groups, own);
    else hipLaunchKernelGGL(fold_rows_kernel<float>, dim3(bx, by), dim3(256), 0, stream, accum, ownPartial, recv, regionFloats, units, rows, first, step, groups, own);
    return hipGetLastError();
}

hipError_t launchFoldUnitRecords(const SampleKernelArgs& args, hipStream_t stream)
{
    const unsigned pixels = args.totalWork / args.groupsPerPixel;
    hipLaunchKernelGGL(fold_unit_records_kernel, dim3((pixels + 255u) / 256u), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launchPrimaryCandidates(const SampleKernelArgs& args, uint2* out, hipStream_t stream)
{
    hipLaunchKernelGGL(primary_candidates_kernel, dim3((args.totalWork + 255u) / 256u), dim3(256), 0, stream, args, out);
    return hipGetLastError();
}

hipError_t launchBuildChunkOrder(const unsigned short* pixelCost, unsigned* cost, unsigned chunkCount, unsigned* order, int byMax, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_chunk_cost_kernel, dim3((chunkCount + 3u) / 4u), dim3(256), 0, stream, pixelCost, chunkCount, cost);
    if (chunkCount > kOrderSingleBlockChunks) {
        unsigned* scratch = cost + 2u * (size_t)chunkCount;            // kChunkOrderScratchWords behind the two cost columns (rtow_api.hip allocates them)
        const unsigned blocks = (chunkCount + kOrderChunksPerBlock - 1u) / kOrderChunksPerBlock;        // order_hist / order_scatter: one workgroup per 2 048 chunks
        hipError_t e = hipMemsetAsync(scratch, 0, (size_t)kChunkOrderScratchWords * sizeof(unsigned), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(order_max_kernel, dim3(blocks), dim3(256), 0, stream, cost, chunkCount, byMax, scratch);
        hipLaunchKernelGGL(order_hist_kernel, dim3(blocks), dim3(256), 0, stream, cost, chunkCount, byMax, scratch);
        hipLaunchKernelGGL(order_prefix_kernel, dim3(1), dim3(1024), 0, stream, scratch);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, cost, chunkCount, byMax, scratch, order);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(build_chunk_order_kernel, dim3(1), dim3(1024), 0, stream, cost, chunkCount, order, byMax);
    return hipGetLastError();
}

hipError_t launchInitTicketMap(unsigned* ticketMap, unsigned count, hipStream_t stream)
{
    if (count == 0u) return hipSuccess;
    hipLaunchKernelGGL(init_ticket_map_kernel, dim3((count + 1023u) / 1024u), dim3(1024), 0, stream, ticketMap, count);
    return hipGetLastError();
}
hipError_t launchRegroupTickets(unsigned short* pixelCost, unsigned* ticketMap, unsigned tilesPerRow, unsigned tileRows, unsigned side, const unsigned classes[3], hipStream_t stream)
{
    if (tilesPerRow == 0u || tileRows == 0u) return hipSuccess;
    if (side == 1u) {                                                            // the tiles as they are, their tickets most expensive first
        const unsigned tiles = tilesPerRow * tileRows;
        hipLaunchKernelGGL(order_tile_tickets_kernel, dim3((tiles + 3u) / 4u), dim3(256), 0, stream, pixelCost, ticketMap, tiles, classes[0]);      // classes[0]: levels (0 = by the ray count itself)
        return hipGetLastError();
    }
    if (side < 2u || side > kRegroupMaxSide) return hipErrorInvalidValue;
    const unsigned blocks = ((tilesPerRow + side - 1u) / side) * ((tileRows + side - 1u) / side);
    unsigned N = 64u;
    while (N < side * side * 64u) N <<= 1;
    hipLaunchKernelGGL(regroup_tickets_kernel, dim3(blocks), dim3(1024), (size_t)N * sizeof(unsigned long long), stream, pixelCost, ticketMap, tilesPerRow, tileRows, side, classes[0], classes[1], classes[2]);
    return hipGetLastError();
}
hipError_t launchPrepareEntities(uint8_t* blob, const SceneLayout& layout, hipStream_t stream)
{
    if (layout.sceneKind < SCENE_KIND_GENERAL) return hipSuccess;
    const unsigned blocks = (layout.sphereCount + 127u) / 128u;
    hipLaunchKernelGGL(prepare_entities_kernel, dim3(blocks ? blocks : 1), dim3(128), 0, stream, blob, layout);
    return hipGetLastError();
}

hipError_t launchPrepareMaterials(uint8_t* blob, const SceneLayout& layout, hipStream_t stream)
{
    const unsigned blocks = (layout.materialCount + 127u) / 128u;
    hipLaunchKernelGGL(prepare_materials_kernel, dim3(blocks ? blocks : 1), dim3(128), 0, stream, blob, layout);
    return hipGetLastError();
}

hipError_t launchCombine(const RtowCombineParams& p, const float* inColor, const float* inNormal, const float* inAlbedo,
                         float* outColor, float* outNormal, float* outAlbedo, hipStream_t stream)
{
    const int n = p.width * p.height;
    const int blocks = n < 256 * RTOW_COMBINE_BLOCKS ? (n + 255) / 256 : RTOW_COMBINE_BLOCKS;
    hipLaunchKernelGGL(combine_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, p, reinterpret_cast<const float4*>(inColor), inNormal, inAlbedo,
                       outColor, outNormal, outAlbedo);
    return hipGetLastError();
}

static_assert(kByteThresholdTableBytes == kByteThresholdFloats * sizeof(float), "rtow_kernels.h and rtow_finalize.hip.h disagree on the table");

hipError_t launchBuildByteThresholds(float* thresholds, hipStream_t stream)
{
    hipLaunchKernelGGL(build_byte_thresholds_kernel, dim3(1), dim3(256), 0, stream, thresholds);
    return hipGetLastError();
}

hipError_t launchFinalize(int pixelCount, const float* inColor, const float* inNormal, const float* inAlbedo,
                          uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, const float* thresholds, hipStream_t stream)
{
    const int blocks = pixelCount < 256 * RTOW_FINALIZE_BLOCKS ? (pixelCount + 255) / 256 : RTOW_FINALIZE_BLOCKS;
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, pixelCount, inColor, inNormal, inAlbedo,
                       reinterpret_cast<uchar4*>(outColor), reinterpret_cast<uchar4*>(outNormal), reinterpret_cast<uchar4*>(outAlbedo), thresholds);
    return hipGetLastError();
}

hipError_t launchReduceMetrics(int pixelCount, const uint8_t* diagnostics, int stride, const float* color, const float* scw,
                               MetricsPartial* partials, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_metrics_kernel, dim3(kMetricsBlocks), dim3(256), 0, stream, pixelCount, diagnostics, stride,
                       reinterpret_cast<const float4*>(color), scw, partials);
    return hipGetLastError();
}

} // namespace rtow
