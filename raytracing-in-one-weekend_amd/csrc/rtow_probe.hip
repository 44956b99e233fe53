// rtow_probe.hip - rtowProbeNearestHit: one ray against the resident scene, walked on the HOST.
//
// Replaces the host's HitWorld (UNITY/Raytracer.cs:1353: BvhRoot->Hit(r, 0, +inf, out hitRec) -> the recursive HitTests.Hit(BvhNode), RT/HitTests.cs:152-196), which
// ScheduleSample calls with the camera's centre ray before every batch to set the focus distance (UNITY/Raytracer.cs:608-609) - the one reason a host that has handed
// its scene to this library would still keep its own serial RebuildBvh alive (UNITY/Raytracer.cs:1306-1351).
//
// Why not a launch: ScheduleSample runs while the previous batch is still tracing (two in flight, UNITY/Raytracer.cs:586-596), and the sample kernel owns every CU's
// register file and LDS until it ends - a probe kernel would start when that batch is over, and the host, waiting for it, could not queue the next batch in time.  One ray
// is microseconds of CPU work on the tree this library already built; the image of the scene (CompiledScene.blob, the very bytes the kernels read, with the derived
// transforms copied back after the device computed them) stays on the host for it.
//
// What the reference's recursion computes: the smallest Entity.Hit distance (tMin 0, tMax +inf) over the entities of the leaves it reaches, and it reaches a leaf iff the
// ray passes the box of every node above it under AxisAlignedBoundingBox.Hit (RT/HitTests.cs:9-21).  A box that encloses another passes whenever the inner one does
// (subtraction, multiplication, min and max are monotone in binary32), so that set is "the entities whose own box the ray passes" - what the leaf children of this
// library's tree carry (the reference's own entity boxes under the reference's own slab test; the leaf's box where the host forced a leaf at MaxBvhDepth).  The walk
// below visits them with the sample kernel's own hit tests (sphere_at / sphere_hit / general_hit of rtow_sample_kernel.hip.h, compiled for the host: the same
// expressions in the same order, IEEE division and square root, no contraction), pruning inner boxes by the best distance so far with the kernel's 2^-12 of slack.
// Hits at bit-identical distance: the entity that comes first in the reference tree's leaf order (what the sample path shades); HitWorld's own recursion prefers its right
// subtree on such a tie - the host only reads the distance.
#include "rtow_sample_kernel.hip.h"

#include <vector>

namespace rtow {

namespace {

// v_min_f32 / v_max_f32 in IEEE mode as the kernel's slab test uses them: a NaN operand yields the other operand
inline float hmin(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
inline float hmax(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }

template <int BASE>      // SCENE_KIND_SPHERES, SCENE_KIND_SPHERES_MOTION, or SCENE_KIND_GENERAL for every kind that keeps GpuPrim records
void walk(const uint8_t* blob, const SceneLayout& L, V3 ro, V3 rd, float time, float& bestT, int& bestPrim)
{
    constexpr bool GENERAL = BASE >= SCENE_KIND_GENERAL;
    constexpr bool HAS_MOTION = BASE == SCENE_KIND_SPHERES_MOTION;
    SceneRefs sc;
    sc.lds = nullptr;
    sc.glob = blob;
    sc.ldsNodeCount = 0;
    float rtime = time;
    if (HAS_MOTION) { if (L.commonTimeRange) rtime = um_max(0.0f, um_min(1.0f, (rtime - L.commonT0) / (L.commonT1 - L.commonT0))); }     // what sphere_at expects (the kernel's REGEN does the same)
    const V3 inv = v3(exact_rcp_nan_to_inf(rd.x), exact_rcp_nan_to_inf(rd.y), exact_rcp_nan_to_inf(rd.z));
    const float a = dot(rd, rd);
    const unsigned* rank = reinterpret_cast<const unsigned*>(blob + L.rankOffset);
    const bool twoChildren = L.sphereCount > 1u;
    float best = __builtin_inff();
    int prim = -1;
    std::vector<int> stack((size_t)L.bvhDepth + 2u);      // one entry per inner level of the tree that was built (never more pending far children than that)
    int sp = 0, cur = 0;
    while (cur >= 0) {
        float4 q0, q1, q2;
        int c0, c1;
        load_node<false>(sc, L, cur, q0, q1, q2, c0, c1);
        const float bestPrune = best * 1.000244140625f;
        int next[2];
        float entry[2];
        int inner = 0;
        for (int side = 0; side < 2; side++) {
            if (side == 1 && !twoChildren) break;
            const int child = side ? c1 : c0;
            const float lox = side ? q0.y : q0.x, loy = side ? q0.w : q0.z, loz = side ? q1.y : q1.x;
            const float hix = side ? q1.w : q1.z, hiy = side ? q2.y : q2.x, hiz = side ? q2.w : q2.z;
            const float tlx = (lox - ro.x) * inv.x, thx = (hix - ro.x) * inv.x;
            const float tly = (loy - ro.y) * inv.y, thy = (hiy - ro.y) * inv.y;
            const float tlz = (loz - ro.z) * inv.z, thz = (hiz - ro.z) * inv.z;
            const float tmin = hmax(hmax(hmin(tlx, thx), hmin(tly, thy)), hmax(hmin(tlz, thz), 0.0f));
            const float tfar = hmin(hmin(hmax(tlx, thx), hmax(tly, thy)), hmax(tlz, thz));
            if (child >= 0) {
                if (tmin <= hmin(tfar, bestPrune)) { next[inner] = child; entry[inner] = tmin; inner++; }      // padded inner box: conservative, pruned by the nearest hit so far
                continue;
            }
            if (!(tmin < tfar)) continue;                                      // AxisAlignedBoundingBox.Hit on the entity's own box (RT/HitTests.cs:15-20)
            const int i = ~child;
            float t;
            bool hit;
            if (GENERAL) {
                const unsigned type = *reinterpret_cast<const unsigned*>(blob + L.matIndexOffset + (uint32_t)i * 4u) >> kPrimTypeShift;
                V3 nl; float4 rq;
                hit = general_hit<false>(sc, L, i, type, ro, rd, rtime, 0.0f, t, nl, rq);
            } else {
                V3 c; float r;
                sphere_at<false, HAS_MOTION>(sc, L, i, rtime, c, r);
                hit = sphere_hit(sub(ro, c), rd, a, r, t);
            }
            if (hit && (t < best || (t == best && prim >= 0 && rank[i] < rank[prim]))) { best = t; prim = i; }
        }
        if (inner == 2) {
            const int far = entry[1] < entry[0] ? 0 : 1;                       // near child first
            if ((size_t)sp == stack.size()) stack.resize(stack.size() * 2);      // (cannot happen for a tree within its own depth bound; never drop a subtree silently)
            stack[sp++] = next[far];
            cur = next[1 - far];
        } else if (inner == 1) {
            cur = next[0];
        } else {
            cur = sp > 0 ? stack[--sp] : -1;
        }
    }
    bestT = best;
    bestPrim = prim;
}

} // namespace

// blob: the HOST image of the scene, derived entity transforms included (rtowUploadScene copies them back).  entityOfPrim: CompiledScene.entityOfPrim (all-triangle scenes
// number their primitives in leaf order) or null.  Returns false on a miss.
bool probeNearestHitHost(const uint8_t* blob, const SceneLayout& L, const int32_t* entityOfPrim, const float origin[3], const float direction[3], float time, float* distance, int* entity)
{
    const V3 ro = v3(origin[0], origin[1], origin[2]), rd = v3(direction[0], direction[1], direction[2]);
    float t = __builtin_inff();
    int prim = -1;
    if (L.sceneKind == SCENE_KIND_SPHERES) walk<SCENE_KIND_SPHERES>(blob, L, ro, rd, time, t, prim);
    else if (L.sceneKind == SCENE_KIND_SPHERES_MOTION) walk<SCENE_KIND_SPHERES_MOTION>(blob, L, ro, rd, time, t, prim);
    else walk<SCENE_KIND_GENERAL>(blob, L, ro, rd, time, t, prim);
    *distance = t;
    *entity = prim >= 0 && entityOfPrim ? entityOfPrim[prim] : prim;
    return prim >= 0;
}

} // namespace rtow
