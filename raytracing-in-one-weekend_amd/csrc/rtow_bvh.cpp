// rtow_bvh.cpp - native scene compiler: RtowSceneDesc -> flat LDS-sized GPU layout (rtow_scene.h).
//
// Replaces, for this path, RebuildEntityBuffers' Entity/Material packing (UNITY/Raytracer.cs:1185-1304) and
// RebuildBvh (UNITY/Raytracer.cs:1306-1351 -> UNITY/BvhNodeData.cs:122-213 -> JOBS/BuildRuntimeBvhJob.cs:20-39).
// The reference builder (sort on the largest axis, half-extent split) exists to feed a CPU walk that collects
// every overlapped leaf; here the tree feeds a 64-wide closest-hit traversal whose cost is LDS traffic and
// divergence, so the builder is a full-sweep SAH with a hard depth bound (the traversal stack lives in LDS and is
// sized by it).  The choice of tree is results-neutral: the nearest hit of a ray does not depend on it.
#include "rtow_bvh.h"
#include "rtow_reforder.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <queue>

namespace rtow {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    double area() const
    {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct TmpNode {
    Box box[2];
    int child[2]; // >= 0: TmpNode index, < 0: ~primitive
};

// Ranges above this many primitives are split with a binned SAH (32 bins per axis, O(n) per node) instead of the full sweep (three sorts per node):
// a million-triangle mesh builds in seconds.  Below it the sweep stands, so every scene of up to 32 768 entities gets the tree it always got.
constexpr int kBinnedAbove = 32768;
constexpr int kBins = 32;

struct Builder {
    std::vector<Box> primBox;
    std::vector<float> centroid[3];
    std::vector<TmpNode> nodes;
    std::vector<int> order;          // sweep scratch, grown once to the root's size
    std::vector<double> rightArea;
    int maxDepthSeen = 0;

    // returns child code; box receives the subtree bounds. depthLeft = inner-node levels still allowed.
    int build(std::vector<int>& idx, int begin, int end, int depthLeft, int depth, Box* outBox)
    {
        const int n = end - begin;
        if (n == 1) {
            *outBox = primBox[idx[begin]];
            return ~idx[begin];
        }
        maxDepthSeen = std::max(maxDepthSeen, depth + 1);
        const long long cap = depthLeft - 1 >= 30 ? (1LL << 30) : (1LL << std::max(depthLeft - 1, 0));

        if (n > kBinnedAbove) return buildBinned(idx, begin, end, depthLeft, depth, cap, outBox);

        double bestCost = DBL_MAX;
        int bestAxis = -1, bestSplit = -1;
        // scratch is shared by the whole recursion (a node's sweep is over before its children start); only the winning (axis, split)
        // is remembered during the sweep and the range is sorted once more along that axis afterwards - copying the sorted order on
        // every cost improvement was O(n) copies of O(n) ints at the top of a 65 535-entity tree
        if ((int)order.size() < n) { order.resize(n); rightArea.resize(n); }
        auto sortAlong = [&](int axis, int* first, int* last) {
            const std::vector<float>& c = centroid[axis];
            std::sort(first, last, [&c](int a, int b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
        };
        for (int axis = 0; axis < 3; axis++) {
            std::copy(idx.begin() + begin, idx.begin() + end, order.begin());
            sortAlong(axis, order.data(), order.data() + n);
            Box acc;
            acc.reset();
            for (int i = n - 1; i >= 1; i--) { acc.grow(primBox[order[i]]); rightArea[i] = acc.area(); }
            acc.reset();
            for (int i = 1; i < n; i++) {
                acc.grow(primBox[order[i - 1]]);
                if (i > cap || (n - i) > cap) continue; // keep both subtrees buildable within the depth bound
                const double cost = acc.area() * i + rightArea[i] * (n - i);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = i; }
            }
        }
        if (bestAxis < 0) bestSplit = n / 2; // cannot happen while n <= 2^depthLeft; defensive median split in the current order
        else sortAlong(bestAxis, idx.data() + begin, idx.data() + end);   // the comparator is a total order: the same sequence the sweep saw

        const int self = (int)nodes.size();
        nodes.push_back(TmpNode{});
        Box b0, b1;
        const int c0 = build(idx, begin, begin + bestSplit, depthLeft - 1, depth + 1, &b0);
        const int c1 = build(idx, begin + bestSplit, end, depthLeft - 1, depth + 1, &b1);
        nodes[self].box[0] = b0; nodes[self].box[1] = b1;
        nodes[self].child[0] = c0; nodes[self].child[1] = c1;
        *outBox = b0;
        outBox->grow(b1);
        return self;
    }

    // binned SAH step for a large range; partitions idx[begin, end) in place and recurses through build()
    int buildBinned(std::vector<int>& idx, int begin, int end, int depthLeft, int depth, long long cap, Box* outBox)
    {
        const int n = end - begin;
        float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = begin; i < end; i++)
            for (int a = 0; a < 3; a++) { const float c = centroid[a][idx[i]]; clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
        double bestCost = DBL_MAX;
        int bestAxis = -1, bestBin = -1;
        for (int axis = 0; axis < 3; axis++) {
            const float ext = chi[axis] - clo[axis];
            if (!(ext > 0)) continue;
            const float scale = (float)kBins / ext;
            Box bb[kBins];
            int cnt[kBins] = {0};
            for (int b = 0; b < kBins; b++) bb[b].reset();
            for (int i = begin; i < end; i++) {
                int b = (int)((centroid[axis][idx[i]] - clo[axis]) * scale);
                b = b < 0 ? 0 : b >= kBins ? kBins - 1 : b;
                cnt[b]++;
                bb[b].grow(primBox[idx[i]]);
            }
            double rArea[kBins];
            int rCnt[kBins];
            Box acc;
            acc.reset();
            int c = 0;
            for (int b = kBins - 1; b >= 1; b--) { if (cnt[b]) acc.grow(bb[b]); c += cnt[b]; rArea[b] = c ? acc.area() : 0.0; rCnt[b] = c; }
            acc.reset();
            c = 0;
            for (int b = 1; b < kBins; b++) {                       // split between bin b - 1 and bin b
                if (cnt[b - 1]) acc.grow(bb[b - 1]);
                c += cnt[b - 1];
                if (c == 0 || rCnt[b] == 0 || c > cap || rCnt[b] > cap) continue;
                const double cost = acc.area() * c + rArea[b] * rCnt[b];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        int mid;
        if (bestAxis >= 0) {
            const float scale = (float)kBins / (chi[bestAxis] - clo[bestAxis]);
            const std::vector<float>& cc = centroid[bestAxis];
            const float lo = clo[bestAxis];
            const int bin = bestBin;
            int* first = idx.data() + begin;
            int* split = std::partition(first, idx.data() + end, [&](int p) {
                int b = (int)((cc[p] - lo) * scale);
                b = b < 0 ? 0 : b >= kBins ? kBins - 1 : b;
                return b < bin;
            });
            mid = begin + (int)(split - first);
        } else {
            // all centroids coincide on every axis, or no bin boundary keeps both sides within the depth bound: median split along the widest axis
            int axis = 0;
            for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
            const std::vector<float>& cc = centroid[axis];
            mid = begin + n / 2;
            std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&cc](int a, int b) { return cc[a] < cc[b] || (cc[a] == cc[b] && a < b); });
        }
        const int self = (int)nodes.size();
        nodes.push_back(TmpNode{});
        Box b0, b1;
        const int c0 = build(idx, begin, mid, depthLeft - 1, depth + 1, &b0);
        const int c1 = build(idx, mid, end, depthLeft - 1, depth + 1, &b1);
        nodes[self].box[0] = b0; nodes[self].box[1] = b1;
        nodes[self].child[0] = c0; nodes[self].child[1] = c1;
        *outBox = b0;
        outBox->grow(b1);
        return self;
    }
};

float texColor(const RtowTexture& t, int c)
{
    switch (t.type) {
        case RTOW_TEXTURE_CONSTANT: return c == 0 ? t.mainColor.x : c == 1 ? t.mainColor.y : t.mainColor.z; // RT/Texture.cs:55-56
        case RTOW_TEXTURE_CONSTANT_SCALAR: return t.parameter;                                             // :58-59
    }
    return 0.0f; // TextureType.None samples as 0 (:92)
}
float texScalar(const RtowTexture& t)
{
    switch (t.type) {
        case RTOW_TEXTURE_CONSTANT: return texColor(t, t.scalarValueChannel); // RT/Texture.cs:100-101
        case RTOW_TEXTURE_CONSTANT_SCALAR: return t.parameter;               // :103-104
    }
    return 0.0f;
}
bool texSupported(const RtowTexture& t)
{
    return t.type == RTOW_TEXTURE_NONE || t.type == RTOW_TEXTURE_CONSTANT || t.type == RTOW_TEXTURE_CONSTANT_SCALAR || t.type == RTOW_TEXTURE_IMAGE;
}
GpuTexture packTexture(const RtowTexture& t)
{
    GpuTexture g{};
    g.type = t.type;
    g.image = t.type == RTOW_TEXTURE_IMAGE ? t.imageIndex : -1;
    g.channel = t.scalarValueChannel;
    g.parameter = t.parameter;
    g.mainColor[0] = t.mainColor.x; g.mainColor[1] = t.mainColor.y; g.mainColor[2] = t.mainColor.z;
    return g;
}
bool almostOne(float v) { return std::fabs(1.0f - v) < 1e-6f; } // UTIL/MathExtensions.cs:24-27 with rhs = 1

uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

void setBoxes(GpuNode& g, const Box& b0, const Box& b1)
{
    g.lox[0] = b0.lo[0]; g.lox[1] = b1.lo[0]; g.loy[0] = b0.lo[1]; g.loy[1] = b1.lo[1]; g.loz[0] = b0.lo[2]; g.loz[1] = b1.lo[2];
    g.hix[0] = b0.hi[0]; g.hix[1] = b1.hi[0]; g.hiy[0] = b0.hi[1]; g.hiy[1] = b1.hi[1]; g.hiz[0] = b0.hi[2]; g.hiz[1] = b1.hi[2];
}

} // namespace

int compileScene(const RtowSceneDesc* desc, int maxDepth, CompiledScene* out, std::string* err)
{
    if (!desc || !out || !desc->entities || !desc->materials || desc->entityCount <= 0 || desc->materialCount <= 0) {
        *err = "null or empty scene description";
        return RTOW_ERROR_INVALID_VALUE;
    }
    const int n = desc->entityCount;
    // Entities: candidate / stack codes are 16 bits wide up to 65 535 entities and tree nodes and 32 bits beyond (the kernels' wide-code variants,
    // chosen at upload); hit codes keep 30 bits for the primitive, and the blob's section offsets are 32-bit.  Materials: the path history names
    // a material in 15 bits.  The reference's live host makes one entity per mesh triangle (UNITY/Raytracer.cs:1193-1198,1290-1300) - hundreds of
    // thousands for its own test scenes (UNITY/GridGenerator.cs:78-159) - and one material per renderer.
    if (n > (1 << 23) || desc->materialCount > 32767) {
        *err = "scene exceeds 8388608 entities or 32767 materials (30-bit hit codes / 32-bit blob offsets, 15-bit path-history codes)";
        return RTOW_ERROR_CAPACITY;
    }
    if (maxDepth <= 0) maxDepth = RTOW_DEFAULT_MAX_BVH_DEPTH;
    if (maxDepth > RTOW_STACK_CAPACITY) maxDepth = RTOW_STACK_CAPACITY;
    if ((1LL << std::min(maxDepth, 30)) < n) {
        *err = "maxBvhDepth too small for the entity count";
        return RTOW_ERROR_CAPACITY;
    }

    // ---- materials (RT/Material.cs:28-46, constant textures folded) ----
    std::vector<GpuMaterial> mats(desc->materialCount);
    std::vector<uint32_t> matClass(desc->materialCount);
    bool hasVolumes = false, hasImageTextures = false;
    for (int i = 0; i < desc->materialCount; i++) {
        const RtowMaterial& m = desc->materials[i];
        if (m.type != RTOW_MATERIAL_STANDARD && m.type != RTOW_MATERIAL_DIELECTRIC && m.type != RTOW_MATERIAL_PROBABILISTIC_VOLUME) {
            *err = "unknown material type";
            return RTOW_ERROR_INVALID_VALUE;
        }
        hasVolumes |= m.type == RTOW_MATERIAL_PROBABILISTIC_VOLUME;
        if (!texSupported(m.albedo) || !texSupported(m.glossiness) || !texSupported(m.emission) || !texSupported(m.metallic)) {
            *err = "texture type not built (CheckerPattern / PerlinNoise are dead code in the reference)";
            return RTOW_ERROR_UNSUPPORTED;
        }
        bool textured = false;
        for (const RtowTexture* t : {&m.albedo, &m.glossiness, &m.emission, &m.metallic}) {
            if (t->type != RTOW_TEXTURE_IMAGE) continue;
            textured = true;
            if (t->imageIndex >= desc->imageCount || (t->imageIndex >= 0 && !desc->images)) {
                *err = "imageIndex out of range";
                return RTOW_ERROR_INVALID_VALUE;
            }
            if (t->scalarValueChannel < 0 || t->scalarValueChannel > 2) {
                *err = "scalarValueChannel out of range";
                return RTOW_ERROR_INVALID_VALUE;
            }
        }
        hasImageTextures |= textured;
        GpuMaterial g{};
        for (int c = 0; c < 3; c++) { g.albedo[c] = texColor(m.albedo, c); g.emission[c] = texColor(m.emission, c); }
        g.type = m.type;
        g.metallic = texScalar(m.metallic);
        g.glossiness = texScalar(m.glossiness);
        g.parameter = (m.type == RTOW_MATERIAL_DIELECTRIC || m.type == RTOW_MATERIAL_PROBABILISTIC_VOLUME) ? m.parameter : 0.0f; // ctor stores it only for Dielectric/Volume
        g.flags = 0;
        bool spec = false;
        if (m.type == RTOW_MATERIAL_DIELECTRIC) spec = true; // RT/Material.cs:187-188
        else if (m.type == RTOW_MATERIAL_STANDARD)
            spec = m.metallic.type == RTOW_TEXTURE_CONSTANT && almostOne(m.metallic.mainColor.x) && almostOne(m.metallic.mainColor.y) &&
                    almostOne(m.metallic.mainColor.z) && m.glossiness.type == RTOW_TEXTURE_CONSTANT && almostOne(m.glossiness.mainColor.x) &&
                    almostOne(m.glossiness.mainColor.y) && almostOne(m.glossiness.mainColor.z); // :190-192
        if (spec) g.flags |= MAT_FLAG_PERFECT_SPECULAR;
        if (textured) g.flags |= MAT_FLAG_TEXTURED;
        mats[i] = g;
        matClass[i] = m.type == RTOW_MATERIAL_PROBABILISTIC_VOLUME ? MAT_CLASS_VOLUME : m.type == RTOW_MATERIAL_DIELECTRIC ? MAT_CLASS_DIELECTRIC
                      : (g.glossiness == 0.0f && g.metallic == 0.0f && !textured) ? MAT_CLASS_LAMBERT : MAT_CLASS_GENERAL;
    }

    // ---- entities -> primitives ----
    std::vector<GpuSphere> spheres(n);
    std::vector<GpuMotion> motion(n);
    std::vector<GpuPrim> prims;
    std::vector<float> cullBoxes;
    std::vector<uint32_t> matIndex(n);
    bool hasMotion = false, general = hasVolumes || hasImageTextures;   // volume / textured scenes always take the general-entity path
    for (int i = 0; i < n; i++) {
        const RtowEntity& e = desc->entities[i];
        const bool identity = e.rotation.x == 0.0f && e.rotation.y == 0.0f && e.rotation.z == 0.0f && e.rotation.w == 1.0f;
        if (e.type != RTOW_ENTITY_SPHERE || !identity) general = true;
    }
    if (general) prims.resize(n);
    cullBoxes.assign((size_t)n * 8, 0.0f);
    Builder b;
    b.primBox.resize(n);
    for (int a = 0; a < 3; a++) b.centroid[a].resize(n);
    for (int i = 0; i < n; i++) {
        const RtowEntity& e = desc->entities[i];
        if (e.type != RTOW_ENTITY_SPHERE && e.type != RTOW_ENTITY_RECT && e.type != RTOW_ENTITY_BOX && e.type != RTOW_ENTITY_TRIANGLE) {
            *err = "unknown entity type";
            return RTOW_ERROR_INVALID_VALUE;
        }
        if (e.materialIndex < 0 || e.materialIndex >= desc->materialCount) {
            *err = "materialIndex out of range";
            return RTOW_ERROR_INVALID_VALUE;
        }
        if (e.moving && e.timeRange.x == e.timeRange.y) { // RT/Entity.cs:53-54
            *err = "time range cannot be empty for moving entities";
            return RTOW_ERROR_INVALID_VALUE;
        }
        if (e.type == RTOW_ENTITY_TRIANGLE && (!desc->triangles || e.contentIndex < 0 || e.contentIndex >= desc->triangleCount)) {
            *err = "triangle contentIndex out of range";
            return RTOW_ERROR_INVALID_VALUE;
        }
        spheres[i] = GpuSphere{e.position.x, e.position.y, e.position.z, e.size.x};
        motion[i] = GpuMotion{e.destinationOffset.x, e.destinationOffset.y, e.destinationOffset.z, e.timeRange.x, e.timeRange.y, e.moving ? 1 : 0, {0, 0}};
        matIndex[i] = (uint32_t)e.materialIndex | (matClass[e.materialIndex] << 16) | ((uint32_t)e.type << kPrimTypeShift);
        hasMotion |= e.moving != 0;

        // local (content) bounds: Sphere.cs:16-23, Rect.cs:17-19, Box.cs:17, Triangle.cs:37-49
        float lo[3], hi[3];
        if (e.type == RTOW_ENTITY_SPHERE) {
            const float r = std::fabs(e.size.x);
            for (int a = 0; a < 3; a++) { lo[a] = -r; hi[a] = r; }
        } else if (e.type == RTOW_ENTITY_RECT) {
            lo[0] = -e.size.x / 2; hi[0] = e.size.x / 2; lo[1] = -e.size.y / 2; hi[1] = e.size.y / 2; lo[2] = -0.001f; hi[2] = 0.001f;
        } else if (e.type == RTOW_ENTITY_BOX) {
            lo[0] = -e.size.x / 2; hi[0] = e.size.x / 2; lo[1] = -e.size.y / 2; hi[1] = e.size.y / 2; lo[2] = -e.size.z / 2; hi[2] = e.size.z / 2;
        } else {
            const RtowTriangle& t = desc->triangles[e.contentIndex];
            const float v[3][3] = {{t.data[2].x, t.data[2].y, t.data[2].z},
                                   {t.data[1].x + t.data[2].x, t.data[1].y + t.data[2].y, t.data[1].z + t.data[2].z},
                                   {t.data[0].x + t.data[2].x, t.data[0].y + t.data[2].y, t.data[0].z + t.data[2].z}};
            const float nn[3][3] = {{t.normals[0].x, t.normals[0].y, t.normals[0].z}, {t.normals[1].x, t.normals[1].y, t.normals[1].z}, {t.normals[2].x, t.normals[2].y, t.normals[2].z}};
            for (int a = 0; a < 3; a++) {
                lo[a] = FLT_MAX; hi[a] = -FLT_MAX;
                for (int k = 0; k < 3; k++) {
                    lo[a] = std::min(lo[a], v[k][a] - std::fabs(nn[k][a]) * 0.001f);
                    hi[a] = std::max(hi[a], v[k][a] + std::fabs(nn[k][a]) * 0.001f);
                }
            }
        }
        // world bounds (UNITY/BvhNodeData.cs:41-80): the 8 corners through the entity transform; moving entities take the union of
        // the start and end positions.  Padded so that the kernel's slab test (own rounding) stays conservative.
        const double qx = e.rotation.x, qy = e.rotation.y, qz = e.rotation.z, qw = e.rotation.w;
        Box bx;
        bx.reset();
        for (int c = 0; c < 8; c++) {
            const double vx = (c & 1) ? hi[0] : lo[0], vy = (c & 2) ? hi[1] : lo[1], vz = (c & 4) ? hi[2] : lo[2];
            // rotate(q, v) = v + q.w * t + cross(q.xyz, t), t = 2 * cross(q.xyz, v)
            const double tx = 2 * (qy * vz - qz * vy), ty = 2 * (qz * vx - qx * vz), tz = 2 * (qx * vy - qy * vx);
            const double rx = vx + qw * tx + (qy * tz - qz * ty), ry = vy + qw * ty + (qz * tx - qx * tz), rz = vz + qw * tz + (qx * ty - qy * tx);
            const double w[3] = {rx + e.position.x, ry + e.position.y, rz + e.position.z};
            const double d[3] = {e.moving ? e.destinationOffset.x : 0.0, e.moving ? e.destinationOffset.y : 0.0, e.moving ? e.destinationOffset.z : 0.0};
            for (int a = 0; a < 3; a++) {
                bx.lo[a] = std::min(bx.lo[a], (float)std::min(w[a], w[a] + d[a]));
                bx.hi[a] = std::max(bx.hi[a], (float)std::max(w[a], w[a] + d[a]));
            }
        }
        {
            // The box the reference's own tree gives this entity (BvhBuildingEntity, UNITY/BvhNodeData.cs:23-81), in its fp32 arithmetic.
            // In the reference a primitive is only ever tested when the ray passes ITS box (single-entity leaves), and that guard is part
            // of the result: the exact test of a far, small sphere reports hits for rays that graze its box from outside.  The GPU tree
            // therefore carries exactly this box on its leaf children (inner boxes stay padded unions); DetermineVolumeContainment's
            // backwards probe uses the same boxes.
            const float q[4] = {e.rotation.x, e.rotation.y, e.rotation.z, e.rotation.w};
            const float p0[3] = {e.position.x, e.position.y, e.position.z};
            float pMin[3], pMax[3];
            for (int a = 0; a < 3; a++) {
                const float dst = p0[a] + (&e.destinationOffset.x)[a];
                pMin[a] = e.moving ? std::min(p0[a], dst) : p0[a];
                pMax[a] = e.moving ? std::max(p0[a], dst) : p0[a];
            }
            float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int c = 0; c < 8; c++) {
                const float v[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                const float t[3] = {2.0f * (q[1] * v[2] - q[2] * v[1]), 2.0f * (q[2] * v[0] - q[0] * v[2]), 2.0f * (q[0] * v[1] - q[1] * v[0])};
                const float u[3] = {q[1] * t[2] - q[2] * t[1], q[2] * t[0] - q[0] * t[2], q[0] * t[1] - q[1] * t[0]};
                for (int a = 0; a < 3; a++) {
                    const float r = v[a] + q[3] * t[a] + u[a];
                    cl[a] = std::min(cl[a], r + pMin[a]);
                    ch[a] = std::max(ch[a], r + pMax[a]);
                }
            }
            cullBoxes[i * 8 + 0] = cl[0]; cullBoxes[i * 8 + 1] = cl[1]; cullBoxes[i * 8 + 2] = cl[2];
            cullBoxes[i * 8 + 4] = ch[0]; cullBoxes[i * 8 + 5] = ch[1]; cullBoxes[i * 8 + 6] = ch[2];
        }
        for (int a = 0; a < 3; a++) {
            const float ext = bx.hi[a] - bx.lo[a];
            const float pad = 1e-5f * std::max(std::max(std::fabs(bx.lo[a]), std::fabs(bx.hi[a])), 1.0f) + 1e-5f * ext;
            b.centroid[a][i] = 0.5f * (bx.lo[a] + bx.hi[a]);
            bx.lo[a] -= pad;
            bx.hi[a] += pad;
        }
        b.primBox[i] = bx;

        if (general) {
            GpuPrim g{};
            if (e.type == RTOW_ENTITY_TRIANGLE) {
                memcpy(g.q, &desc->triangles[e.contentIndex], sizeof(RtowTriangle));   // q0..q5
                g.q[24] = e.rotation.x; g.q[25] = e.rotation.y; g.q[26] = e.rotation.z; g.q[27] = e.rotation.w;
            } else {
                g.q[0] = e.rotation.x; g.q[1] = e.rotation.y; g.q[2] = e.rotation.z; g.q[3] = e.rotation.w;
                g.q[8] = e.position.x; g.q[9] = e.position.y; g.q[10] = e.position.z;
                const int32_t mv = e.moving ? 1 : 0;
                memcpy(&g.q[11], &mv, 4);
                g.q[12] = e.destinationOffset.x; g.q[13] = e.destinationOffset.y; g.q[14] = e.destinationOffset.z; g.q[15] = e.timeRange.x;
                g.q[16] = e.timeRange.y;
                if (e.type == RTOW_ENTITY_SPHERE) {
                    g.q[20] = e.size.x;
                } else if (e.type == RTOW_ENTITY_RECT) {                 // Rect.cs:12-16: From = -size / 2, To = size / 2
                    g.q[20] = -e.size.x / 2; g.q[21] = -e.size.y / 2; g.q[22] = e.size.x / 2; g.q[23] = e.size.y / 2;
                } else {                                                  // Box.cs:11-15: Extents = size / 2, InverseExtents = 1 / Extents
                    g.q[20] = e.size.x / 2; g.q[21] = e.size.y / 2; g.q[22] = e.size.z / 2;
                    g.q[23] = 1 / g.q[20]; g.q[24] = 1 / g.q[21]; g.q[25] = 1 / g.q[22];
                }
            }
            prims[i] = g;
        }
    }

    // What the reference's own tree contributes to the result (rtow_reforder.h): the order of hits at bit-identical distances (the entity's
    // place in its leaf order), and the box that guards each entity's exact test - the bounds of the reference LEAF it sits in, which is the
    // entity's own box except in leaves forced at MaxBvhDepth, where it is their union.
    std::vector<float> guardBoxes;
    std::vector<RefTreeNode> refTree;      // the tree RebuildBvh would build: only its FULL_DIAGNOSTICS counters need it on the device
    const std::vector<uint32_t> ranks = referenceLeafRanks(cullBoxes, n, desc->maxBvhDepth > 0 ? desc->maxBvhDepth : 32 /* prefab default */, &guardBoxes, &refTree);
    out->refTree.assign(reinterpret_cast<const uint8_t*>(refTree.data()), reinterpret_cast<const uint8_t*>(refTree.data()) + refTree.size() * sizeof(RefTreeNode));
    out->refTreeDepth = desc->maxBvhDepth > 0 ? desc->maxBvhDepth : 32;
    for (int i = 0; i < n; i++) {
        bool wider = false;
        for (int a = 0; a < 3; a++) wider |= guardBoxes[(size_t)i * 8 + a] < cullBoxes[(size_t)i * 8 + a] || guardBoxes[(size_t)i * 8 + 4 + a] > cullBoxes[(size_t)i * 8 + 4 + a];
        if (!wider) continue;
        // a forced leaf: the GPU tree must enclose the (larger) guard box too, so that its inner boxes never cull what the reference tests
        Box& bx = b.primBox[i];
        for (int a = 0; a < 3; a++) {
            const float lo = guardBoxes[(size_t)i * 8 + a], hi = guardBoxes[(size_t)i * 8 + 4 + a];
            const float pad = 1e-5f * std::max(std::max(std::fabs(lo), std::fabs(hi)), 1.0f) + 1e-5f * (hi - lo);
            bx.lo[a] = std::min(bx.lo[a], lo - pad);
            bx.hi[a] = std::max(bx.hi[a], hi + pad);
            b.centroid[a][i] = 0.5f * (bx.lo[a] + bx.hi[a]);
        }
    }
    cullBoxes = guardBoxes;   // from here on: leaf-child boxes of the GPU tree and the backwards probe's guard

    // All-triangle scenes without volumes (SCENE_KIND_TRIANGLES / _TEXTURED) number their primitives in this tree's leaf order and get the compact GpuTriHot / GpuTriCold records
    // (rtow_scene.h): the candidates of neighbouring leaves then sit next to each other in memory.  Results-neutral: a primitive number never leaves the library
    // (the probe maps it back), and ties are decided by rank[] - the place in the REFERENCE tree's leaf order - which is permuted along.
    bool allTriangles = general;
    for (int i = 0; i < n && allTriangles; i++) allTriangles = desc->entities[i].type == RTOW_ENTITY_TRIANGLE;
    const bool triKind = allTriangles && !hasVolumes;
    std::vector<int> placeOf;                // entity -> place in leaf order (triKind only)
    auto leafPlace = [&](int entity) { return triKind && !placeOf.empty() ? placeOf[entity] : entity; };
    std::vector<uint32_t> ranksInLeafOrder(triKind ? n : 0);
    out->entityOfPrim.clear();

    // ---- SAH build, then breadth-first renumbering ----
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    Box rootBox;
    std::vector<GpuNode> gnodes;
    int depthSeen = 0;
    if (n == 1) {
        GpuNode g{};
        Box empty;
        empty.reset(); // lo = +FLT_MAX, hi = -FLT_MAX: never hit
        Box only;
        for (int a = 0; a < 3; a++) { only.lo[a] = cullBoxes[a]; only.hi[a] = cullBoxes[4 + a]; }   // the reference's own entity box, like every leaf child
        setBoxes(g, only, empty);
        g.child0 = ~0; g.child1 = ~0;
        gnodes.push_back(g);
        depthSeen = 1;
    } else {
        b.nodes.reserve(n);
        const int root = b.build(idx, 0, n, maxDepth, 0, &rootBox);
        depthSeen = b.maxDepthSeen;
        if (triKind) {                       // build() partitions idx in place: what is left is the leaves from left to right
            placeOf.resize(n);
            for (int k = 0; k < n; k++) placeOf[idx[k]] = k;
        }
        // Numbering.  The first kBreadthFirstNodes nodes are numbered breadth-first - "the first K nodes" must be the top levels, because a scene
        // that does not fit LDS stages a prefix of the node array (at most (160 KB - 64 KB) / 64 B = 1 535 nodes) - and every subtree hanging
        // below that front is numbered depth-first (pre-order): a ray that has descended into a subtree keeps reading nodes that lie next to
        // each other in memory, instead of one 64-byte line per level spread over a 16-32 MB array (a 250 000-triangle mesh: 500 000 nodes; each
        // XCD's L2 holds 4 MB).  Children still come after their parent.  Scenes of up to kBreadthFirstNodes nodes get the order they always got.
        constexpr size_t kBreadthFirstNodes = 2048;
        std::vector<int> newIndex(b.nodes.size(), -1);
        std::vector<int> bfs;
        bfs.reserve(b.nodes.size());
        std::queue<int> q;
        q.push(root);
        while (!q.empty() && bfs.size() < kBreadthFirstNodes) {
            const int t = q.front();
            q.pop();
            newIndex[t] = (int)bfs.size();
            bfs.push_back(t);
            for (int c = 0; c < 2; c++) if (b.nodes[t].child[c] >= 0) q.push(b.nodes[t].child[c]);
        }
        std::vector<int> dfs;
        while (!q.empty()) {                                  // the front: one depth-first run per subtree, in breadth-first order of their roots
            dfs.push_back(q.front());
            q.pop();
            while (!dfs.empty()) {
                const int t = dfs.back();
                dfs.pop_back();
                newIndex[t] = (int)bfs.size();
                bfs.push_back(t);
                for (int c = 1; c >= 0; c--) if (b.nodes[t].child[c] >= 0) dfs.push_back(b.nodes[t].child[c]);   // child 0 next
            }
        }
        gnodes.resize(bfs.size());
        for (size_t i = 0; i < bfs.size(); i++) {
            const TmpNode& t = b.nodes[bfs[i]];
            GpuNode g{};
            Box cb[2] = {t.box[0], t.box[1]};
            for (int c = 0; c < 2; c++)
                if (t.child[c] < 0) {                                         // leaf child: the reference's own entity box, unpadded
                    const float* e = &cullBoxes[(size_t)(~t.child[c]) * 8];
                    for (int a = 0; a < 3; a++) { cb[c].lo[a] = e[a]; cb[c].hi[a] = e[4 + a]; }
                }
            setBoxes(g, cb[0], cb[1]);
            g.child0 = t.child[0] >= 0 ? newIndex[t.child[0]] : ~leafPlace(~t.child[0]);
            g.child1 = t.child[1] >= 0 ? newIndex[t.child[1]] : ~leafPlace(~t.child[1]);
            gnodes[i] = g;
        }
    }
    if (triKind && n > 1) {
        // every per-entity array follows the entities into leaf order; entityOfPrim takes a primitive number back to the host's entity index (rtowProbeNearestHit)
        auto permute = [&](auto& v, size_t per) {
            if (v.empty()) return;
            auto w = v;
            for (int i = 0; i < n; i++) std::copy(w.begin() + (size_t)i * per, w.begin() + (size_t)(i + 1) * per, v.begin() + (size_t)placeOf[i] * per);
        };
        permute(spheres, 1); permute(motion, 1); permute(prims, 1); permute(cullBoxes, 8); permute(matIndex, 1);
        std::vector<uint32_t> r = ranks;
        for (int i = 0; i < n; i++) ranksInLeafOrder[placeOf[i]] = r[i];
        out->entityOfPrim.assign(n, 0);
        for (int i = 0; i < n; i++) out->entityOfPrim[placeOf[i]] = i;
    }
    {
        // section offsets are 32-bit: the whole image must stay below 4 GiB
        const uint64_t bytes = (uint64_t)gnodes.size() * sizeof(GpuNode) + (uint64_t)n * (sizeof(GpuSphere) + (hasMotion ? sizeof(GpuMotion) : 0) + (general ? sizeof(GpuPrim) : 0) + (hasVolumes ? 32u : 0u) + (triKind ? sizeof(GpuTriHot) + sizeof(GpuTriCold) : 0) + 8u) +
                               (uint64_t)mats.size() * sizeof(GpuMaterial) + 256u;
        if (bytes >= 0xfff00000ull) {
            *err = "scene image exceeds 4 GiB";
            return RTOW_ERROR_CAPACITY;
        }
    }

    // ---- pack the blob ----
    SceneLayout L{};
    uint32_t off = 0;
    L.nodeOffset = off; L.nodeCount = (uint32_t)gnodes.size(); off = align16(off + L.nodeCount * (uint32_t)sizeof(GpuNode));
    L.sphereOffset = off; L.sphereCount = (uint32_t)n; off = align16(off + (uint32_t)n * (uint32_t)sizeof(GpuSphere));
    L.hasMotion = hasMotion ? 1u : 0u;
    if (hasMotion) {
        // do all moving entities share one TimeRange?  (bit comparison: the hoisted expression must be the very same float program)
        bool first = true, same = true;
        uint32_t t0 = 0, t1 = 0;
        for (int i = 0; i < n && same; i++) {
            const RtowEntity& e = desc->entities[i];
            if (!e.moving) continue;
            uint32_t a, b;
            memcpy(&a, &e.timeRange.x, 4); memcpy(&b, &e.timeRange.y, 4);
            if (first) { t0 = a; t1 = b; first = false; }
            else same = a == t0 && b == t1;
        }
        if (same && !first) { L.commonTimeRange = 1u; memcpy(&L.commonT0, &t0, 4); memcpy(&L.commonT1, &t1, 4); }
    }
    L.motionOffset = off; if (hasMotion) off = align16(off + (uint32_t)n * (uint32_t)sizeof(GpuMotion));
    L.sceneKind = hasVolumes ? (hasImageTextures ? SCENE_KIND_VOLUMES_TEXTURED : SCENE_KIND_VOLUMES) : hasImageTextures ? (allTriangles ? SCENE_KIND_TRIANGLES_TEXTURED : SCENE_KIND_TEXTURED) : allTriangles ? SCENE_KIND_TRIANGLES
                  : general ? SCENE_KIND_GENERAL : hasMotion ? SCENE_KIND_SPHERES_MOTION : SCENE_KIND_SPHERES;
    // Does the scene hold the same primitive twice (same geometry, any material)?  Two such surfaces coincide everywhere - the same float
    // program produces both distances - and tie at the nearest hit of whole image regions; then the kernel variant that settles nearest-hit
    // ties through the reference's whole hit list is used (kExactTiesBit).  Without duplicates a tie needs two different
    // float programs to agree to the last bit (faces in one plane, shared mesh edges) and the leaf-order rule stands, exact up to 16 hits
    // per ray.  Volume scenes resolve every tie in their hit list anyway.
    L.exactTies = 0u;
    if (!hasVolumes && (general || hasImageTextures) && n > 16) L.exactTies = 1u;      // see below; decided first so that a million-triangle mesh skips the duplicate scan
    if (triKind && L.exactTies) {
        // An all-triangle scene may run on the rank-rule kernels with the tie watch (the exact-tie kernels then render only the marked pixels) unless it holds the same
        // triangle twice - coinciding surfaces tie over whole image regions.  Triangles are tested in world space on `Data` alone (RT/Entity.cs:91-93, RT/HitTests.cs:116-139): 64-bit
        // hashes of those nine floats (-0 folded into +0), sorted; equal hashes count as duplicates without a second look (a collision only costs speed: the scene keeps
        // the exact-tie kernels).
        std::vector<uint64_t> h(n);
        for (int i = 0; i < n; i++) {
            const float* f = reinterpret_cast<const float*>(&desc->triangles[desc->entities[i].contentIndex]);
            uint64_t x = 0x9E3779B97F4A7C15ull;
            for (int k = 0; k < 9; k++) {
                const float v = f[k] + 0.0f;
                uint32_t w; memcpy(&w, &v, 4);
                x ^= w; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
            }
            h[i] = x;
        }
        std::sort(h.begin(), h.end());
        L.tieWatchOk = std::adjacent_find(h.begin(), h.end()) == h.end() ? 1u : 0u;
    }
    if (!hasVolumes && !L.exactTies) {
        std::vector<std::array<uint32_t, 38>> keys(n);
        for (int i = 0; i < n; i++) {
            const RtowEntity& e = desc->entities[i];
            float f[38] = {0};
            f[0] = (float)e.type;
            if (e.type == RTOW_ENTITY_TRIANGLE) {
                memcpy(&f[1], &desc->triangles[e.contentIndex], 24 * sizeof(float));       // data, normals, texture coordinates
            } else {
                f[1] = e.position.x; f[2] = e.position.y; f[3] = e.position.z;
                f[4] = e.rotation.x; f[5] = e.rotation.y; f[6] = e.rotation.z; f[7] = e.rotation.w;
                f[8] = e.type == RTOW_ENTITY_SPHERE ? std::fabs(e.size.x) : e.size.x;       // a sphere and its negative-radius twin hit alike
                f[9] = e.type == RTOW_ENTITY_SPHERE ? 0.0f : e.size.y;
                f[10] = e.type == RTOW_ENTITY_BOX ? e.size.z : 0.0f;
            }
            if (e.moving) { f[30] = 1.0f; f[31] = e.destinationOffset.x; f[32] = e.destinationOffset.y; f[33] = e.destinationOffset.z; f[34] = e.timeRange.x; f[35] = e.timeRange.y; }
            for (int k = 0; k < 38; k++) { f[k] += 0.0f; memcpy(&keys[i][k], &f[k], 4); }              // + 0: -0 and +0 are the same place
        }
        std::sort(keys.begin(), keys.end());
        for (int i = 1; i < n; i++) if (keys[i] == keys[i - 1]) { L.exactTies = 1u; break; }
        // Rects, boxes and triangles do tie without being duplicates (faces in one plane, shared mesh edges), and the leaf-order rule is the
        // reference's answer only while a ray has at most 16 hits: with more than 16 such entities in the scene a ray can have more, so those
        // scenes get the exact-tie kernels too (20 % slower on a mesh, far more where every ray ties; RTOW_CONTEXT_EXACT_TIES_NEVER keeps the rank rule).  Sphere-only scenes do not:
        // two different spheres meeting a ray at bit-identical distance is not a situation geometry produces.
        if ((general || hasImageTextures) && n > 16) L.exactTies = 1u;
    }
    L.primOffset = off; if (general) off = align16(off + (uint32_t)n * (uint32_t)sizeof(GpuPrim));
    L.cullOffset = off; if (hasVolumes) off = align16(off + (uint32_t)n * 32u);
    L.rankOffset = off; off = align16(off + (uint32_t)n * 4u);
    L.matIndexOffset = off; off = align16(off + (uint32_t)n * 4u);
    if (triKind) {
        off = (off + 127u) & ~127u;              // records never straddle a 64-byte memory sector they need not
        L.triHotOffset = off; off = align16(off + (uint32_t)n * (uint32_t)sizeof(GpuTriHot));
        off = (off + 127u) & ~127u;
        L.triColdOffset = off; off = align16(off + (uint32_t)n * (uint32_t)sizeof(GpuTriCold));
    }
    L.materialOffset = off; L.materialCount = (uint32_t)mats.size(); off = align16(off + L.materialCount * (uint32_t)sizeof(GpuMaterial));
    L.totalBytes = off;
    L.bvhDepth = (uint32_t)depthSeen;

    out->blob.assign(off, 0);
    memcpy(out->blob.data() + L.nodeOffset, gnodes.data(), gnodes.size() * sizeof(GpuNode));
    memcpy(out->blob.data() + L.sphereOffset, spheres.data(), spheres.size() * sizeof(GpuSphere));
    if (hasMotion) memcpy(out->blob.data() + L.motionOffset, motion.data(), motion.size() * sizeof(GpuMotion));
    if (general) memcpy(out->blob.data() + L.primOffset, prims.data(), prims.size() * sizeof(GpuPrim));
    if (hasVolumes) memcpy(out->blob.data() + L.cullOffset, cullBoxes.data(), cullBoxes.size() * 4u);
    memcpy(out->blob.data() + L.rankOffset, triKind && n > 1 ? ranksInLeafOrder.data() : ranks.data(), ranks.size() * 4u);
    if (triKind) {
        GpuTriHot* hot = reinterpret_cast<GpuTriHot*>(out->blob.data() + L.triHotOffset);
        GpuTriCold* cold = reinterpret_cast<GpuTriCold*>(out->blob.data() + L.triColdOffset);
        for (int i = 0; i < n; i++) {
            const float* q = prims[i].q;             // RtowTriangle verbatim: Data (e0 e1 v0), Normals, TextureCoordinates; q[24..27] = rotation
            memcpy(hot[i].e0, q, 9 * sizeof(float));
            memcpy(cold[i].n0, q + 9, 9 * sizeof(float));
            memcpy(cold[i].rot, q + 24, 4 * sizeof(float));
        }
    }
    memcpy(out->blob.data() + L.matIndexOffset, matIndex.data(), matIndex.size() * 4u);
    memcpy(out->blob.data() + L.materialOffset, mats.data(), mats.size() * sizeof(GpuMaterial));
    out->layout = L;
    // ---- Image textures: a second blob, HBM only ----
    out->texBlob.clear();
    out->texLayout = TexLayout{};
    if (hasImageTextures) {
        TexLayout T{};
        std::vector<GpuTexMaterial> tm(desc->materialCount);
        for (int i = 0; i < desc->materialCount; i++) {
            const RtowMaterial& m = desc->materials[i];
            tm[i] = GpuTexMaterial{packTexture(m.albedo), packTexture(m.glossiness), packTexture(m.emission), packTexture(m.metallic)};
        }
        std::vector<GpuImage> images(desc->imageCount > 0 ? desc->imageCount : 0);
        uint64_t pixelBytes = 0;
        for (size_t i = 0; i < images.size(); i++) {
            const RtowImage& im = desc->images[i];
            if (im.width <= 0 || im.height <= 0 || im.width > 32768 || im.height > 32768 || im.pixelStride < 3 || im.pixelStride > 16 || !im.pixels) {
                *err = "bad image (size, pixelStride < 3 or null pixels)";
                return RTOW_ERROR_INVALID_VALUE;
            }
            images[i] = GpuImage{(uint32_t)pixelBytes, im.width, im.height, im.pixelStride};
            pixelBytes += ((uint64_t)im.width * im.height * im.pixelStride + 15u) & ~15ull;
            if (pixelBytes > 0xf0000000ull) {
                *err = "more than 3.75 GiB of image pixels";
                return RTOW_ERROR_CAPACITY;
            }
        }
        T.materialOffset = 0;
        T.imageOffset = align16((uint32_t)(tm.size() * sizeof(GpuTexMaterial)));
        T.pixelOffset = align16(T.imageOffset + (uint32_t)(images.size() * sizeof(GpuImage)));
        const uint64_t total = (uint64_t)T.pixelOffset + pixelBytes;
        if (total > 0xffffffffull) {
            *err = "texture blob exceeds 4 GiB";
            return RTOW_ERROR_CAPACITY;
        }
        T.totalBytes = (uint32_t)total;
        out->texBlob.assign(T.totalBytes, 0);
        memcpy(out->texBlob.data() + T.materialOffset, tm.data(), tm.size() * sizeof(GpuTexMaterial));
        if (!images.empty()) memcpy(out->texBlob.data() + T.imageOffset, images.data(), images.size() * sizeof(GpuImage));
        for (size_t i = 0; i < images.size(); i++)
            memcpy(out->texBlob.data() + T.pixelOffset + images[i].offset, desc->images[i].pixels, (size_t)images[i].width * images[i].height * images[i].pixelStride);
        out->texLayout = T;
    }
    out->entityCount = n;
    out->materialCount = desc->materialCount;
    return RTOW_SUCCESS;
}

} // namespace rtow
