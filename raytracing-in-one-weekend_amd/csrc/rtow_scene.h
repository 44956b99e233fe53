// rtow_scene.h - flat GPU scene layout shared by the host-side builder (rtow_bvh.cpp) and the kernels.
//
// The reference's runtime scene is a pointer graph (RT/BvhNode.cs:7-10: AABB + Left/Right/EntitiesStart pointers,
// RT/Entity.cs:27-37: 104-byte Entity with two RigidTransforms, Material*, void* Content).  None of that can live
// in LDS or be chased efficiently by 64-wide wavefronts, so the native layout is index based and sized for LDS:
//
//   GpuNode   64 B  one INNER node holding the boxes of BOTH children (one 64-byte fetch decides both subtrees,
//                   near child first).  child >= 0: inner node index; child < 0: leaf, primitive index = ~child.
//                   Nodes are stored breadth-first, so "the first K nodes" == "the top levels" when a scene
//                   is too large for LDS and only a prefix is staged.
//   GpuSphere 16 B  centre + signed radius (RT/EntityTypes/Sphere.cs:8; the centre is Entity.OriginTransform.pos).
//   GpuMotion 32 B  only for scenes with moving entities: DestinationOffset + TimeRange (RT/Entity.cs:33-34).
//   GpuMaterial 64 B constant-texture material + per-material derived constants (RT/Material.cs:16-47 with RT/Texture.cs constant branches folded).
//
// Everything is packed into ONE device blob (16-byte aligned sections) so a workgroup stages it into LDS with a
// single coalesced 16-byte-per-lane copy.
#pragma once
#include <stdint.h>

namespace rtow {

struct GpuNode {
    // bounds interleaved as (child0, child1) pairs so both boxes are tested with packed fp32 math:
    // q0 = (lo0.x lo1.x lo0.y lo1.y)  q1 = (lo0.z lo1.z hi0.x hi1.x)  q2 = (hi0.y hi1.y hi0.z hi1.z)  q3 = (c0 c1 - -)
    float lox[2], loy[2];
    float loz[2], hix[2];
    float hiy[2], hiz[2];
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(GpuNode) == 64, "GpuNode must be 64 bytes");

struct GpuSphere {
    float cx, cy, cz, radius;
};
static_assert(sizeof(GpuSphere) == 16, "GpuSphere must be 16 bytes");

struct GpuMotion {
    float dx, dy, dz;   // DestinationOffset
    float t0, t1;       // TimeRange
    int32_t moving;     // Entity.Moving
    int32_t pad[2];
};
static_assert(sizeof(GpuMotion) == 32, "GpuMotion must be 32 bytes");

enum : uint32_t {
    MAT_FLAG_PERFECT_SPECULAR = 1u, // Material.IsPerfectSpecular (RT/Material.cs:181-196)
};
// Shading class of a material; the scene compiler packs it into bits 16.. of materialIndex[] so the kernel can sort a
// hit into its scheduler stage with one LDS read.
enum : uint32_t {
    MAT_CLASS_LAMBERT = 0, // Standard, glossiness == 0 and metallic == 0
    MAT_CLASS_GENERAL = 1, // any other Standard
    MAT_CLASS_DIELECTRIC = 2,
    MAT_CLASS_VOLUME = 3,  // ProbabilisticVolume
};

struct GpuMaterial {
    float albedo[3];     // Albedo.SampleColor
    float emission[3];   // Emission.SampleColor
    int32_t type;        // RtowMaterialType
    float metallic;      // Metallic.SampleScalar
    float glossiness;    // Glossiness.SampleScalar
    float parameter;     // IndexOfRefraction / Density
    uint32_t flags;
    // derived on the device by prepare_materials_kernel (same float program as the per-hit code):
    float roughness;     // Standard: pow(1 - glossiness, 2); Dielectric: 1 - glossiness
    float alpha;         // Standard: RoughnessToAlpha(roughness)
    float ior;           // Standard: lerp(1.5, 1.1, metallic); Dielectric: parameter
    float r0;            // Schlick r0 = ((1 - ior) / (1 + ior))^2
    float invIor;        // Dielectric: 1 / ior
};
static_assert(sizeof(GpuMaterial) == 64, "GpuMaterial must be 64 bytes");

// Image textures (RT/Texture.cs:80-89,126-135) live in a second blob that stays in HBM (images do not fit LDS):
//   GpuTexMaterial[materialCount]  |  GpuImage[imageCount]  |  pixel bytes
struct GpuTexture {
    int32_t type;        // RtowTextureType
    int32_t image;       // index into the GpuImage table, < 0: null ImagePointer
    int32_t channel;     // ScalarValueChannel
    float parameter;     // ConstantValue
    float mainColor[3];
    float pad;
};
struct GpuTexMaterial {
    GpuTexture albedo, glossiness, emission, metallic;
};
struct GpuImage {
    uint32_t offset;     // of the first pixel, from the start of the pixel section
    int32_t width, height, pixelStride;
};
static_assert(sizeof(GpuTexture) == 32 && sizeof(GpuTexMaterial) == 128 && sizeof(GpuImage) == 16, "texture blob records");
struct TexLayout {
    uint32_t materialOffset, imageOffset, pixelOffset, totalBytes;   // totalBytes == 0: the scene has no Image texture
};
constexpr uint32_t MAT_FLAG_TEXTURED = 2u;      // GpuMaterial.flags: at least one of the four textures is an Image - evaluate per hit

// General primitive record (128 B) for scenes that are not identity-rotation spheres: Rect / Box / Triangle entities,
// rotated entities (RT/Entity.cs:27-127).  One fixed-stride array so a leaf code indexes it directly.
//   transformed entity (sphere / rect / box):
//     q0 rot  q1 inverse(rot)*  q2 (pos.xyz, moving)  q3 (DestinationOffset.xyz, t0)  q4 (t1, inverse translation*.xyz)
//     q5 (p0 p1 p2 p3)  q6 (p4 p5 p6 -)      sphere: p0 = radius; rect: From.xy To.xy; box: Extents.xyz, InverseExtents.xyz (p3..p5)
//   triangle: q0..q5 = RtowTriangle verbatim (Data, Normals, TextureCoordinates), q6 = rot
//   (* derived on the device by prepare_entities_kernel with the path's own float program)
struct GpuPrim {
    float q[32];
};
static_assert(sizeof(GpuPrim) == 128, "GpuPrim must be 128 bytes");

// All-triangle scenes (SCENE_KIND_TRIANGLES*: what the reference's live host produces, one entity per mesh triangle, UNITY/Raytracer.cs:1193-1198) number their
// entities in the LEAF ORDER of this library's tree and carry, next to the GpuPrim records, two compact ones:
//   GpuTriHot   what the exact test reads for EVERY candidate: the two edges and the first vertex (RT/EntityTypes/Triangle.cs:8-12 `Data`), one 64-byte memory sector
//               per triangle, siblings of a leaf parent next to each other - a quarter-million-triangle mesh keeps 16 MB of these hot instead of 32 MB of GpuPrim
//   GpuTriCold  what only the WINNER needs, in HIT: the three vertex normals and the entity's rotation (the barycentric blend of RT/HitTests.cs:140-146 runs there, on
//               the (u, v) the test kept - the same float program on the same operands)
// Texture coordinates stay in the GpuPrim record (textured meshes re-run the winner's test on it in HIT, as before).
#ifndef RTOW_TRI_HOT_BYTES
#define RTOW_TRI_HOT_BYTES 64
#endif
struct GpuTriHot {
    float e0[3], e1[3], v0[3];                       // q0 = (e0.xyz e1.x)  q1 = (e1.yz v0.xy)  q2 = (v0.z - - -)
    float pad[RTOW_TRI_HOT_BYTES / 4 - 9];
};
struct GpuTriCold {
    float n0[3], n1[3], n2[3];                       // q0 = (n0.xyz n1.x)  q1 = (n1.yz n2.xy)  q2 = (n2.z - - -)
    float pad[3];
    float rot[4];                                    // q3
};
static_assert(sizeof(GpuTriHot) == RTOW_TRI_HOT_BYTES && (RTOW_TRI_HOT_BYTES == 48 || RTOW_TRI_HOT_BYTES == 64) && sizeof(GpuTriCold) == 64, "triangle records");

enum : uint32_t {
    SCENE_KIND_SPHERES = 0,        // identity-rotation static spheres: GpuSphere only
    SCENE_KIND_SPHERES_MOTION = 1, // identity-rotation spheres, some moving: GpuSphere + GpuMotion
    SCENE_KIND_GENERAL = 2,        // anything else: GpuPrim
    SCENE_KIND_VOLUMES = 3,        // GENERAL + at least one ProbabilisticVolume material: all hits of a ray are collected and sorted
    SCENE_KIND_TEXTURED = 4,       // GENERAL + at least one Image texture: materials are evaluated per hit at the hit's texture coordinates
    SCENE_KIND_VOLUMES_TEXTURED = 5, // both
    SCENE_KIND_TRIANGLES = 6,      // GENERAL whose entities are ALL triangles (what the reference's live host produces: one entity per mesh triangle,
                                   // UNITY/Raytracer.cs:1193-1198): the same GpuPrim records and float program, without the type dispatch and the
                                   // transform code of the other primitives in the kernel (+11 % on a 250 000-triangle mesh)
    SCENE_KIND_TRIANGLES_TEXTURED = 7, // TEXTURED whose entities are all triangles: the same specialisation for meshes with image textures
};
// template flag OR-ed to the kind of the sample kernel: settle nearest-hit ties with the reference's whole procedure (SceneLayout::exactTies)
constexpr int kExactTiesBit = 8;
// materialIndex[] word: bits 0..15 material, 16..17 shading class, 18..20 RtowEntityType
constexpr uint32_t kPrimTypeShift = 18;

// Byte offsets of the sections inside the scene blob (all multiples of 16).
struct SceneLayout {
    uint32_t nodeOffset, nodeCount;         // GpuNode[nodeCount]
    uint32_t sphereOffset, sphereCount;     // GpuSphere[sphereCount]
    uint32_t motionOffset, hasMotion;       // GpuMotion[sphereCount] when hasMotion
    uint32_t matIndexOffset;                // uint32 materialIndex[sphereCount]
    uint32_t materialOffset, materialCount; // GpuMaterial[materialCount]
    uint32_t totalBytes;                    // size of the blob
    uint32_t bvhDepth;                      // max number of inner nodes on a root->leaf path == stack bound
    uint32_t sceneKind;                     // SCENE_KIND_*
    uint32_t exactTies;                     // scenes without volumes: two surfaces can coincide exactly (duplicate spheres; any non-sphere geometry), so the
                                            // kernel variant that resolves nearest-hit ties through the reference's whole hit list is used
    uint32_t primOffset;                    // GpuPrim[sphereCount] when sceneKind == SCENE_KIND_GENERAL
    uint32_t cullOffset;                    // float[8] {min.xyz, -, max.xyz, -} per entity when sceneKind == SCENE_KIND_VOLUMES: the reference tree's entity box
    uint32_t rankOffset;                    // uint32 per entity when sceneKind >= SCENE_KIND_GENERAL: place in the reference tree's leaf order (rtow_reforder.h)
    uint32_t commonTimeRange;               // 1: every moving entity has the same TimeRange (the generated scenes: (0, 1)); then the kernel evaluates
    float commonT0, commonT1;               //    clamp(unlerp(t0, t1, ray.Time), 0, 1) (RT/Entity.cs:124-127) once per sample instead of once per sphere test
    uint32_t triHotOffset, triColdOffset;   // GpuTriHot[sphereCount], GpuTriCold[sphereCount] when sceneKind == SCENE_KIND_TRIANGLES / _TEXTURED (entities in leaf order), else 0
    uint32_t tieWatchOk;                    // all-triangle scenes: no two entities are the same triangle, so nearest-hit ties are rare events (shared edges) and the rank-rule kernels may
                                            // trace the frame with the tie watch on, the exact-tie kernels only the marked pixels; 0: exact-tie kernels for every pixel
};

} // namespace rtow
