/*
 * rtow.h - C ABI of the MI355X-native sample-batch path (librtow_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of renaudbedard/raytracing-in-one-weekend:
 * the Burst `SampleBatchJob` (per-pixel sample loop + BVH traversal + Sphere.Hit +
 * Material.Scatter) and the post passes immediately downstream of it.  The Unity C# host
 * (scene build, camera, textures, blit, denoiser hand-off) stays; it P/Invokes these
 * entry points instead of scheduling the Burst job.  See INTEGRATION.md for the C# stub.
 *
 * Conventions follow the reference's own native-plugin boundary
 * (OptixDenoiser/OptixDenoiser/OptixDenoiser.h:1-67 and
 *  Assets/ThirdParty/nVidia OptiX Denoiser/OptixApi.cs:24-251):
 *   - extern "C", flat exported functions, cdecl;
 *   - every call returns an int error enum, 0 == success (OptixApi.cs:24-31,42-78);
 *   - opaque handles are single pointers with create(..., &handle) / destroy(handle) pairs
 *     (OptixApi.cs:172-224);
 *   - POD option structs are passed by pointer (OptixApi.cs:106-142);
 *   - host buffers are raw pointers owned by the caller for the duration of the call only
 *     (Runtime/Jobs/DenoiseJobs.cs:75-78,116-117);
 *   - explicit device alloc / copy / free (OptixDenoiser.h:57-64, OptixApi.cs:226-251).
 *
 * All file:line citations are relative to the reference repository root, with
 *   JOBS/ = RaytracingInOneWeekend/Assets/Scripts/Runtime/Jobs/
 *   RT/   = RaytracingInOneWeekend/Assets/Scripts/Runtime/
 *   UNITY/= RaytracingInOneWeekend/Assets/Scripts/Unity/
 *
 * No torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef RTOW_H
#define RTOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define RTOW_API __declspec(dllexport)
#else
#define RTOW_API __attribute__((visibility("default")))
#endif

#define RTOW_API_VERSION 11

/* ---- result codes (0 == success, like CudaError/OptixResult in OptixApi.cs:24-78) ---- */
typedef enum RtowResult {
    RTOW_SUCCESS = 0,
    RTOW_ERROR_INVALID_VALUE = 1,     /* null pointer, bad size, bad enum                      */
    RTOW_ERROR_MEMORY_ALLOCATION = 2, /* hipMalloc / host allocation failed                     */
    RTOW_ERROR_NO_DEVICE = 3,         /* no gfx950 device / HIP runtime unusable                */
    RTOW_ERROR_NO_SCENE = 4,          /* rtowSampleBatch* before rtowUploadScene                */
    RTOW_ERROR_UNSUPPORTED = 5,       /* entity / material / noise kind not built yet           */
    RTOW_ERROR_LAUNCH_FAILURE = 6,    /* kernel launch or stream error (hipError in the log)    */
    RTOW_ERROR_CANCELLED = 7,         /* cancellation flag observed; outputs unspecified        */
    RTOW_ERROR_CAPACITY = 8,          /* a bound was exceeded: at upload (entities, tree depth) or - reported by a cancellable rtowSampleBatchDevice or the
                                       * next rtowGetBatchStatus / rtowSynchronize - by a ray that met more surfaces than the hit lists hold.  The reference's
                                       * list grows without bound; with RtowContextOptions.hitListCapacity == 0 so do these: by the time the error is reported
                                       * the lists are twice as long, the host-buffer calls (rtowSampleBatch, rtowSampleBatchChain) have run the batch again
                                       * themselves and do not report it at all, and a device-resident caller issues the batch again (see hitListCapacity) */
    RTOW_ERROR_INTERNAL = 99
} RtowResult;

/* ---- small vector PODs (Unity.Mathematics float2/float3/float4, tightly packed) ---- */
typedef struct RtowFloat2 { float x, y; } RtowFloat2;
typedef struct RtowFloat3 { float x, y, z; } RtowFloat3;
typedef struct RtowFloat4 { float x, y, z, w; } RtowFloat4; /* quaternion: (x,y,z,w) = quaternion.value */

/* ---- enums mirroring the reference's runtime enums ---- */
typedef enum RtowEntityType {      /* RT/Entity.cs:13-20 */
    RTOW_ENTITY_NONE = 0,
    RTOW_ENTITY_SPHERE = 1,
    RTOW_ENTITY_RECT = 2,
    RTOW_ENTITY_BOX = 3,
    RTOW_ENTITY_TRIANGLE = 4
} RtowEntityType;

typedef enum RtowMaterialType {    /* RT/Material.cs:9-14 */
    RTOW_MATERIAL_STANDARD = 0,
    RTOW_MATERIAL_DIELECTRIC = 1,
    RTOW_MATERIAL_PROBABILISTIC_VOLUME = 2
} RtowMaterialType;

typedef enum RtowTextureType {     /* RT/Texture.cs:13-21 */
    RTOW_TEXTURE_NONE = 0,
    RTOW_TEXTURE_CONSTANT = 1,
    RTOW_TEXTURE_CHECKER_PATTERN = 2, /* dead in the reference (commented out, RT/Texture.cs:61-78) */
    RTOW_TEXTURE_PERLIN_NOISE = 3,    /* dead in the reference                                      */
    RTOW_TEXTURE_IMAGE = 4,           /* RT/Texture.cs:80-89,126-135: byte image, point sampled at the hit's texture coordinates */
    RTOW_TEXTURE_CONSTANT_SCALAR = 5
} RtowTextureType;

typedef enum RtowSkyType {         /* RT/Environment.cs:5-10 */
    RTOW_SKY_NONE = 0,
    RTOW_SKY_GRADIENT = 1,
    RTOW_SKY_CUBEMAP = 2              /* Cubemap.Sample(ray.Direction) on the faces given to rtowUploadSkyCubemap */
} RtowSkyType;

typedef enum RtowNoiseColor {      /* RT/RandomSource.cs:8-13 */
    RTOW_NOISE_WHITE = 0,
    RTOW_NOISE_BLUE = 1,              /* RT/BlueNoise.cs: texels of the set given to rtowUploadBlueNoise */
    RTOW_NOISE_SPATIOTEMPORAL_BLUE = 2 /* RT/SpatioTemporalBlueNoise.cs: the five texture sets given to rtowUploadStbNoise */
} RtowNoiseColor;

/* ---- scene description: flat, index-based PODs ----
 * The reference links BvhNode* / Entity* / Material* / void* Content by host pointer
 * (RT/BvhNode.cs:8-9, RT/Entity.cs:35-37); that graph cannot cross to a device, so the
 * boundary takes the same information as flat arrays with indices. */

/* RT/Texture.cs:23-49 (Type, MainColor, Parameter, ScalarValueChannel; ImageSize / ImagePointer / PixelStride through imageIndex). */
typedef struct RtowTexture {
    int32_t type;               /* RtowTextureType */
    RtowFloat3 mainColor;       /* Texture.MainColor */
    float parameter;            /* Texture.Parameter (ConstantValue for ConstantScalar) */
    int32_t scalarValueChannel; /* Texture.ScalarValueChannel (0..2) */
    int32_t imageIndex;         /* RTOW_TEXTURE_IMAGE: index into RtowSceneDesc.images; < 0 = null ImagePointer (samples as 0, :82-83) */
} RtowTexture;

/* The pixel data an Image texture points at (Texture.ImagePointer / ImageSize / PixelStride, RT/Texture.cs:28-30): rows of `width`
 * pixels, `pixelStride` bytes each, channels 0..2 = r, g, b (one byte each).  Point sampled at (int2)(uv * ImageSize) (:85-88);
 * the reference reads out of bounds for uv outside [0, 1) - here the texel coordinate is clamped into the image. */
typedef struct RtowImage {
    int32_t width, height;
    int32_t pixelStride;
    const uint8_t* pixels;      /* host memory, copied by rtowUploadScene */
} RtowImage;

/* RT/Material.cs:16-47.  `parameter` is IndexOfRefraction (Dielectric) or Density (Volume);
 * the reference ctor leaves it 0 for Standard (Material.cs:36-45). */
typedef struct RtowMaterial {
    int32_t type;               /* RtowMaterialType */
    RtowTexture albedo, glossiness, emission, metallic;
    float parameter;
} RtowMaterial;

/* RT/Entity.cs:27-56 + the Content struct it points to (RT/EntityTypes/Sphere.cs:6-24 ...).
 * `size`: Sphere -> (radius, -, -)  [signed radius, Sphere.cs:8-14]
 *         Rect   -> (sizeX, sizeY, -)   [RT/EntityTypes/Rect.cs:12-16: From = -size/2, To = size/2]
 *         Box    -> (sizeX, sizeY, sizeZ) [RT/EntityTypes/Box.cs:11-15: Extents = size/2]
 *         Triangle -> `contentIndex` selects RtowSceneDesc.triangles[contentIndex]. */
typedef struct RtowEntity {
    int32_t type;                   /* RtowEntityType */
    int32_t moving;                 /* Entity.Moving (0/1) */
    RtowFloat4 rotation;            /* Entity.OriginTransform.rot */
    RtowFloat3 position;            /* Entity.OriginTransform.pos */
    RtowFloat3 destinationOffset;   /* Entity.DestinationOffset */
    RtowFloat2 timeRange;           /* Entity.TimeRange */
    int32_t materialIndex;          /* index into RtowSceneDesc.materials (Entity.Material) */
    RtowFloat3 size;                /* content parameters, see above */
    int32_t contentIndex;           /* triangle payload index (RTOW_ENTITY_TRIANGLE only) */
} RtowEntity;

/* RT/EntityTypes/Triangle.cs:8-12, byte for byte (float3x3 Data, float3x3 Normals, float2x3 TextureCoordinates = 96 bytes),
 * so the host can pass its NativeList<Triangle> buffer (UNITY/Raytracer.cs:1198,1290-1300) without conversion.
 * data[0] = v2 - v0, data[1] = v1 - v0, data[2] = v0 (world space: triangles are never transformed, RT/Entity.cs:91-93). */
typedef struct RtowTriangle {
    RtowFloat3 data[3];
    RtowFloat3 normals[3];
    RtowFloat2 textureCoordinates[3];
} RtowTriangle;

typedef struct RtowSceneDesc {
    const RtowEntity* entities;
    int32_t entityCount;
    const RtowMaterial* materials;
    int32_t materialCount;
    int32_t maxBvhDepth;            /* the host's MaxBvhDepth (UNITY/Raytracer.cs:88, prefab default 32; 0 = that default).  The library builds
                                       its own tree to its own depth; this value only reproduces the order in which the host's tree
                                       would enumerate hits at identical distances (leaves are forced at that depth) */
    const RtowTriangle* triangles;  /* payloads of RTOW_ENTITY_TRIANGLE entities (RtowEntity.contentIndex); may be NULL */
    int32_t triangleCount;
    const RtowImage* images;        /* pixel data of RTOW_TEXTURE_IMAGE textures (RtowTexture.imageIndex); may be NULL */
    int32_t imageCount;
} RtowSceneDesc;

typedef struct RtowSceneInfo {
    int32_t entityCount;
    int32_t materialCount;
    int32_t bvhNodeCount;           /* inner nodes of the native BVH */
    int32_t bvhDepth;               /* max leaf depth == traversal-stack bound */
    int32_t ldsBytesScene;          /* bytes of scene data staged into LDS per workgroup */
    int32_t sceneInLds;             /* 1 if the whole scene is LDS resident, 0 if only the top levels */
    uint64_t sceneBytesDevice;      /* bytes of scene data resident in HBM */
    uint64_t hitSpillBytes;         /* bytes of HBM this scene reserves for hit lists beyond the 24 entries a lane holds itself (scenes with
                                       ProbabilisticVolume materials, exact-tie kernels): 16 B x 1024 lanes x CUs x (list capacity - 24) entries,
                                       0 where no ray can need it.  Grow-only per context; RtowContextOptions.hitListCapacity sizes it */
    int32_t hitListCapacity;        /* most surfaces one ray may meet in this scene before the batch reports RTOW_ERROR_CAPACITY (0: only the nearest hit is kept; for sphere scenes
                                       whose rare nearest-hit ties are settled by the fix-up pass: the capacity of that pass's lists).  A launch also reports RTOW_ERROR_CAPACITY - final,
                                       this value unchanged - when it marks more than 2^20 pixel-batches for the fix-up pass: a scene of coinciding spheres that should run with
                                       RTOW_CONTEXT_EXACT_TIES_ALWAYS */
    int32_t wideCodes;              /* 1: more than 65 535 entities or tree nodes - the kernels that keep 32-bit candidate / stack codes run (tree read from HBM) */
    int32_t thresholdSet;           /* stage thresholds in use for this scene: -1 the built-in ones of its kernel kind (nothing measured yet - the probes of a measurement may be in
                                       flight -, too few samples asked for so far, tuning off, or RtowContextOptions.schedulerTune given); 0 / 1 / 2 the sphere family / the general family /
                                       the general family with REGEN and SKY from 1/8, as measured (or as measured earlier for a like scene); 3 / 4 / 5 the same with the volume stage waiting too */
    int32_t schedulerTune[9];       /* the values themselves (REGEN TRAV TEST HIT SKY VOL | hand-over count | pixel regrouping | walk slice), as RtowContextOptions.schedulerTune would set them */
} RtowSceneInfo;

/* ---- the operator's parameter block: SampleBatchJob's public fields (JOBS/SampleBatchJob.cs:23-51) ---- */

/* RT/View.cs:8-14 (7 x float3 + float = 88 bytes). Built on the host by View's ctor (View.cs:16-36). */
typedef struct RtowView {
    RtowFloat3 origin;
    RtowFloat3 lowerLeftCorner;
    RtowFloat3 horizontal, vertical;
    RtowFloat3 forward, up, right;
    float lensRadius;
} RtowView;

/* RT/Texture.cs:141-211 `Cubemap`: the six faces of the sky cube as the host's own pixel data.  The reference keeps a pointer to
 * the face +X of a Unity cubemap and reaches the others by `faceStride`, i.e. it relies on the faces being contiguous in the
 * order +X, -X, +Y, -Y, +Z, -Z (Unity's CubemapFace order), each face `faceHeight` rows of `faceWidth` pixels, row 0 first.
 * channelType follows the reference's ChannelType: its ctor only accepts R16G16B16A16_SFloat (SignedHalf, pixelStride 8), its
 * Sample() also decodes UnsignedByte (value / 255). */
/* How white-noise generators are assigned (the texture-driven colours are per pixel by construction).
 *   RTOW_RNG_REFERENCE   the reference: one Unity.Mathematics.Random per pixel per batch, seeded from (Seed, pixel index), running
 *                        through all the pixel's samples (JOBS/SampleBatchJob.cs:91,132-157).  Same seed, same image as the reference.
 *   RTOW_RNG_PER_SAMPLE  NOT the reference's stream (a different, statistically equivalent image): sample s of pixel i gets its own
 *                        generator, seeded ((Seed * 0x8C4CA03F) ^ (i * 0x7383ED49)) ^ ((s + 1) * 0x9E3779B9) (0x9E3779B9 if that is 0)
 *                        and advanced once like the Random ctor does; samples are taken in groups of 16, every group summed from zero
 *                        in sample order, and the groups added to the accumulators in group order (colour, normal, albedo, successes,
 *                        sample-count weight; the fallback AOVs are sample 0's).  A pixel's samples become independent units of work:
 *                        no long tail at the end of a batch, and a frame can be split over GPUs by rows without starving lanes.
 *   RTOW_RNG_PER_SAMPLE_XOROSHIRO  the same policy (units, groups, fold order) with the generator north_star names: xoroshiro64** (Blackman & Vigna,
 *                        two 32-bit words per lane).  s0 = the RTOW_RNG_PER_SAMPLE seed of the sample, s1 = s0 * 0x85EBCA6B ^ 0xC2B2AE35 (0x9E3779B9 if both
 *                        are 0), one output discarded; NextFloat() takes the top 23 bits of an output exactly like Unity's (asfloat(0x3f800000 | (x >> 9)) - 1). */
typedef enum RtowRngPolicy { RTOW_RNG_REFERENCE = 0, RTOW_RNG_PER_SAMPLE = 1, RTOW_RNG_PER_SAMPLE_XOROSHIRO = 2 } RtowRngPolicy;

typedef enum RtowCubemapChannelType { RTOW_CUBEMAP_UNSIGNED_BYTE = 0, RTOW_CUBEMAP_SIGNED_HALF = 1 } RtowCubemapChannelType;
typedef struct RtowCubemapDesc {
    int32_t faceWidth, faceHeight;
    int32_t channelType;            /* RtowCubemapChannelType */
    int32_t pixelStride;            /* bytes from one pixel to the next (8 for RGBA half, 3 or 4 for bytes); channels 0..2 = r, g, b */
    const void* faces;              /* 6 * faceWidth * faceHeight * pixelStride bytes, host memory, copied by the call */
} RtowCubemapDesc;

/* The host's noise textures (UNITY/BlueNoiseData.cs:19-57, UNITY/SpatioTemporalBlueNoiseData.cs:18-44): square textures of
 * `rowStride` x `rowStride` texels, `textureCount` of them back to back; a batch reads ONE of them, RtowSampleParams.noiseTextureIndex
 * (the host's textureIndex after CycleTexture(), UNITY/Raytracer.cs:658-659).  Texel formats are the ones the reference reads:
 * blue = half4 (8 bytes, .x / .xy used, RT/BlueNoise.cs:26-28); STBN scalar = 1 byte, vector2 / unitVector2 / unitVector3 = RGB24,
 * cosineUnitVector3 = RGBA32 (RT/SpatioTemporalBlueNoise.cs:61-85).  Host memory, copied by the call; NULL desc drops the set. */
typedef struct RtowBlueNoiseDesc {
    uint32_t rowStride, textureCount;
    const void* texels;
} RtowBlueNoiseDesc;
typedef struct RtowStbNoiseDesc {
    uint32_t rowStride, textureCount;
    const void* scalar;
    const void* vector2;
    const void* cosineUnitVector3;
    const void* unitVector2;
    const void* unitVector3;
} RtowStbNoiseDesc;

/* RT/Environment.cs:12-17; the cubemap handle of SkyType.CubeMap is the one uploaded with rtowUploadSkyCubemap. */
typedef struct RtowEnvironment {
    int32_t skyType;                /* RtowSkyType */
    RtowFloat3 skyBottomColor;
    RtowFloat3 skyTopColor;
} RtowEnvironment;

typedef struct RtowSampleParams {
    RtowFloat2 size;                /* SampleBatchJob.Size (float2; coordinates use (int)Size.x, :64-67) */
    int32_t sliceOffset;            /* :26  rows with (row % sliceDivider) != sliceOffset are skipped (:69-70) */
    int32_t sliceDivider;           /* :27  (>= 1) */
    uint32_t seed;                  /* :28 */
    RtowView view;                  /* :29 */
    RtowEnvironment environment;    /* :30 */
    uint32_t sampleCountRange[2];   /* :31  uint2 (x = min, y = max per batch) */
    int32_t traceDepth;             /* :32 */
    int32_t subPixelJitter;         /* :33  bool */
    int32_t noiseColor;             /* :38  RtowNoiseColor */
    RtowFloat2 sampleCountWeightExtrema; /* :39 */
    int32_t diagnosticsStride;      /* bytes per pixel of the diagnostics buffer: 4 = {RayCount},
                                       16 = FULL_DIAGNOSTICS {RayCount, BoundsHitCount, CandidateCount,
                                       SampleCountWeight} (UNITY/Raytracer.cs:54-64) */
    int32_t noiseTextureIndex;      /* which texture of the uploaded blue / STBN set this batch reads (ignored for white noise) */
    int32_t rngPolicy;              /* RtowRngPolicy; 0 = the reference's stream */
} RtowSampleParams;

/* The four accumulation buffers (JOBS/SampleBatchJob.cs:41-49): W*H elements each, tightly packed,
 * pixel index = row * W + col with row 0 at the BOTTOM of the image (:64-67, RT/View.cs:44-46).
 * color.xyz is the colour SUM, color.w the successful-sample COUNT as a float (:72-78,159). */
typedef struct RtowAccumBuffers {
    float* color;               /* float4[W*H] */
    float* normal;              /* float3[W*H] */
    float* albedo;              /* float3[W*H] */
    float* sampleCountWeight;   /* float [W*H] */
} RtowAccumBuffers;

/* ---- context ---- */
typedef struct RtowContext_t* RtowContext;

/* level: 0 disable, 1 fatal, 2 error, 3 warning, 4 print (OptixLogLevel, OptixApi.cs:80-87).
 * May be invoked from any thread (OptixApi.cs:145-152). */
typedef void (*RtowLogCallback)(int32_t level, const char* tag, const char* message, void* userData);

/* Behaviour switches of a context (RtowContextOptions.flags).  None of them changes what the library computes for a given scene and
 * parameter block except where stated; nothing is read from the environment. */
typedef enum RtowContextFlags {
    RTOW_CONTEXT_EXACT_TIES_ALWAYS = 1u << 0,      /* settle every nearest-hit tie with the reference's whole procedure (walk again unpruned, sort the hit list like
                                                    * NativeSortExtension.Sort, take [0]) INSIDE the sample kernel in every scene without volumes; default: in scenes that hold the
                                                    * same primitive twice and in scenes of more than 16 entities with any rect, box or rotated entity.  Sphere-only scenes and
                                                    * all-triangle scenes (one entity per mesh triangle: what the live host makes) run the rank-rule kernels and have the rare
                                                    * pixel that meets two different surfaces at bit-identical distance rendered again by the exact kernel in a fix-up pass after
                                                    * the launch; a triangle scene that marks thousands of pixels in one launch moves to the exact kernels for
                                                    * good: the same results either way */
    RTOW_CONTEXT_EXACT_TIES_NEVER = 1u << 1,       /* never (the rank rule everywhere, no fix-up pass: exact for rays of at most 16 hits) */
    RTOW_CONTEXT_REFERENCE_DIAGNOSTICS = 1u << 2,  /* FULL_DIAGNOSTICS records (diagnosticsStride 16): BoundsHitCount / CandidateCount count the REFERENCE's tree -
                                                    * node boxes a ray passes and entities of the leaves it reaches in the tree RebuildBvh would build
                                                    * (UNITY/BvhNodeData.cs:122-213, JOBS/SampleBatchJob.cs:427-440, UNITY/Raytracer.cs:54-64) - instead of the
                                                    * library's own tree.  Costs a second, unpruned walk per ray in such batches */
    RTOW_CONTEXT_NO_CAMERA_RAY_LISTS = 1u << 3,    /* development: walk the tree for camera rays too */
    RTOW_CONTEXT_NO_CHUNK_ORDER = 1u << 4,         /* development: hand out pixel chunks in row order, not most-expensive-first */
    RTOW_CONTEXT_FORCE_WIDE_CODES = 1u << 5,       /* development: run the scene through the kernels with 32-bit candidate / stack codes (every scene kind has them)
                                                    * (what scenes beyond 65 535 entities or tree nodes use; the tree is then read from HBM) */
    RTOW_CONTEXT_NO_CHAIN_FUSION = 1u << 7,        /* run chained batches (rtowSampleBatchChain*) one launch per batch - what a context does by itself when its same-XCD hand-over
                                                    * litmus fails (rtowCreateContext measures, on the device it runs on, that plain stores + sc1 loads hand data over inside an XCD
                                                    * the way a chained launch relies on; logged at level 4, a failure at level 3).  Same results either way */
    RTOW_CONTEXT_NO_THRESHOLD_TUNING = 1u << 6     /* keep the built-in stage thresholds of the scene's kernel kind.  By default a scene that has been asked for 64 samples per pixel
                                                    * since its upload gets three (volume scenes: six) threshold sets measured with 4-sample probes of the batch's own frame, enqueued in front
                                                    * of that batch and timed with events that a LATER call reads - no call waits for them - and the fastest set runs from then on; a re-upload of
                                                    * a scene of the same kind and size reuses what was measured.  Thresholds are pure scheduling and never change a result */
} RtowContextFlags;

typedef struct RtowContextOptions {
    int32_t deviceOrdinal;          /* HIP device ordinal */
    RtowLogCallback logCallback;    /* may be NULL */
    void* logCallbackData;
    int32_t logCallbackLevel;
    uint32_t flags;                 /* RtowContextFlags, 0 = defaults */
    int32_t ldsSceneBudgetBytes;    /* development: cap on the bytes of scene image staged into LDS (0 = all that fits); smaller scenes then run
                                     * through the kernels that read the tree from HBM */
    int32_t schedulerTune[9];       /* development: stage thresholds in 64ths of the live lanes (REGEN TRAV TEST HIT SKY VOL; values below 1 mean 1 = any lane), the number of
                                     * candidates at which a box walk hands over to the exact tests (1 .. 7; 0 = the built-in 3), the pixel regrouping (below), and the box-walk slice (node visits
                                     * per trip; 0 = the built-in value of the scene); all zero = everything built in, thresholds measured per scene.
                                     * [7], which pixel a ticket stands for, is a knob of its own (setting it says nothing about the thresholds): 0 = the default (3); 1 = a wave's 64 tickets are
                                     * an 8 x 8 tile of the image in row order; 3 = the same tile, its tickets ordered most expensive pixel first by the ray counts of the previous launch
                                     * (a wave's lanes take a chunk's tickets one by one as they finish their previous pixels, so a chunk's expensive pixel should not be its last ticket:
                                     * +0.7 ... 1.3 %); n + 16 * mode with n = 2 / 4 / 8: inside super-tiles of n x n tiles the pixels are sorted by ray count (mode 0), or by class only -
                                     * sky / not sky (mode 1), four classes of rays per sample (mode 2) - keeping their tile order inside a class, and dealt out 64 at a time (0 ... -5 %:
                                     * measured, not used).  Re-sorted behind every launch.  + 256 * K (K = 1 .. 15): a wave of a batch GROUP reserves K (chunk, batch) queue slots
                                     * at a time instead of the built-in 4 (it then works through one tile's batches, and its neighbours' in the cost order, one after the other: groups
                                     * +1.8 % at 4; A/B runs).  Scheduling only, like everything here: results do not change (the reference hands
                                     * pixels out in no defined order, UNITY/Raytracer.cs:730) */
    int32_t hitListCapacity;        /* most surfaces one ray may meet where every hit of a ray is kept (scenes with ProbabilisticVolume materials, and
                                     * the exact-tie procedure): the reference's hitRecordBuffer grows on the heap (UTIL/HybridCollections.cs:65-71);
                                     * here a lane holds 24 hits itself and longer lists continue in device memory, 16 bytes x 262 144 lanes per
                                     * entry, sized at rtowUploadScene to min(this, the most the scene can produce: 2 per entity with volumes, else 1).
                                     * A ray beyond it makes the batch report RTOW_ERROR_CAPACITY, always: the caller has chosen the memory it spends.
                                     * 0 = the lists grow like the reference's: they start at 1024 entries in scenes with volumes, 128 elsewhere, and every batch
                                     * that meets a longer ray doubles them (up to what the scene can produce, and to a quarter of the free device memory) when its
                                     * status is read.  rtowSampleBatch / rtowSampleBatchChain read it themselves and run the batch again from the caller's inputs
                                     * (unless outputs registered with rtowRegisterHostBuffer ARE the inputs); the device-resident forms report RTOW_ERROR_CAPACITY
                                     * once through rtowGetBatchStatus / rtowSynchronize and the same call, issued again from inputs it did not overwrite, has room (if
                                     * rtowGetSceneInfo.hitListCapacity did not change across the error the lists could not grow - device memory - and the error is final).
                                     * The capacity a context has grown to stays across rtowUploadScene (rtowGetSceneInfo.hitListCapacity shows it) */
    int32_t sliceBlockThreads;      /* reserved: 0 (or 1024).  Rounds 2 - 3 could run 512 / 256 lanes per workgroup for launches that own about one pixel per
                                     * resident lane; measured slower at every slice count and removed.  Other values: RTOW_ERROR_INVALID_VALUE */
} RtowContextOptions;

RTOW_API int rtowGetApiVersion(void);
RTOW_API const char* rtowErrorString(int result);

/* replaces: nothing in the Burst path (in-process job); modelled on createDeviceContext / destroyDeviceContext
 * (OptixDenoiser.h:21-25). Fails with RTOW_ERROR_NO_DEVICE when no HIP device is usable: there is NO CPU fallback. */
RTOW_API int rtowCreateContext(const RtowContextOptions* options, RtowContext* outContext);
RTOW_API int rtowDestroyContext(RtowContext context);

/* replaces: the tail of Raytracer.RebuildWorld (UNITY/Raytracer.cs:1167-1183): RebuildEntityBuffers' Entity/Material
 * buffers (:1185-1304) and RebuildBvh (:1306-1351, UNITY/BvhNodeData.cs:122-213, JOBS/BuildRuntimeBvhJob.cs:20-39).
 * Copies the description, builds the native BVH, uploads the flat GPU layout. Called only when the world changes. */
RTOW_API int rtowUploadScene(RtowContext context, const RtowSceneDesc* scene);
RTOW_API int rtowGetSceneInfo(RtowContext context, RtowSceneInfo* outInfo);

/* replaces: `new Cubemap(skyCubemap)` when the environment is built (UNITY/Raytracer.cs, RT/Texture.cs:150-169): copies the six
 * faces to the device; sample batches whose environment.skyType is RTOW_SKY_CUBEMAP then evaluate Cubemap.Sample(ray.Direction)
 * (RT/Texture.cs:171-210, JOBS/SampleBatchJob.cs:356-358).  NULL `cubemap` (or NULL faces) drops the current one: like the
 * reference's null data pointer, sampling then yields black. */
RTOW_API int rtowUploadSkyCubemap(RtowContext context, const RtowCubemapDesc* cubemap);

/* replaces: blueNoise.GetRuntimeData(frameSeed) / stbNoise.GetRuntimeData(frameSeed) (UNITY/Raytracer.cs:702-703): the texture sets the
 * BlueNoise / SpatioTemporalBlueNoise samplers walk (RT/PerPixelNoise.cs).  A batch with noiseColor Blue / SpatioTemporalBlue and no
 * uploaded set fails with RTOW_ERROR_INVALID_VALUE. */
RTOW_API int rtowUploadBlueNoise(RtowContext context, const RtowBlueNoiseDesc* noise);
RTOW_API int rtowUploadStbNoise(RtowContext context, const RtowStbNoiseDesc* noise);

/* replaces: sampleBatchJob.Schedule(totalBufferSize, 1, ...) bracketed by RecordTimeJob 0/1
 * (UNITY/Raytracer.cs:729-738) == SampleBatchJob.Execute for every pixel index (JOBS/SampleBatchJob.cs:59-164).
 * Host-buffer form: uploads `in`, runs the kernel, downloads `out` + diagnostics; blocks until done, like IJob.Execute().
 * `diagnostics` has W*H records of params->diagnosticsStride bytes (may be NULL).
 * `cancel` (may be NULL) is polled while the kernel runs (JOBS/SampleBatchJob.cs:61-62); when it becomes non-zero
 * the call returns RTOW_ERROR_CANCELLED and `out` is unspecified (the host discards it, UNITY/Raytracer.cs:489-516).
 * Pixels skipped by the slice test are NOT written (the host pre-copies them, UNITY/Raytracer.cs:719-726). */
RTOW_API int rtowSampleBatch(RtowContext context, const RtowSampleParams* params,
                             const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                             void* diagnostics, const volatile uint8_t* cancel);

/* Same operator with DEVICE-resident buffers (pointers from rtowDeviceAlloc or any HIP allocation in this process),
 * enqueued on `stream` (a hipStream_t, NULL = the context's own stream). Does not block unless `cancel` is non-NULL.
 * `in` and `out` may alias element-for-element (each pixel is read before it is written by the same lane). */
RTOW_API int rtowSampleBatchDevice(RtowContext context, const RtowSampleParams* params,
                                   const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                   void* diagnostics, void* stream, const volatile uint8_t* cancel);

/* `count` successive batches of ONE frame, enqueued together: exactly what
 *     rtowSampleBatchDevice(params[0], in, out, diagnostics[0]); rtowSampleBatchDevice(params[k], out, out, diagnostics[k]) for k = 1 .. count-1
 * computes, bit for bit - the reference keeps two such batches in flight, the second scheduled with a dependency on the first
 * (UNITY/Raytracer.cs:586-593 "kick if needed (with double-buffering)", :798-811).  When the batches differ in nothing but `seed`
 * (successive batches of a frame, :656-661) they run as ONE launch in which every 64-pixel chunk of batch k + 1 starts as soon as that chunk
 * of batch k is stored, so the CUs that run out of batch-k pixels do not idle until its slowest pixel ends (the reference stream makes a
 * pixel's samples one indivisible unit of work; a batch on its own ends with a tail of about one pixel-time).  `diagnostics`: `count`
 * device pointers (or NULL), one record buffer per batch.  `cancel` as for rtowSampleBatchDevice.  rtowGetLastSampleKernelMs then
 * reports the last launch (all its batches). */
RTOW_API int rtowSampleBatchChainDevice(RtowContext context, int32_t count, const RtowSampleParams* params /* [count] */,
                                        const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                        void* const* diagnostics /* [count] or NULL */, void* stream, const volatile uint8_t* cancel);

/* `count` INDEPENDENT batches of one frame, enqueued together: exactly what
 *     rtowSampleBatchDevice(params[k], in, &outs[k], diagnostics[k])   for k = 0 .. count-1
 * computes, bit for bit - every batch reads the same `in` and stores to its own outs[k] (count > 1: no output may share a buffer with another output
 * or with `in`).  When the batches differ in nothing but `seed` they run as ONE launch whose work queue holds (pixel chunk, batch) pairs, so the launch
 * ends when its slowest PIXEL-BATCH does, not `count` of them one after the other: under the reference's random stream a pixel's samples are one
 * sequential unit of work, and a chain (above) adds the pixel's successive batches to that sequence - at the reference host's own default
 * traceDepth 32 the cover scene's slowest pixel (6 429 path segments per 256 samples, ~13 us each) holds a batch for 87 ms where the machine's
 * throughput needs 57 (profiles/r04g_ray_count_stats.txt).  What this is for: the sub-batches of a tiles x batches partition (below; a rank renders the
 * sub-batches of several steps, each from zeroed inputs, in one launch), and hosts that accept the reference's accumulation up to the association of
 * its float sums (render B batches from zero, fold them in order with rtowAddAccumDevice).  Up to 16 batches per launch. */
RTOW_API int rtowSampleBatchGroupDevice(RtowContext context, int32_t count, const RtowSampleParams* params /* [count] */,
                                        const RtowAccumBuffers* in, const RtowAccumBuffers* outs /* [count] */,
                                        void* const* diagnostics /* [count] or NULL */, void* stream, const volatile uint8_t* cancel);

/* The same chain with HOST buffers, blocking like rtowSampleBatch: the inputs travel once, the batches accumulate in place on the device, the
 * final accumulators (rows this slice owns) and each batch's diagnostics travel back.  All batches share size, slice and diagnosticsStride.
 * For hosts that keep their accumulators in NativeArrays and have more than one batch queued. */
RTOW_API int rtowSampleBatchChain(RtowContext context, int32_t count, const RtowSampleParams* params /* [count] */,
                                  const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                  void* const* diagnostics /* [count] host pointers or NULL */, const volatile uint8_t* cancel);

/* Pinned host buffers for rtowSampleBatch.  The host's accumulation buffers are long-lived pools (UNITY/Raytracer.cs:279-288,
 * Allocator.Persistent); registering them once (hipHostRegister) lets rtowSampleBatch move the inputs with one pinned DMA and lets the
 * kernel store the outputs and diagnostics straight into host memory, so the host-buffer form runs at the speed of the device-resident
 * one.  Unregistered pointers keep working (pageable staging copies).  A registration covers [pointer, pointer + sizeInBytes) and must
 * be dropped with rtowUnregisterHostBuffer before the memory is freed.  The reference's own plugin pattern is H2D -> work -> D2H per call
 * (JOBS/DenoiseJobs.cs:75-117); this is that pattern with the copies taken off the critical path. */
RTOW_API int rtowRegisterHostBuffer(RtowContext context, void* pointer, size_t sizeInBytes);
RTOW_API int rtowUnregisterHostBuffer(RtowContext context, void* pointer);

/* One ray against the resident scene: the nearest Entity.Hit (tMin 0, tMax +inf) along origin + t * direction at ray time `time`.
 * replaces: HitWorld (UNITY/Raytracer.cs:1353: BvhRoot->Hit(r, 0, float.PositiveInfinity, out hitRec) -> the recursive HitTests.Hit(BvhNode),
 * RT/HitTests.cs:152-196), which ScheduleSample calls with the camera's centre ray before EVERY batch to set the focus distance
 * (UNITY/Raytracer.cs:608-609: `if (HitWorld(new Ray(origin, forward), out hitRec)) focusDistance = hitRec.Distance;` - ray time 0).  With this call the host
 * no longer needs a tree of its own, i.e. its serial RebuildBvh (UNITY/Raytracer.cs:1306-1351), once the scene is uploaded.
 * *distance: hitRec.Distance, bit for bit (the path's own float program; +INFINITY on a miss).  *entityIndex: index of the hit entity in
 * RtowSceneDesc.entities, -1 on a miss; of several entities at the bit-identical nearest distance the one that comes first in the reference tree's
 * leaf order - the one the sample path shades (the reference's recursion prefers its right subtree on such a tie; the host reads the distance only).
 * Either pointer may be NULL.  Walked on the host through the tree rtowUploadScene built (its image stays in host memory): microseconds, no device work, so batches
 * in flight are neither waited for nor disturbed - the host calls this from ScheduleSample while the previous batch is still tracing (UNITY/Raytracer.cs:586-611), and a
 * launch could only start when that batch ends.  RTOW_ERROR_NO_SCENE before rtowUploadScene. */
RTOW_API int rtowProbeNearestHit(RtowContext context, const RtowFloat3* origin, const RtowFloat3* direction, float time, float* distance, int32_t* entityIndex);

/* Device time (ms) of the most recent sample kernel of this context, measured with HIP events recorded on the
 * stream the kernel was launched on (the RecordTimeJob 0/1 bracket, JOBS/UtilJobs.cs:77-86). Synchronises on the end event. */
RTOW_API int rtowGetLastSampleKernelMs(RtowContext context, float* outMs);

/* replaces: ReduceMetricsJob.Execute (JOBS/ReduceMetricsJob.cs:22-45). Device buffers in, host scalars out. */
typedef struct RtowMetrics {
    int32_t totalRayCount;              /* sum of (int)RayCount */
    int32_t totalSamples;               /* sum of (int)color.w */
    RtowFloat2 sampleCountWeightExtrema;/* min / max of sampleCountWeight / sampleCount */
    int32_t sampleCountExtrema[2];      /* min / max of (int)color.w */
    int64_t totalRayCount64;            /* same sums without the reference's int32 wrap */
    int64_t totalSamples64;
} RtowMetrics;
RTOW_API int rtowReduceMetricsDevice(RtowContext context, int32_t pixelCount, const void* diagnostics,
                                     int32_t diagnosticsStride, const float* color, const float* sampleCountWeight,
                                     void* stream, RtowMetrics* outMetrics);

/* The same reduction without the wait: the per-block partial results are folded on the device and the record is written to `outMetrics` in stream
 * order - memory the device can write: a range given to rtowRegisterHostBuffer, pinned host memory of this process (hipHostMalloc) or device memory; pageable
 * host memory is RTOW_ERROR_INVALID_VALUE.  The reference consumes these numbers a frame later (UNITY/Raytracer.cs:518-550): the host reads the record when the
 * batch's other results are in (rtowSynchronize / a stream event), and the call costs what its kernels cost. */
RTOW_API int rtowReduceMetricsDeviceAsync(RtowContext context, int32_t pixelCount, const void* diagnostics,
                                          int32_t diagnosticsStride, const float* color, const float* sampleCountWeight,
                                          void* stream, RtowMetrics* outMetrics);

/* replaces: CombineJob.Execute (JOBS/CombineJob.cs:29-71): sum -> mean, interlace look-around, NaN handling. */
typedef struct RtowCombineParams {
    int32_t width, height;          /* CombineJob.Size */
    int32_t debugMode;              /* CombineJob.DebugMode */
    int32_t ldrAlbedo;              /* CombineJob.LdrAlbedo */
} RtowCombineParams;
RTOW_API int rtowCombineDevice(RtowContext context, const RtowCombineParams* params,
                               const float* inColor /*float4*/, const float* inNormal, const float* inAlbedo,
                               float* outColor /*float3*/, float* outNormal, float* outAlbedo, void* stream);

/* replaces: FinalizeTexturesJob.Execute (JOBS/FinalizeTexturesJob.cs:23-55): gamma + RGBA32 pack. */
RTOW_API int rtowFinalizeDevice(RtowContext context, int32_t pixelCount,
                                const float* inColor /*float3*/, const float* inNormal, const float* inAlbedo,
                                uint8_t* outColor /*RGBA32*/, uint8_t* outNormal, uint8_t* outAlbedo, void* stream);

/* replaces: CombineJob -> FinalizeTexturesJob back to back, the reference's default chain (denoiseMode 0, Assets/Prefabs/Raytracer.prefab:391; UNITY/Raytracer.cs:806-807),
 * in ONE pass: the accumulators in, the three RGBA32 textures out - 40 B read + 12 B written per pixel instead of 80 + 48 through the float3 intermediates.  The same
 * bytes as rtowCombineDevice followed by rtowFinalizeDevice. */
RTOW_API int rtowCombineFinalizeDevice(RtowContext context, const RtowCombineParams* params,
                                       const float* inColor /*float4*/, const float* inNormal, const float* inAlbedo,
                                       uint8_t* outColor /*RGBA32*/, uint8_t* outNormal, uint8_t* outAlbedo, void* stream);

/* dst += src for the four accumulation buffers (device pointers, `pixelCount` elements each).  The reference accumulates
 * successive batches by feeding a batch's outputs to the next one as inputs (UNITY/Raytracer.cs:798-802); when batches run
 * CONCURRENTLY on several GPUs (each from zeroed accumulators, its own Seed) their partial sums are folded with this, in
 * a fixed order, on the root.  color.w (the success count) adds like the other channels. */
RTOW_API int rtowAddAccumDevice(RtowContext context, int32_t pixelCount, const RtowAccumBuffers* dst, const RtowAccumBuffers* src, void* stream);

/* replaces: allocateCudaBuffer / copyCudaBuffer / deallocateCudaBuffer (OptixDenoiser.h:57-64). kind: RtowMemcpyKind. */
typedef enum RtowMemcpyKind {       /* CudaMemcpyKind, OptixApi.cs:33-40 */
    RTOW_MEMCPY_HOST_TO_HOST = 0,
    RTOW_MEMCPY_HOST_TO_DEVICE = 1,
    RTOW_MEMCPY_DEVICE_TO_HOST = 2,
    RTOW_MEMCPY_DEVICE_TO_DEVICE = 3
} RtowMemcpyKind;
/* ---- multi-GPU: one process per GPU, the frame row-interleaved with the reference's own slice contract ----
 * Process g of G renders with sliceOffset = g, sliceDivider = G (JOBS/SampleBatchJob.cs:69-70: rows with row % G == g) and owns exactly
 * those rows of every buffer; pixels are independent given (params, scene, Seed, global pixel index), so the gathered frame is
 * bit-identical to the single-GPU frame.  The ONE collective of a batch is a gather of the owned rows to a root over RCCL (xGMI: every
 * peer has its own link to the root).  RCCL is loaded on first use (dlopen of librccl.so), so single-GPU hosts never touch it.
 *   rank 0:  rtowCommGetUniqueId(&id), hand the 128 bytes to the other processes by any host channel (the host's own IPC, a file, MPI ...)
 *   all:     rtowCommInit(ctx, &id, rank, worldSize)            collective, blocks until every rank has joined
 *   batch:   rtowSampleBatchDevice(...slice params...); rtowGatherRowsDevice(ctx, W, H, G, &mine, &frame, what, root, stream)
 *   all:     rtowCommDestroy(ctx)
 * rtowGatherRowsDevice: `mine` holds this rank's owned rows in place (full-frame buffers, other rows ignored); on the root `frame`
 * receives every rank's rows in place (full-frame buffers; may be the same buffers as `mine`); on other ranks `frame` is ignored.
 * `what` selects the buffers (RtowGatherMask); rows are packed, sent and unpacked on `stream` (NULL = the context's own), asynchronously. */
typedef struct RtowCommId { char bytes[128]; } RtowCommId;        /* ncclUniqueId */
/* Which RCCL build rtowComm* loads.  NULL (the default): the copy this process already holds, else librccl.so.1 / librccl.so on the loader
 * path, else /opt/rocm/lib.  A host that ships its own ROCm names the file here - before the first rtowComm* call of the process (afterwards:
 * RTOW_ERROR_INVALID_VALUE; the library is loaded once).  tests/ point it at a stand-in transport (tests/native/fake_rccl.cpp: the same eight
 * nccl* entry points over shared memory) so that the multi-rank gather runs on a box with one GPU, which RCCL itself refuses. */
RTOW_API int rtowCommSetLibraryPath(const char* path);
typedef enum RtowGatherMask {
    RTOW_GATHER_COLOR = 1, RTOW_GATHER_NORMAL = 2, RTOW_GATHER_ALBEDO = 4, RTOW_GATHER_SAMPLE_COUNT_WEIGHT = 8, RTOW_GATHER_ALL = 15,
    RTOW_GATHER_NO_BATCH_WAIT = 16,  /* rtowGatherRowsDevice / rtowExchangeAccumDevice normally start after the context's most recent sample batch, whatever
                                      * stream that batch was given (its rows are what travels).  With this bit the call is ordered by `stream` alone: for rows
                                      * that do not come out of that batch - the output of rtowExchangeAccumDevice on the same stream, or a partial result
                                      * whose batch the caller has already ordered before `stream` itself - while the NEXT batch is already enqueued */
    RTOW_GATHER_LOOPBACK = 32        /* rtowGatherRowsDevice on a communicator of ONE rank (rtowCommInit(ctx, id, 0, 1)): send the rows to itself THROUGH the transport - pack,
                                      * ncclSend + ncclRecv to / from rank 0 in one group, scatter - instead of copying them.  A deployment check: on a box with a single GPU
                                      * (RCCL refuses two ranks on one device) it runs the loaded library's ncclGetUniqueId / ncclCommInitRank / ncclGroupStart / ncclSend /
                                      * ncclRecv / ncclGroupEnd / ncclCommDestroy with this library's own argument conventions (the 128-byte id by value, ncclFloat32, counts in
                                      * elements, the caller's stream).  Ignored by communicators of more than one rank */
} RtowGatherMask;
RTOW_API int rtowCommGetUniqueId(RtowCommId* outId);
RTOW_API int rtowCommInit(RtowContext context, const RtowCommId* id, int32_t rank, int32_t worldSize);
RTOW_API int rtowCommDestroy(RtowContext context);
RTOW_API int rtowGatherRowsDevice(RtowContext context, int32_t width, int32_t height, int32_t sliceDivider,
                                  const RtowAccumBuffers* mine, const RtowAccumBuffers* frame, int32_t what, int32_t root, void* stream);

/* ---- multi-GPU, tiles x batches: G ranks = T row slices x B seed groups (G = T * B) ----
 * The tile partition alone stops scaling where a GPU owns about one pixel per resident lane: under the reference's random stream a pixel's
 * samples are one sequential unit of work (JOBS/SampleBatchJob.cs:91,132-157), so the slowest pixel bounds the batch.  What the reference itself
 * splits a frame's samples into is BATCHES: successive batches with a fresh Seed (frameSeed, UNITY/Raytracer.cs:656-661), each feeding its sums to
 * the next (:798-802).  Batches are independent of each other, so B of them can run at the same time: rank r = tile + T * group renders the rows
 * of slice `tile` of T (the reference's slice fields) with its share of the batch's samples and the Seed of batch `group`, FROM ZEROED accumulators
 * (a partial sum); afterwards the B ranks of a tile exchange their partial sums so that every row is folded on ONE rank -
 *     accum[row] += partial_0[row]; accum[row] += partial_1[row]; ... += partial_{B-1}[row]        (group order; row % G == rank)
 * - the same float additions in the same order wherever the row is folded, so the frame is reproducible bit for bit, and equal to the
 * reference's sequential accumulation of those B batches up to the association of the float sums (each partial is summed from zero instead
 * of on top of its predecessor: <= 1e-4 of the mean at the sample counts of the benchmark configurations, asserted in tests/).  The folded rows are
 * then gathered like a tile partition's: rtowGatherRowsDevice(..., sliceDivider = G, mine = accum, ...).
 *   plan:   rtowHybridPlan(G, rank, T, samplesPerBatch, step, &plan)        pure arithmetic: this rank's slice fields, sample share and Seed
 *   batch:  rtowSampleBatchDevice(params{sliceOffset, sliceDivider, seed, sampleCountRange = plan}, in = zeros, out = partial)
 *           rtowExchangeAccumDevice(ctx, W, H, T, &partial, &accum, what, stream)      all ranks; one grouped ncclSend / ncclRecv per peer of the tile
 *           rtowGatherRowsDevice(ctx, W, H, G, &accum, &frame, what | RTOW_GATHER_NO_BATCH_WAIT, root, stream)
 * T = 1 splits only the samples (every rank renders the whole frame), T = G only the rows (then the exchange folds a rank's own rows and nothing
 * travels).  `partial` and `accum` are full-frame buffers that must not overlap; rows outside a rank's tile are never read, rows it does not
 * fold are never written.  Traffic per rank and batch: (B - 1) / B of its tile's rows out and as much in, spread over its B - 1 direct xGMI links. */
typedef struct RtowHybridPlan {
    int32_t tileCount, groupCount;      /* T, B = worldSize / T */
    int32_t tile, group;                /* of this rank: rank % T, rank / T */
    int32_t sliceOffset, sliceDivider;  /* what this rank renders with: (tile, T) */
    uint32_t samples;                   /* this rank's share of samplesPerBatch: samplesPerBatch / B, the remainder to the low groups */
    uint32_t seed;                      /* Seed of this rank's sub-batch of step `step` (1-based): (step - 1) * B + group + 1 - consecutive integers over
                                           the groups and steps, like frameSeed over the host's successive batches (UNITY/Raytracer.cs:660) */
} RtowHybridPlan;
RTOW_API int rtowHybridPlan(int32_t worldSize, int32_t rank, int32_t tileCount, uint32_t samplesPerBatch, uint32_t step, RtowHybridPlan* outPlan);
RTOW_API int rtowExchangeAccumDevice(RtowContext context, int32_t width, int32_t height, int32_t tileCount,
                                     const RtowAccumBuffers* partial, const RtowAccumBuffers* accum, int32_t what, void* stream);

RTOW_API int rtowDeviceAlloc(RtowContext context, size_t sizeInBytes, void** outPointer);
RTOW_API int rtowDeviceFree(RtowContext context, void* pointer);
RTOW_API int rtowDeviceCopy(RtowContext context, const void* source, void* destination, size_t sizeInBytes, int kind);
RTOW_API int rtowDeviceMemset(RtowContext context, void* pointer, int value, size_t sizeInBytes);
/* Waits for everything the context's own stream holds AND for the most recent sample batch wherever it was enqueued, then reports that
 * batch history's status: RTOW_ERROR_CAPACITY if any batch since the last report met a ray beyond the hit-list capacity (the flag is sticky
 * until reported), else RTOW_SUCCESS. */
RTOW_API int rtowSynchronize(RtowContext context);
/* The same status query without draining the context's stream: blocks only until the most recent sample batch (on whatever stream it
 * was given) has finished.  For callers of the asynchronous rtowSampleBatchDevice, which cannot return the batch's status itself. */
RTOW_API int rtowGetBatchStatus(RtowContext context);

#ifdef __cplusplus
}
#endif
#endif /* RTOW_H */
