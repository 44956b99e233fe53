// rtow_kernels.h - launch interface between the C-ABI layer (rtow_api.hip) and the gfx950 kernels (rtow_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtow.h"
#include "rtow_bvh.h"
#include "rtow_scene.h"

namespace rtow {

// One persistent workgroup per CU: 1024 lanes (16 wavefronts, 4 per SIMD, <= 128 VGPRs) share one LDS image of the scene.
#ifndef RTOW_BLOCK_THREADS
#define RTOW_BLOCK_THREADS 1024
#endif
constexpr int kBlockThreads = RTOW_BLOCK_THREADS;
constexpr int kLdsBytesMax = 160 * 1024;
#ifndef RTOW_TRAV_SLICE
#define RTOW_TRAV_SLICE 8   // box-walk node visits per scheduler trip
#endif
constexpr int kCandCapacity = 8;   // per-lane list of leaf candidates awaiting their exact test (flushed when full)
constexpr int kQueueBytes = 384;   // per-wave pixel-ticket chunk {next, end, needDone, chunk} (16 waves x 16 B), then 128 bytes of launch constants (view, sky, frame size: RTOW_LDS_VIEW)
constexpr int kHistoryInRegisters = 8;   // path-history codes (one per surface hit of a path) every variant keeps in registers; the variants for deeper paths keep the rest in LDS rows

// LDS of one workgroup of the sample kernel, front to back.  Every size is decided per launch (round 6; rounds 1 - 5 reserved 24 stack rows whatever the tree):
//   [candidates: 8 rows][traversal stack: one row per inner level of THIS scene's tree]      rows of 1024 codes (2 bytes; 4 with wide codes) - one per-lane base, row offsets are immediates
//   [wide codes only: the 256-byte dump row of the walk's prefetches]
//   [path history: traceDepth - 8 rows of 1024 16-bit codes]                                 only the variants for paths deeper than 16 (and the generic ones): the codes of depth >= 8
//   [wave queues 256 B][launch constants 128 B]
//   [scene image: the whole blob, or the top of the node array]
// The cover scene's tree is 11 levels deep: 26 KB come back, which is where the reference host's committed traceDepth 32 keeps its path history (48 KB) - rounds 1 - 5 kept it in a
// 448-byte private segment per lane, cleared per sample: 186 GB of HBM writes per 10-batch launch (profiles/r05_hostdefault_pmc_summary.json).
struct LdsPlan {
    uint32_t stackRows;      // >= 1
    uint32_t histOffset;     // byte offset of the history rows (0: none in LDS)
    uint32_t histRows;       // history rows in LDS
    uint32_t histSpillRows;  // ... and the rest of the rows the launch needs, in HBM (SampleKernelArgs.histSpill)
    uint32_t frontBytes;     // everything in front of the wave queues
    uint32_t sceneBytes;     // bytes of the blob staged behind the queues
    uint32_t nodeCount;      // nodes [0, nodeCount) are LDS resident
    bool allLds;
};
// histRowsNeeded = traceDepth - kHistoryInRegisters for the variants that keep codes in rows (else 0).  A scene stays whole in LDS only if ALL the rows fit next to it (the
// scene-in-LDS kernels carry no code for rows elsewhere: it cost them their freedom from a private segment); otherwise the top of the tree is staged - at least 256 nodes - the
// history gets the rows that fit, and the rest of the rows live in HBM: paths that deep are rare (2.5 segments on average) and rows are only touched beyond depth 8.
inline LdsPlan planLds(bool wide, const SceneLayout& L, uint32_t histRowsNeeded, uint32_t budgetOverride)
{
    LdsPlan p{};
    const uint32_t codeBytes = wide ? 4u : 2u, rowBytes = (uint32_t)kBlockThreads * 2u;
    p.stackRows = L.bvhDepth < 1u ? 1u : L.bvhDepth;
    const uint32_t rowsEnd = ((uint32_t)kCandCapacity + p.stackRows) * (uint32_t)kBlockThreads * codeBytes + (wide ? 256u : 0u);
    const uint32_t room = (uint32_t)kLdsBytesMax - rowsEnd - (uint32_t)kQueueBytes;           // (8 + 24) x 4 KB + 640 B at most: always positive
    uint32_t sceneBudget = room;
    if (budgetOverride >= sizeof(GpuNode) && budgetOverride < sceneBudget) sceneBudget = budgetOverride;      // development aid: small scenes through the tree-in-HBM kernels
    const bool whole = !wide && L.totalBytes <= sceneBudget && (uint64_t)L.totalBytes + (uint64_t)histRowsNeeded * rowBytes <= (uint64_t)room;
    const uint32_t keep = whole ? L.totalBytes : (uint32_t)sizeof(GpuNode) * (L.nodeCount < 256u ? (L.nodeCount ? L.nodeCount : 1u) : 256u);   // what the history rows must leave room for
    const uint32_t keepFits = keep <= sceneBudget ? keep : sceneBudget;
    uint32_t rowsFit = room > keepFits ? (room - keepFits) / rowBytes : 0u;
    p.histRows = histRowsNeeded < rowsFit ? histRowsNeeded : rowsFit;
    p.histSpillRows = histRowsNeeded - p.histRows;
    p.histOffset = p.histRows ? rowsEnd : 0u;
    p.frontBytes = rowsEnd + p.histRows * rowBytes;
    uint32_t budget = (uint32_t)kLdsBytesMax - p.frontBytes - (uint32_t)kQueueBytes;
    if (budget > sceneBudget) budget = sceneBudget;
    if (whole) {
        p.sceneBytes = L.totalBytes; p.nodeCount = L.nodeCount; p.allLds = true;
    } else {
        // too large for LDS: stage the top of the (breadth-first) node array, read the rest through L2
        uint32_t nodes = budget / (uint32_t)sizeof(GpuNode);
        if (nodes > L.nodeCount) nodes = L.nodeCount;
        p.nodeCount = nodes; p.sceneBytes = nodes * (uint32_t)sizeof(GpuNode); p.allLds = false;
    }
    return p;
}
// History words of the variant that serves a launch (launchByDiagGeo, rtow_sample_kernel.hip.h): 4 / 8 = every code in registers (trace depth <= 8 / <= 16), 32 = the generic
// variants, whose codes beyond kHistoryInRegisters live in LDS rows - the host sizes the launch's LDS from this (rtow_api.hip)
inline int historyWords(int noiseColor, bool perSample, bool wide, bool ties, bool fullDiag, int traceDepth)
{
    if (noiseColor != RTOW_NOISE_WHITE) return 32;
    if (perSample) return (!ties && !wide && !fullDiag && traceDepth <= 8) ? 4 : 32;
    if (fullDiag) return 32;
    if (traceDepth <= 8) return 4;
    if (traceDepth <= 16 && (!wide || ties)) return 8;
    return 32;
}
constexpr unsigned kNoPrimaryList = 0x0000ffffu;   // pixelCandidates[pix].x: first slot empty, second not - "no list, walk the tree"
#ifndef RTOW_URGENT_LANES
#define RTOW_URGENT_LANES 1   // 0: A/B build without the lanes in a hurry (rtow_sample_kernel.hip.h HURRY; rtow_api.hip then sets no rate)
#endif
#ifndef RTOW_SAMPLE_GROUP
#define RTOW_SAMPLE_GROUP 16     // (other values: timing builds only - the oracle's restatement of the policy sums groups of 16)
#endif
constexpr unsigned kSampleGroup = RTOW_SAMPLE_GROUP;   // RTOW_RNG_PER_SAMPLE: samples per work unit (part of that policy's definition: partial sums are per group)
constexpr unsigned kChunkOrderScratchWords = 1056;   // behind the 2 x chunkCount words of the chunk-cost array: histogram and maximum of the multi-workgroup chunk order (rtow_kernels.hip)
constexpr int kMaxChain = 16;      // successive batches one launch can run (rtowSampleBatchChainDevice)
constexpr int kLocalHitEntries = 24;                // entries of a ray's hit list a lane holds itself; longer lists continue in SampleKernelArgs.hitSpill
constexpr uint32_t kDefaultHitListCapacity = 1024;  // RtowContextOptions.hitListCapacity == 0, scenes with volumes (long lists are their normal case)
constexpr uint32_t kTieRedoCapacity = 1u << 20;     // pixel-batches one launch may hand to the tie fix-up pass (beyond it: RTOW_ERROR_CAPACITY)
constexpr int kTieRedoBlocks = 8;                   // workgroups of the fix-up launch (it exits at once when the list is empty)
constexpr uint32_t kDefaultTieListCapacity = 128;   // ... scenes without: only the exact-tie procedure keeps a whole list, and only for the ray that ties

// Chained launches keep all batches of a pixel chunk on one XCD (rtow_sample_kernel.hip.h, "Chained batches"): per XCD, how many chunks it took for
// batch 0 and the ticket counter of its later batches; kMaxXcds lists of chunk numbers (chunkCount entries each, 0xffffffff = not written yet) follow
// the struct in the same allocation.  Zeroed (lists: 0xff) before every chained launch.
constexpr unsigned kMaxXcds = 16;        // XCC_ID is four bits wide; MI355X has eight
struct XcdState {
    unsigned owned[kMaxXcds];     // chunks this XCD took for batch 0
    unsigned ticket[kMaxXcds];    // tickets handed out for its later batches
    unsigned listed;              // batch-0 chunks whose list entry is written (all XCDs): the later batches start at chunkCount
    unsigned pad[15];
};

// Per-batch fields of a chained launch (rtowSampleBatchChainDevice): everything else is shared by the chain's batches.
struct ChainBatch {
    uint8_t* diagnostics;   // of this batch, may be null
    uint32_t seed;          // Seed (JOBS/SampleBatchJob.cs:28,91)
    uint32_t pad;
    float *outColor, *outNormal, *outAlbedo, *outScw;   // batch groups (rtowSampleBatchGroupDevice): this batch's own outputs; unused in a chain
};

// Everything the sample kernel needs, passed by value (kernarg segment).
// Owned pixel number -> (column, owned row).  The 64 tickets of a chunk go to the 64 lanes of one wave: numbered row by row they are a 64 x 1 strip of
// the image, numbered in tiles an 8 x 8 block - whose camera rays meet the same few entities and materials, so the lanes of a wave agree more often
// on the stage they are in and on the shading class they run (same box, alternating runs, gpurun_out/r03ah: cover +3 %, 10 000 spheres +4 %,
// moving + defocus +3 %, 4K +2.8 %, 250 k-triangle mesh +2.5 %).  Results do not depend on the numbering (a pixel's samples depend on its index
// and the seed only).  Tiles need a width that is a multiple of 8; the owned rows beyond the last whole tile row (fewer than 8) are numbered row by
// row behind the tiles.  RTOW_TICKET_TILES=0 builds strips only, for A/B runs.
// [ticket numbering: begin]  (tests/test_ticket_numbering.py compiles the text between the two markers for the host and checks that it is a bijection)
#ifndef RTOW_TICKET_TILES
#define RTOW_TICKET_TILES 1
#endif
#ifndef RTOW_TICKET_TILE_W
#define RTOW_TICKET_TILE_W 8            // tile width in pixels (a power of two up to 64); the tile is this wide and 64 / width high
#endif
constexpr unsigned kTileW = RTOW_TICKET_TILE_W, kTileH = 64u / kTileW;
static_assert(kTileW * kTileH == 64u && (kTileW & (kTileW - 1u)) == 0u, "a tile is one 64-ticket chunk");
__host__ __device__ inline void owned_pixel_xy(unsigned n, unsigned width, unsigned tilesPerRow, unsigned tiledPixels, int& cx, int& ownedRow)
{
    if (n < tiledPixels) {
        const unsigned tile = n >> 6, i = n & 63u;
        const unsigned ty = tile / tilesPerRow, tx = tile - ty * tilesPerRow;
        cx = (int)(tx * kTileW + (i & (kTileW - 1u)));
        ownedRow = (int)(ty * kTileH + i / kTileW);
    } else {
        ownedRow = (int)(n / width);                       // tiledPixels is a whole number of rows: the rows behind the tiles keep their row-major numbers
        cx = (int)(n - (unsigned)ownedRow * width);
    }
}
// [ticket numbering: end]

struct SampleKernelArgs {
    // accumulators (JOBS/SampleBatchJob.cs:41-51)
    const float* inColor;   // float4[N]
    const float* inNormal;  // float3[N]
    const float* inAlbedo;  // float3[N]
    const float* inScw;     // float [N]
    float* outColor;
    float* outNormal;
    float* outAlbedo;
    float* outScw;
    uint8_t* diagnostics;   // may be null
    int32_t diagnosticsStride;

    // scene
    const uint8_t* sceneBlob;
    SceneLayout layout;
    uint32_t ldsSceneBytes;  // bytes of the blob staged into LDS (whole blob, or a node prefix)
    uint32_t ldsNodeCount;   // nodes [0, ldsNodeCount) are LDS resident
    uint32_t ldsStackRows, ldsHistOffset, ldsFrontBytes;   // this launch's LDS plan (LdsPlan above): traversal-stack rows, where the path-history rows start (0: none), where the wave queues start
    uint32_t ldsHistRows, histSpillRows, histSpillStride;  // history rows in LDS; rows beyond them in histSpill ([row][workgroup x 1024 + lane], histSpillStride entries per row)
    unsigned short* histSpill;                             // null = none

    // work distribution
    unsigned int* workCounter;            // zeroed before the launch; counts 64-pixel ticket chunks
    const unsigned int* chunkOrder;       // chunk launch order (most expensive first), null = natural order
    unsigned short* pixelCost;            // [64 * chunkCount], ticket order: ray count of every pixel of THIS launch (input of the next launch's order); null = not recorded
    const unsigned int* ticketMap;        // [tiledPixels] ticket -> owned-pixel number (owned_pixel_xy's argument): which pixels share a chunk, i.e. a wave (regroup_tickets_kernel: the pixels of a
                                          // super-tile of tiles sorted by ray count and dealt out 64 at a time); a permutation inside every super-tile; null = the tiles themselves
    uint32_t chunkCount;
    uint32_t slotBlock, groupRecip;       // batch groups: (chunk, batch) slots a wave reserves at a time (>= 1: it then works through the batches of one tile, and the neighbours in the cost order, one after the other) and 2^32 / chainCount + 1 (slot -> chunk by a multiply)
    const uint2* pixelCandidates;         // [2 x width * height] = one uint4 per pixel (8 x 16-bit node codes, or 4 x 32-bit): camera-ray candidate list of every owned pixel (primary_candidates_kernel), null = walk every ray;
                                          // wideCodes: uint4 records (4 x 32-bit node indices) behind the same pointer
    int32_t probeOnly;                    // > 0: probe - this many samples per pixel, nothing stored but pixelCost (1: the cost probe; 4: the threshold-tuning probes)
    const volatile uint32_t* cancelFlag;  // host-pinned, may be null
    uint32_t* overflowFlag;               // host-pinned: set when a ray's hit list (volume scenes) exceeds the per-lane capacity
    uint32_t totalWork;                   // owned pixels = ownedRows * width
    int32_t width, height;
    uint32_t tilesPerRow, tiledPixels;    // owned pixels [0, tiledPixels) are numbered in 8 x 8 tiles, width / 8 tiles per row, the rest row by row (owned_pixel_xy)

    // SampleBatchJob parameter block (:25-39)
    float sizeX, sizeY;
    int32_t sliceOffset, sliceDivider;
    uint32_t seed;
    RtowView view;
    RtowEnvironment environment;
    uint32_t sampleCountMin, sampleCountMax;
    int32_t traceDepth;
    int32_t subPixelJitter;
    float extremaX, extremaY;

    // RTOW_RNG_PER_SAMPLE: work units are (owned pixel, group of kSampleGroup samples); totalWork counts units
    float* unitRecords;                   // [totalWork] x 16 floats, null = reference policy (units are pixels)
    uint32_t groupsPerPixel;              // ceil(sampleCountMax / kSampleGroup), 1 under the reference policy
    int32_t xoroshiro;                    // per-sample policies: 1 = xoroshiro64** (RTOW_RNG_PER_SAMPLE_XOROSHIRO), 0 = Unity's xorshift32 reseeded per sample

    // Image textures (SCENE_KIND_TEXTURED): GpuTexMaterial / GpuImage tables and pixels, in HBM
    const uint8_t* texBlob;
    TexLayout texLayout;

    // noise source (RT/RandomSource.cs): the texture of this batch for Blue / SpatioTemporalBlue (null for white)
    int32_t noiseColor;                   // RtowNoiseColor
    uint32_t blueRowStride, stbRowStride;
    const void* blueNoise;                // half4[blueRowStride^2]
    const uint8_t* stbScalar;             // byte
    const uint8_t* stbVector2;            // RGB24
    const uint8_t* stbCosineUnitVector3;  // RGBA32
    const uint8_t* stbUnitVector2;        // RGB24
    const uint8_t* stbUnitVector3;        // RGB24

    // sky cubemap (RT/Texture.cs:141-211), used when environment.skyType == RTOW_SKY_CUBEMAP; cubemapData may be null (-> black)
    const uint8_t* cubemapData;
    int32_t cubemapHalfW, cubemapHalfH, cubemapW1, cubemapH1;   // halfFaceSize, faceSizeMinusOne
    int32_t cubemapPixelStride, cubemapRowStride, cubemapFaceStride, cubemapChannelType;

    // RTOW_CONTEXT_REFERENCE_DIAGNOSTICS: the reference's own tree (RefTreeNode[], HBM only); when set, BoundsHitCount / CandidateCount of the
    // 16-byte diagnostics count THAT tree's boxes and leaves (JOBS/SampleBatchJob.cs:427-440), one extra unpruned walk per ray
    const uint8_t* refTree;

    // hit lists longer than the 24 entries a lane holds itself (volume scenes, exact-tie kernels; the reference's HybridList grows on the heap,
    // UTIL/HybridCollections.cs:65-71): entry e >= 24 of lane l of workgroup g is hitSpill[(e - 24) * hitSpillStride + g * 1024 + l]
    uint4* hitSpill;                      // null = none
    uint32_t hitSpillEntries, hitSpillStride;

    // chained batches (rtowSampleBatchChainDevice): this launch runs chainCount successive batches of the same frame; batch b of a
    // 64-pixel chunk starts as soon as batch b - 1 of that chunk is stored (chunkDone), whichever CU traced it
    uint32_t chainCount;                  // >= 1; 1 = a plain batch
    int32_t chainIndependent;             // batch group (rtowSampleBatchGroupDevice): the chainCount batches all read the launch's inputs and store to their own outputs
                                          // (ChainBatch.out*): no hand-off between them, tickets are (chunk, batch) pairs from the one queue
    const ChainBatch* chainBatches;       // [chainCount] what differs between the batches of the chain (device memory: indexed per lane)
    unsigned int* chunkDone;              // [chunkCount] pixels stored so far, all batches of this launch; zeroed before the launch
    XcdState* xcdState;                   // chunk ownership per XCD + the lists behind it (chained launches only)

    // Nearest-hit ties in the rank-rule sphere kernels.  A lane that meets two DIFFERENT spheres at bit-identical distance sets its pixel's bit in
    // tieBits (a bitmap over the frame's pixels, zeroed before the launch); afterwards collect_tied_pixels turns the bits into tieRedo ([0] = count, [4 ...] = entries:
    // batch << 27 | frame pixel index) and a second launch of the exact-tie kernel of the same kind (redoMode) renders the listed pixels again - every batch of the
    // launch, from the launch's inputs (a copy of them when the launch accumulates in place) - over what the first launch stored.  Null / 0: no watch (scenes of at
    // most 16 entities, exact-tie kernels, per-sample policies, probes).
    unsigned* tieBits;
    unsigned* tieRedo;
    uint32_t tieRedoCapacity;
    int32_t redoMode;

    // launch geometry (rtow_sample_kernel.hip.h, GEO): lanes per workgroup of THIS launch (1024) and whether candidate / stack codes are 32 bits
    // wide (scenes beyond 65 535 entities or tree nodes)
    int32_t blockThreads;
    int32_t wideCodes;

    // scheduler: minimum lane population for a stage to run, indexed by lane state (REGEN TRAV TEST HIT SKY), + box-walk slice.  tune[7]: bits 0 .. 7 = lanes of a wave that must want a
    // pixel boundary before the boundary block runs; bits 8 .. 31 = the float bits (low eight dropped) of the rate - rays per sample done so far - beyond which a pixel's lane stops
    // waiting for company (kernel: HURRY; 0 = none: the launch runs the variants without that code, which read tune[7] as the gate alone)
    int32_t tune[8];
    int32_t travSlice;
};

struct KernelInfo {
    int numRegs;          // VGPRs per lane
    int sharedSizeBytes;  // static LDS
    int maxDynamicLds;
};

// launchers (defined in rtow_kernels.hip)
// per scene kind, each defined in its own translation unit (rtow_sample_*.hip)
hipError_t launchSampleSpheres(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleSpheresMotion(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleGeneral(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleGeneralTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTexturedTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleSpheresTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleSpheresMotionTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleVolumes(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTextured(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleVolumesTextured(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTriangles(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTrianglesTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTrianglesTextured(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleTrianglesTexturedTies(const SampleKernelArgs& args, int numBlocks, hipStream_t stream, bool allLds);
hipError_t launchSampleBatch(const SampleKernelArgs& args, int numBlocks, hipStream_t stream);
hipError_t launchPrepareMaterials(uint8_t* blob, const SceneLayout& layout, hipStream_t stream); // derived material constants, on device
hipError_t launchPrimaryCandidates(const SampleKernelArgs& args, uint2* out, hipStream_t stream);
hipError_t launchFoldUnitRecords(const SampleKernelArgs& args, hipStream_t stream);
hipError_t launchBuildChunkOrder(const unsigned short* pixelCost, unsigned* cost, unsigned chunkCount, unsigned* order, int byMax, hipStream_t stream);
// ticketMap[t] = t for t < count (the tiles themselves)
hipError_t launchInitTicketMap(unsigned* ticketMap, unsigned count, hipStream_t stream);
// Which pixels share a wave.  Per super-tile of side x side tiles (8 x 8 pixels each; tile t = chunk t = tickets [64 t, 64 t + 64)): the super-tile's pixels sorted by the ray count the
// last launch measured for them (pixelCost, ticket order under the map as it is), most expensive first, and dealt out to the super-tile's own chunks 64 at a time; pixelCost is
// permuted along, so that it stays in ticket order under the NEW map.  side: 2 / 4 / 8.
constexpr unsigned kRegroupMaxSide = 8;
// classes: {0, -, -} = sort by the ray count itself; {t1, t2, t3} = by cost class (<= t1, <= t2, <= t3, beyond), pixels of a class in tile order
hipError_t launchRegroupTickets(unsigned short* pixelCost, unsigned* ticketMap, unsigned tilesPerRow, unsigned tileRows, unsigned side, const unsigned classes[3], hipStream_t stream);
hipError_t launchPrepareEntities(uint8_t* blob, const SceneLayout& layout, hipStream_t stream);  // inverse transforms of general entities, on device
hipError_t launchCombine(const RtowCombineParams& p, const float* inColor, const float* inNormal, const float* inAlbedo,
                         float* outColor, float* outNormal, float* outAlbedo, hipStream_t stream);
// thresholds: the 261-float step table of the float -> byte conversion (rtow_finalize.hip.h), built once per context by launchBuildByteThresholds
constexpr size_t kByteThresholdTableBytes = 259 * sizeof(float);
hipError_t launchBuildByteThresholds(float* thresholds, hipStream_t stream);
hipError_t launchFinalize(int pixelCount, const float* inColor, const float* inNormal, const float* inAlbedo,
                          uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, const float* thresholds, hipStream_t stream);

// CombineJob -> FinalizeTexturesJob in one pass (the bytes of the two kernels one after the other)
hipError_t launchCombineFinalize(const RtowCombineParams& p, const float* inColor, const float* inNormal, const float* inAlbedo,
                                 uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, const float* thresholds, hipStream_t stream);
// dst[k] += src[k] for the four accumulators (float4 / float3 / float3 / float per pixel) in one launch
hipError_t launchAddAccum(size_t pixels, float* const dst[4], const float* const src[4], hipStream_t stream);
// rows first, first + step, ... (`rows` of them, `rowFloats` floats each) of a full-frame buffer -> / <- one contiguous block
// bits of tieBits -> entries of tieRedo (one per marked pixel; `batches` per pixel, batch index in bits 27.., for a batch group)
hipError_t launchCollectTiedPixels(const unsigned* tieBits, unsigned words, unsigned* tieRedo, unsigned capacity, unsigned batches, uint32_t* overflowFlag, unsigned busyAt, hipStream_t stream);
hipError_t launchCopyRows(float* frame, float* packed, unsigned rowFloats, unsigned rows, unsigned first, unsigned step, bool toFrame, hipStream_t stream);
// accum[row] += src_0[row] ... += src_{groups-1}[row] (group order) for rows first, first + step, ...; src_g = ownPartial (frame layout) for g == own, else packed rows at recv + g * regionFloats
hipError_t launchFoldRows(float* accum, const float* ownPartial, const float* recv, size_t regionFloats, unsigned rowFloats, unsigned rows, unsigned first, unsigned step,
                          unsigned groups, unsigned own, hipStream_t stream);

// rtowProbeNearestHit (rtow_probe.hip): one ray walked on the host through the scene's host image (derived entity transforms included); false = miss
bool probeNearestHitHost(const uint8_t* blob, const SceneLayout& L, const int32_t* entityOfPrim, const float origin[3], const float direction[3], float time, float* distance, int* entity);

// same-XCD hand-over litmus of the chained launches (rtow_kernels.hip): pairs of workgroups that ran on one XCD, stale dwords seen, waits that timed out
hipError_t runXcdCoherenceLitmus(int cuCount, hipStream_t stream, unsigned* outPairs, unsigned* outStale, unsigned* outTimeouts);

// Per-block partial results of the metrics reduction; the host folds them in block order.
struct MetricsPartial {
    long long rays, samples;
    float minW, maxW, minS, maxS;
};
constexpr int kMetricsBlocks = 2048;   // 8 workgroups of 256 lanes per CU: 256 blocks (one per CU) left the reduction at 3.5 TB/s, latency bound (profiles/r03a_post_passes.json)
hipError_t launchReduceMetrics(int pixelCount, const uint8_t* diagnostics, int stride, const float* color, const float* scw,
                               MetricsPartial* partials, hipStream_t stream);
// the partials of launchReduceMetrics -> one RtowMetrics record in device-visible memory (the asynchronous form of the reduction)
hipError_t launchFoldMetrics(const MetricsPartial* partials, RtowMetrics* out, hipStream_t stream);

} // namespace rtow
