// rtow_api.hip - implementation of the C ABI declared in include/rtow.h (librtow_hip.so).
//
// Host-side runtime of the path: context / device selection, scene compilation + upload, grow-only device staging
// for the host-buffer entry point, kernel launch on a HIP stream with HIP-event timing, cooperative cancellation,
// and the post passes.  There is no CPU implementation behind any entry point: without a usable HIP device
// rtowCreateContext fails with RTOW_ERROR_NO_DEVICE and nothing else can be called.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtow.h"
#include "rtow_bvh.h"
#include "rtow_kernels.h"

// minimum lane population per stage, in 64ths of the wave's live lanes: REGEN TRAV TEST HIT SKY VOL | candidates that end a walk (hand-over to TEST) | unused | box-walk slice (node visits per trip).
// Sphere kinds: REGEN from 3/8, the walk and HIT from 1/2, SKY from 7/16, TEST at once.  Since a chunk's 64 tickets are an 8 x 8 tile of the image
// (rtow_kernels.h) the lanes of a wave meet the same few materials, and a HIT stage that waits for half of them runs its class bodies a third as
// often with three times the lanes; with 64 x 1 strips any threshold on HIT lost (rounds 1 / 2).  Same box, alternating runs, gpurun_out/r03am-r03ao:
// cover 9 417 against 8 685 Msamples/s (+8.4 %), 10 000 spheres 7 513 against 7 042 (+6.7 %), moving + defocus 6 306 against 5 958 (+5.8 %),
// 4K / 1024 spp / 16 bounces 9 464 against 8 851 (+6.9 %).  The general-entity kinds keep REGEN 1/4, walk 3/4, HIT and SKY at once (kGeneralTune):
// on the 250 k-triangle mesh HIT from 1/2 loses 6 % (1 732 against 1 840), SKY from 1/2 3 %.
#ifndef RTOW_GROUP_SLOT_BLOCK
#define RTOW_GROUP_SLOT_BLOCK 4     // (chunk, batch) slots a wave reserves at a time in a batch group (SampleKernelArgs.slotBlock): groups + fold 9 978 -> 10 160 Msamples/s at 4, 10 118 at 2
#endif
static constexpr unsigned kGroupSlotBlock = RTOW_GROUP_SLOT_BLOCK;
#ifndef RTOW_DEFAULT_REGROUP_SIDE
// RtowContextOptions.schedulerTune[7], which pixel a ticket stands for: 1 = its place in its 8 x 8 tile; 3 = the tiles as they are, each tile's tickets most expensive pixel first
// (order_tile_tickets_kernel: +0.7 % on the headline, +0.9 % as plain launches, +1.3 % as groups, same box, three alternating runs - profiles/r05a_pixel_regrouping.json);
// 2 / 4 / 8 (+ 16 x mode): pixels regrouped by cost or class inside super-tiles of that many tiles (0 ... -5 %: measured, not used)
#define RTOW_DEFAULT_REGROUP_SIDE 3
#endif
#ifndef RTOW_PIXEL_GATE
#define RTOW_PIXEL_GATE 1     // lanes of a wave that must want a pixel boundary before the boundary block runs (1 = at once; the kernel's A.tune[7]); measured: see HISTORY.md round 6
#endif
#ifndef RTOW_URGENT_RAYS_PER_SAMPLE
// lanes in a hurry (kernel: HURRY): the rate - rays per sample done so far - beyond which a pixel's lane stops waiting for company.  Same box (profiles/r06x_lanes_in_a_hurry.json,
// run_r06ag.sh / run_r06ah.sh): cover scene at depth 32, chains: 14 -> 7 126 Msamples/s, 18 -> 7 302, 24 -> 6 220 (nobody is in a hurry any more); 10 000 spheres at depth 32:
// 14 -> 5 370, 18 -> 4 895, 24 -> 4 180.  (Float bits with the low eight zero: they share tune[7] with the pixel gate.)
#define RTOW_URGENT_RAYS_PER_SAMPLE 18.0f
#define RTOW_URGENT_RAYS_PER_SAMPLE_BEYOND_LDS 14.0f
#endif
#ifndef RTOW_DEFAULT_TUNE
#define RTOW_DEFAULT_TUNE 24, 32, 1, 32, 28, 1, 3, 1, 16
#endif
#ifndef RTOW_GENERAL_TUNE
#define RTOW_GENERAL_TUNE 16, 48, 1, 1, 1, 1, 3, 1, 16
#endif
// A second general family - REGEN and SKY from 1/8 - is what the 250 k-triangle mesh wants (2 167 against 2 037 Msamples/s) and the Cornell box with
// volumes does not (840 against 920; mixed primitives +0.7 %; gpurun_out/r03bv): a third candidate for the per-scene measurement below.
#ifndef RTOW_GENERAL_TUNE_2
#define RTOW_GENERAL_TUNE_2 8, 48, 1, 1, 8, 1, 3, 1, 16
#endif
// Which of the two families suits a scene is a property of the scene, not of its kernel kind: an image-textured scene of spheres runs 25 % faster
// on the sphere kinds' thresholds (10 160 against 8 100 Msamples/s), a scene of rects, boxes and triangles 9 % slower (3 220 against 3 540), a mesh with
// fog volumes 15 % faster when the volume stage too waits for half of the live lanes (460 against 399; gpurun_out/r03ax).  So the first batch after
// rtowUploadScene MEASURES them: each candidate (three families; six for volume scenes) renders kTuneProbeSamples samples per pixel through the batch's own kernel (a probe:
// nothing is stored), twice, timed with events on the batch's stream; the fastest one stays for the scene.  A probe of a few samples per pixel ranks the
// candidates like the full workload does (cover, textured, mixed, volumes, 10 000 spheres at 2 / 4 / 8 / 64 samples per pixel: same order every
// time, gpurun_out/r03ay).  Thresholds never change a result (tests/test_gpu_fullsize.py: frames under every schedule).  The call that tunes
// waits for its probes (a few milliseconds to ~0.1 s, once per scene); RTOW_CONTEXT_NO_THRESHOLD_TUNING keeps the per-kind values above.
constexpr int kTuneProbeSamples = 4, kTuneProbeRepeats = 3;
constexpr uint64_t kTuneMinSamplesPerPixel = 64;    // ... and the scene must have been asked for this many samples per pixel since its upload before it is worth 40 - 76 of probes
constexpr float kTuneMargin = 0.98f;      // a candidate replaces the kind's built-in family only if its best probe is more than 2 % faster: single probes (1 - 10 ms)
                                          // scatter by a few per cent, and a wrong pick costs more than a missed one (10 000 spheres: family 2 picked once in r03bw, -7 %)

using namespace rtow;

struct RtowContext_t {
    int device = 0;
    int cuCount = 0;
    RtowLogCallback logCb = nullptr;
    void* logData = nullptr;
    int logLevel = 0;

    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipEvent_t evBatchDone = nullptr;   // end of everything the last sample batch enqueued (kernel + chunk-order refresh)
    bool haveBatchDone = false;
    bool haveTiming = false;

    // scene
    bool haveScene = false;
    CompiledScene scene;
    uint8_t* dScene = nullptr;
    size_t dSceneCapacity = 0;
    uint32_t ldsSceneBytes = 0, ldsNodeCount = 0;
    unsigned short* dHistSpill = nullptr; // path-history rows that do not fit LDS (LdsPlan.histSpillRows), [row][workgroup x 1024 + lane]
    size_t histSpillBytes = 0;
    LdsPlan ldsPlan{};                    // of launches whose variant keeps its whole path history in registers (trace depth <= 16); the others plan per launch (launchSample)

    // work distribution / cancellation
    unsigned int* dWorkCounter = nullptr;
    // chunk cost map -> launch order (longest chunks first); valid for one (width, height, slice, scene) configuration
    unsigned int* dChunkDone = nullptr;   // chained batches: pixels stored per chunk
    uint8_t* dXcdState = nullptr;         // chained batches: XcdState + kMaxXcds lists of chunkDoneCapacity entries (which XCD owns which chunk)
    ChainBatch* dChainBatches = nullptr;  // chained batches: per-batch seed / diagnostics table of the launch being enqueued
    uint32_t chunkDoneCapacity = 0;
    unsigned int *dChunkCost = nullptr, *dChunkOrder = nullptr;
    unsigned short* dPixelCost = nullptr;
    unsigned int* dTicketMap = nullptr;   // ticket -> owned pixel (SampleKernelArgs.ticketMap), chunkCapacity * 64 entries; re-sorted from every launch's cost map
    bool orderMapped = false;             // the cost map / order on hand were recorded under dTicketMap (else under the tiles themselves)
    uint32_t chunkCapacity = 0;
    bool orderValid = false;
    int orderW = 0, orderH = 0, orderOff = 0, orderDiv = 0;
    volatile uint32_t* hCancel = nullptr; // pinned, device-visible: [0] cancel, [1] hit-list overflow, [2] tie-list overflow; the metrics record of rtowReduceMetricsDevice at byte 64
    volatile RtowMetrics* hMetricsRecord = nullptr;
    RtowMetrics* dMetricsRecord = nullptr;
    // RTOW_RNG_PER_SAMPLE: one 64-byte record per (owned pixel, sample group) unit
    float* dUnitRecords = nullptr;
    size_t unitRecordCapacity = 0;
    uint32_t orderGroups = 1;     // groups per pixel the chunk cost map was recorded with
    // RTOW_CONTEXT_REFERENCE_DIAGNOSTICS: the reference's own tree of the current scene (CompiledScene.refTree), HBM only
    uint8_t* dRefTree = nullptr;
    size_t refTreeCapacity = 0;
    // hit lists beyond the 24 entries a lane holds itself (volume scenes, exact-tie kernels): [entry][lane] columns, grow-only
    uint4* dHitSpill = nullptr;
    uint32_t hitSpillEntries = 0;         // of the current scene (<= hitSpillCapacity)
    uint32_t hitSpillCapacity = 0;        // entries per lane the allocation holds
    uint32_t hitListCapacity = 0;         // RtowContextOptions.hitListCapacity (0 = default)
    uint32_t grownListCapacity = 0;       // hitListCapacity == 0 only: what the capacity has grown to after batches that met longer lists (growHitList); kept across scenes
    bool triWatchOff = false;             // this all-triangle scene ties too often for the tie watch (a watched launch marked thousands of pixels, or more than the list holds): exact-tie kernels from now on
    bool overflowGrew = false;            // the last reported overflow enlarged the capacity: the same batch, issued again, has room
    // Image-texture blob of the current scene (CompiledScene.texBlob), HBM only
    uint8_t* dTexBlob = nullptr;
    size_t texBlobCapacity = 0;
    // noise texture sets (rtowUploadBlueNoise / rtowUploadStbNoise): device copies, `textureCount` textures back to back
    uint8_t* dBlueNoise = nullptr;
    uint32_t blueRowStride = 0, blueTextureCount = 0;
    uint8_t* dStbNoise = nullptr;        // scalar | vector2 | cosineUnitVector3 | unitVector2 | unitVector3 sets, in this order
    uint32_t stbRowStride = 0, stbTextureCount = 0;
    // sky cubemap (rtowUploadSkyCubemap)
    uint8_t* dCubemap = nullptr;
    size_t cubemapCapacity = 0;
    RtowCubemapDesc cubemap{};   // .faces is not kept (host pointer): dCubemap holds the copy, null when none
    // camera-ray candidate lists (primary_candidates_kernel): valid for one (scene upload, view, size, slice, jitter) configuration
    uint2* dPixCand = nullptr;
    size_t pixCandCapacity = 0;           // bytes
    bool pixCandValid = false;
    uint64_t sceneSerial = 0, pixCandScene = 0;
    RtowView pixCandView{};
    int pixCandW = 0, pixCandH = 0, pixCandOff = 0, pixCandDiv = 0, pixCandJitter = 0;

    // grow-only staging for rtowSampleBatch (host buffers) - like CudaBuffer.EnsureCapacity (OptixApi.cs:240-251)
    float *dColor = nullptr, *dNormal = nullptr, *dAlbedo = nullptr, *dScw = nullptr;
    uint8_t* dDiag = nullptr;
    size_t stagingPixels = 0, stagingDiagBytes = 0;

    MetricsPartial* dPartials = nullptr;

    // nearest-hit ties of the rank-rule sphere kernels (SampleKernelArgs.tieBits / tieRedo): the bitmap the fast kernel marks, the list the fix-up launch renders, a copy
    // of the inputs of launches that accumulate in place, and the fix-up launch's own (small) hit-list spill area
    unsigned* dTieRedo = nullptr;
    unsigned* dTieBits = nullptr;
    size_t tieBitsWords = 0;
    float* dTieInputs = nullptr;          // colour | normal | albedo | weight of `tieInputPixels` pixels
    size_t tieInputPixels = 0;
    uint4* dRedoSpill = nullptr;
    uint32_t redoSpillEntries = 0;

    // RtowContextOptions: behaviour switches and development knobs (nothing is read from the environment)
    bool wideCodes = false;               // current scene: more than 65 535 entities or tree nodes (32-bit candidate / stack codes, tree read from HBM)
    uint32_t flags = 0;
    uint32_t ldsSceneBudget = 0;          // 0 = everything that fits
    int tune[9] = {RTOW_DEFAULT_TUNE};
    bool userTune = false;                // RtowContextOptions.schedulerTune was given: no per-scene adjustment
    int regroupSide = RTOW_DEFAULT_REGROUP_SIDE;   // RtowContextOptions.schedulerTune[7] (see RTOW_DEFAULT_REGROUP_SIDE)
    bool userSliceDefault = false;        // ... with a zero walk slice: the per-scene built-in value
    bool chainFusion = true;              // the same-XCD hand-over litmus passed on this device (rtowCreateContext): chains may run as one launch
    uint64_t tunedScene = ~0ull;          // sceneSerial whose thresholds were measured (tuneThresholds)
    int tunedCandidate = -1;              // which candidate won (rtowGetSceneInfo-independent; logged)
    bool tunePending = false;             // probes of scene tunePendingScene are enqueued; their events are read by a later call, never waited for
    uint64_t tunePendingScene = 0;
    int tuneCandidates = 0, tuneBuiltin = 0;
    std::vector<hipEvent_t> tuneEvents;
    uint32_t* dProbeSink = nullptr;       // where probes report rays beyond the hit-list capacity (not the batch's flag)
    uint64_t sppSinceUpload = 0;          // samples per pixel this scene has been asked for since its upload: a measurement must be worth its probes
    uint64_t sceneSignatureNow = 0;       // of the current scene
    struct TuneCacheEntry { uint64_t signature; int winner; };
    std::vector<TuneCacheEntry> tuneCache;   // winners by scene signature: a re-upload of a like scene does not measure again

    // rtowRegisterHostBuffer: pinned + device-mapped ranges of caller memory
    struct HostRange { uint8_t* base; size_t size; uint8_t* device; };
    std::vector<HostRange> hostRanges;

    // rtowComm*: RCCL communicator of this rank (one process per GPU) and the packed-row staging of rtowGatherRowsDevice
    void* comm = nullptr;                 // ncclComm_t
    int commRank = 0, commWorld = 1;
    float *dGatherSend = nullptr, *dGatherRecv = nullptr;
    size_t gatherSendFloats = 0, gatherRecvFloats = 0;
    float* dByteThresholds = nullptr;     // FinalizeTexturesJob's float -> byte step table (rtow_finalize.hip.h), built when the context is created
    hipEvent_t evGatherDone = nullptr;    // end of the last gather: the staging blocks are per context, gathers may come on different streams
    bool haveGatherDone = false;
    hipEvent_t evMetricsDone = nullptr;   // end of the last metrics reduction (the per-block partials are per context)
    bool haveMetricsDone = false;

    std::mutex mu;
    std::mutex sceneMu;      // guards the HOST image of the scene (scene.blob / layout / entityOfPrim, haveScene) between rtowUploadScene and rtowProbeNearestHit; taken after mu, never the other way round
};

namespace {

void logf(RtowContext ctx, int level, const char* tag, const char* fmt, ...)
{
    if (!ctx || !ctx->logCb || level > ctx->logLevel) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->logCb(level, tag, buf, ctx->logData);
}

#define HIP_TRY(ctx, expr, result)                                                                    \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            logf(ctx, 2, "hip", "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (result);                                                                          \
        }                                                                                             \
    } while (0)

int validateParams(const RtowSampleParams* p)
{
    if (!p) return RTOW_ERROR_INVALID_VALUE;
    const int w = (int)p->size.x, h = (int)p->size.y;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return RTOW_ERROR_INVALID_VALUE;
    if (p->sliceDivider < 1 || p->sliceOffset < 0 || p->sliceOffset >= p->sliceDivider) return RTOW_ERROR_INVALID_VALUE;
    if (p->traceDepth < 1 || p->traceDepth > 64) return p->traceDepth < 1 ? RTOW_ERROR_INVALID_VALUE : RTOW_ERROR_CAPACITY;
    if (p->noiseColor < RTOW_NOISE_WHITE || p->noiseColor > RTOW_NOISE_SPATIOTEMPORAL_BLUE) return RTOW_ERROR_INVALID_VALUE;
    if (p->rngPolicy != RTOW_RNG_REFERENCE && p->rngPolicy != RTOW_RNG_PER_SAMPLE && p->rngPolicy != RTOW_RNG_PER_SAMPLE_XOROSHIRO) return RTOW_ERROR_INVALID_VALUE;
    if (p->rngPolicy != RTOW_RNG_REFERENCE && p->noiseColor != RTOW_NOISE_WHITE) return RTOW_ERROR_INVALID_VALUE;   // the texture walks are per pixel by construction
    if (p->environment.skyType < RTOW_SKY_NONE || p->environment.skyType > RTOW_SKY_CUBEMAP) return RTOW_ERROR_INVALID_VALUE;
    if (p->diagnosticsStride != 4 && p->diagnosticsStride != 16) return RTOW_ERROR_INVALID_VALUE;
    return RTOW_SUCCESS;
}

int ownedRows(const RtowSampleParams* p)
{
    const int h = (int)p->size.y;
    if (p->sliceOffset >= h) return 0;
    return (h - p->sliceOffset + p->sliceDivider - 1) / p->sliceDivider;
}

// chain (optional): {count, seeds[count], diagnostics[count]} - `count` successive batches of the frame in this one launch (seeds[0] / diags[0] are
// batch 0's; p->seed and diag are ignored then)
// outs (optional): a batch group - batch b stores to outs[b] and every batch reads `in` (rtowSampleBatchGroupDevice); null: a chain, batch b reads what b - 1 stored to `out`
struct ChainSpec { int count; const uint32_t* seeds; void* const* diags; const RtowAccumBuffers* outs; };

// per-scene threshold measurement (launchSample): forget a measurement in flight (new scene, context going away)
void dropThresholdTuning(RtowContext ctx)
{
    for (auto& e : ctx->tuneEvents) if (e) (void)hipEventDestroy(e);
    ctx->tuneEvents.clear();
    ctx->tunePending = false;
}

// Which scenes get the same thresholds without being measured again: a host that re-uploads on every scene edit (sceneSerial moves) keeps what was
// measured for a scene of the same kernel kind and size.
uint64_t sceneSignature(const CompiledScene& sc, bool wide)
{
    return ((uint64_t)sc.layout.sceneKind << 60) ^ ((uint64_t)(sc.layout.exactTies ? 1 : 0) << 59) ^ ((uint64_t)(wide ? 1 : 0) << 58) ^ ((uint64_t)sc.layout.nodeCount << 29) ^
           (uint64_t)(uint32_t)sc.entityCount ^ ((uint64_t)sc.layout.materialCount << 44);
}

// read the probes' events if they are all complete (wait == false: otherwise leave them for a later call) and switch to the fastest candidate; true = thresholds changed or confirmed
bool finishThresholdTuning(RtowContext ctx, bool wait)
{
    if (!ctx->tunePending || ctx->tuneEvents.empty()) return false;
    const hipError_t q = wait ? hipEventSynchronize(ctx->tuneEvents.back()) : hipEventQuery(ctx->tuneEvents.back());
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return false; }
    constexpr int kFamilies = 3;
    static const int kSets[kFamilies][9] = {{RTOW_DEFAULT_TUNE}, {RTOW_GENERAL_TUNE}, {RTOW_GENERAL_TUNE_2}};
    const int candidates = ctx->tuneCandidates, builtin = ctx->tuneBuiltin;
    bool ok = q == hipSuccess;
    float best = 0.0f, builtinMs = 0.0f;
    int winner = -1;
    for (int c = 0; ok && c < candidates; c++) {
        float t = 0.0f;
        for (int r = 0; r < kTuneProbeRepeats; r++) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ctx->tuneEvents[(size_t)(r * candidates + c)], ctx->tuneEvents[(size_t)(r * candidates + c) + 1]) != hipSuccess) { ms = 1e30f; (void)hipGetLastError(); }
            t = (r == 0 || ms < t) ? ms : t;
        }
        if (c == builtin) builtinMs = t;
        if (winner < 0 || t < best) { best = t; winner = c; }
    }
    if (ok && winner >= 0) {
        if (winner != builtin && !(best < kTuneMargin * builtinMs)) { winner = builtin; best = builtinMs; }
        for (int k = 0; k < 8; k++) ctx->tune[k] = kSets[winner % kFamilies][k];
        if (winner >= kFamilies) ctx->tune[5] = 32;
        ctx->tunedCandidate = winner;
        ctx->tuneCache.push_back({ctx->sceneSignatureNow, winner});
        logf(ctx, 4, "tune", "stage thresholds measured on this scene: candidate %d of %d (%.3f ms per %d-sample probe)", winner, candidates, best, kTuneProbeSamples);
    } else {
        (void)hipGetLastError();
        ctx->tunedCandidate = -1;
        logf(ctx, 2, "tune", "threshold probes failed; the per-kind values stay");
    }
    ctx->tunedScene = ctx->tunePendingScene;                    // measured (or not measurable): do not try again for this scene
    dropThresholdTuning(ctx);
    return true;
}

uint64_t listCapacity(const RtowContext_t* ctx, bool volumes);    // (defined with growHitList below)
inline bool triangleKind(uint32_t kind) { return kind == SCENE_KIND_TRIANGLES || kind == SCENE_KIND_TRIANGLES_TEXTURED; }
constexpr unsigned kTieWatchBusy = 4096;   // a watched launch of an all-triangle scene that lists more pixel-batches than this (8 workgroups render them) sends the scene to the exact-tie kernels

int launchSample(RtowContext ctx, const RtowSampleParams* p, const RtowAccumBuffers* in, const RtowAccumBuffers* out, void* diag,
                 hipStream_t stream, bool useCancelFlag, const ChainSpec* chain = nullptr)
{
    SampleKernelArgs a{};
    a.inColor = in->color; a.inNormal = in->normal; a.inAlbedo = in->albedo; a.inScw = in->sampleCountWeight;
    a.outColor = out->color; a.outNormal = out->normal; a.outAlbedo = out->albedo; a.outScw = out->sampleCountWeight;
    a.diagnostics = (uint8_t*)diag;
    a.diagnosticsStride = p->diagnosticsStride;
    a.sceneBlob = ctx->dScene;
    a.layout = ctx->scene.layout;
    a.ldsSceneBytes = ctx->ldsSceneBytes;
    a.ldsNodeCount = ctx->ldsNodeCount;
    a.ldsStackRows = ctx->ldsPlan.stackRows; a.ldsHistOffset = 0u; a.ldsFrontBytes = ctx->ldsPlan.frontBytes;      // (launches of the generic variants plan again below: their history rows)
    a.workCounter = ctx->dWorkCounter;
    a.cancelFlag = useCancelFlag ? ctx->hCancel : nullptr;
    a.overflowFlag = const_cast<uint32_t*>(ctx->hCancel) + 1;      // [1]: a ray beyond the hit-list capacity (grows: takeOverflow); [2]: more tied pixel-batches than the fix-up list holds (final)
    a.width = (int)p->size.x;
    a.height = (int)p->size.y;
    a.totalWork = (uint32_t)ownedRows(p) * (uint32_t)a.width;
    a.tilesPerRow = (RTOW_TICKET_TILES && a.width % (int)kTileW == 0) ? (uint32_t)a.width / kTileW : 0u;
    a.tiledPixels = a.tilesPerRow ? ((uint32_t)ownedRows(p) / kTileH) * kTileH * (uint32_t)a.width : 0u;
    a.sizeX = p->size.x; a.sizeY = p->size.y;
    a.sliceOffset = p->sliceOffset; a.sliceDivider = p->sliceDivider;
    a.seed = chain ? chain->seeds[0] : p->seed;
    a.chainCount = chain ? (uint32_t)chain->count : 1u;
    a.chainIndependent = (chain && chain->outs) ? 1 : 0;
    // slot / chainCount = (slot * groupRecip) >> 32 for every slot of a launch (chunkCount * chainCount <= 2^25); a count of one has no 32-bit reciprocal (2^32 + 1 would truncate
    // to 1): 2^32 - 1 is exact for slots below 2^32 - and groups of one batch are plain launches anyway (rtowSampleBatchGroupDevice)
    a.groupRecip = a.chainCount <= 1u ? 0xffffffffu : (uint32_t)((1ull << 32) / a.chainCount + 1ull);
    if (chain) {
        if (diag == nullptr && chain->diags) diag = chain->diags[0];
        a.diagnostics = (uint8_t*)diag;
    }
    a.view = p->view;
    a.environment = p->environment;
    a.sampleCountMin = p->sampleCountRange[0];
    a.sampleCountMax = p->sampleCountRange[1];
    a.traceDepth = p->traceDepth;
    a.subPixelJitter = p->subPixelJitter;
    a.extremaX = p->sampleCountWeightExtrema.x;
    a.extremaY = p->sampleCountWeightExtrema.y;
    a.refTree = (ctx->flags & RTOW_CONTEXT_REFERENCE_DIAGNOSTICS) ? ctx->dRefTree : nullptr;
    a.hitSpill = ctx->hitSpillEntries ? ctx->dHitSpill : nullptr;
    a.hitSpillEntries = ctx->hitSpillEntries;
    a.hitSpillStride = (uint32_t)ctx->cuCount * (uint32_t)kBlockThreads;
    a.texBlob = ctx->dTexBlob;
    a.texLayout = ctx->scene.texLayout;
    a.noiseColor = p->noiseColor;
    if (p->noiseColor == RTOW_NOISE_BLUE) {
        if (!ctx->dBlueNoise || p->noiseTextureIndex < 0 || (uint32_t)p->noiseTextureIndex >= ctx->blueTextureCount) return RTOW_ERROR_INVALID_VALUE;
        a.blueRowStride = ctx->blueRowStride;
        a.blueNoise = ctx->dBlueNoise + (size_t)p->noiseTextureIndex * ctx->blueRowStride * ctx->blueRowStride * 8u;
    } else if (p->noiseColor == RTOW_NOISE_SPATIOTEMPORAL_BLUE) {
        if (!ctx->dStbNoise || p->noiseTextureIndex < 0 || (uint32_t)p->noiseTextureIndex >= ctx->stbTextureCount) return RTOW_ERROR_INVALID_VALUE;
        const size_t texels = (size_t)ctx->stbRowStride * ctx->stbRowStride, all = texels * ctx->stbTextureCount, t = (size_t)p->noiseTextureIndex * texels;
        a.stbRowStride = ctx->stbRowStride;
        a.stbScalar = ctx->dStbNoise + t;                                    // 1 byte per texel
        a.stbVector2 = ctx->dStbNoise + all + t * 3;                         // RGB24
        a.stbCosineUnitVector3 = ctx->dStbNoise + all * 4 + t * 4;           // RGBA32
        a.stbUnitVector2 = ctx->dStbNoise + all * 8 + t * 3;                 // RGB24
        a.stbUnitVector3 = ctx->dStbNoise + all * 11 + t * 3;                // RGB24
    }
    a.cubemapData = ctx->dCubemap;
    a.cubemapHalfW = ctx->cubemap.faceWidth / 2; a.cubemapHalfH = ctx->cubemap.faceHeight / 2;                  // RT/Texture.cs:152-154
    a.cubemapW1 = ctx->cubemap.faceWidth - 1; a.cubemapH1 = ctx->cubemap.faceHeight - 1;
    a.cubemapPixelStride = ctx->cubemap.pixelStride; a.cubemapRowStride = ctx->cubemap.pixelStride * ctx->cubemap.faceWidth;   // :167
    a.cubemapFaceStride = ctx->cubemap.pixelStride * ctx->cubemap.faceWidth * ctx->cubemap.faceHeight;         // :168
    a.cubemapChannelType = ctx->cubemap.channelType;

    {
        // The variants for paths deeper than 16 (and the generic ones: 16-byte records, texture-driven noise, the per-sample policies beyond depth 8) keep the path-history codes
        // beyond the first eight in LDS rows: this launch's LDS is planned with them, and the scene image takes what is left (a scene that no longer fits whole keeps the top of
        // its tree there and reads the rest through L2, like any scene beyond LDS)
        const bool fullDiag = a.diagnostics && a.diagnosticsStride >= 16;
        const bool perSample = p->rngPolicy != RTOW_RNG_REFERENCE;
        int hw = historyWords(a.noiseColor, perSample, ctx->wideCodes, ctx->scene.layout.exactTies != 0, fullDiag, a.traceDepth);
        // (an all-triangle scene under the tie watch launches its rank-rule kernels first and its exact-tie kernels on the marked pixels, with the one plan: the wider of the two)
        if (ctx->scene.layout.exactTies) hw = std::max(hw, historyWords(a.noiseColor, perSample, ctx->wideCodes, false, fullDiag, a.traceDepth));
        if (hw == 32 && a.traceDepth > kHistoryInRegisters) {
            const LdsPlan plan = planLds(ctx->wideCodes, ctx->scene.layout, (uint32_t)(a.traceDepth - kHistoryInRegisters), ctx->ldsSceneBudget);
            a.ldsStackRows = plan.stackRows; a.ldsHistOffset = plan.histOffset; a.ldsFrontBytes = plan.frontBytes; a.ldsHistRows = plan.histRows;
            a.ldsSceneBytes = plan.sceneBytes; a.ldsNodeCount = plan.nodeCount;
            if (plan.histSpillRows) {
                // the rows that do not fit LDS (32-bit stack rows of a deep tree next to a deep trace depth; a scene kept whole in LDS): 2 bytes per lane and row in HBM
                const size_t stride = (size_t)ctx->cuCount * (size_t)kBlockThreads, need = (size_t)plan.histSpillRows * stride * sizeof(unsigned short);
                if (need > ctx->histSpillBytes) {
                    HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);          // a batch in flight may still use the smaller area
                    if (ctx->dHistSpill) (void)hipFree(ctx->dHistSpill);
                    ctx->dHistSpill = nullptr;
                    ctx->histSpillBytes = 0;
                    HIP_TRY(ctx, hipMalloc(&ctx->dHistSpill, need), RTOW_ERROR_MEMORY_ALLOCATION);
                    ctx->histSpillBytes = need;
                }
                a.histSpill = ctx->dHistSpill; a.histSpillRows = plan.histSpillRows; a.histSpillStride = (uint32_t)stride;
            }
        }
    }
    // scheduler thresholds (lane population a stage needs before it runs) and box-walk slice (RtowContextOptions.schedulerTune overrides)
    for (int i = 0; i < 8; i++) a.tune[i] = ctx->tune[i] < 1 ? 1 : ctx->tune[i];
    a.travSlice = ctx->tune[8] < 1 ? 1 : ctx->tune[8];
    // Lanes that wait for company at a pixel boundary (kernel: A.tune[7]; schedulerTune[7] bits 12 .. 15 override).  Measured (profiles/r06d_pixel_boundaries_in_company.json):
    // worth it only where boundaries are frequent - the reference host's 50 samples per batch: groups +2.5 % with 3 - 4 lanes, the per-sample policies' 16-sample units +1.6 % -
    // (the adaptive {1, 50} schedule +0.5 ... 1 %) and a loss of 1 - 3 % where a pixel takes hundreds of samples (the wait costs more than the shared instructions save):
    // so by the samples a unit of work takes at most
    const unsigned unitSamples = p->rngPolicy != RTOW_RNG_REFERENCE ? kSampleGroup : (a.sampleCountMax > a.sampleCountMin ? a.sampleCountMax : a.sampleCountMin);
    const int pixelGate = (ctx->regroupSide >> 12) & 15 ? (ctx->regroupSide >> 12) & 15 : (unitSamples <= 64u ? 4 : RTOW_PIXEL_GATE);
    a.tune[7] = pixelGate;
    // Lanes in a hurry (kernel: HURRY; twins of the static-sphere kind's generic reference-stream variants): a pixel that runs at more than this many rays per sample stops waiting for
    // company.  Batch groups run a pixel's batches side by side and keep every wave busy to the end: no bound, and the variants without the code.  Static spheres only: measured at
    // depth 32, same box (profiles/r06x_lanes_in_a_hurry.json) - cover scene +21 % (adaptive) / +33 % (chains), 10 000 spheres +24 %; in its first form (a bound on the batch's
    // rays) moving spheres with a lens -1.6 %, the 250 882-triangle mesh -8.6 % (a stage run for one lane costs the whole wave a memory round trip there).
    {
        // (exactly the launches launchByDiagGeo serves from a twin: every other variant reads tune[7] as the pixel gate alone)
        const bool records16 = a.diagnostics && a.diagnosticsStride >= 16;
        const bool twin = RTOW_URGENT_LANES && !a.chainIndependent && a.layout.sceneKind == SCENE_KIND_SPHERES && !a.layout.exactTies && !ctx->wideCodes && p->rngPolicy == RTOW_RNG_REFERENCE &&
                          a.noiseColor == RTOW_NOISE_WHITE && !(records16 && a.refTree) && historyWords(a.noiseColor, false, false, false, records16, a.traceDepth) == 32;
        const float urgentRays = !twin ? __builtin_inff() : (a.ldsSceneBytes == a.layout.totalBytes ? RTOW_URGENT_RAYS_PER_SAMPLE : RTOW_URGENT_RAYS_PER_SAMPLE_BEYOND_LDS);
        uint32_t bits;
        memcpy(&bits, &urgentRays, sizeof bits);
        if (!(urgentRays < __builtin_inff())) bits = 0u;                                  // no bound: the variants without the code (they read tune[7] as the pixel gate alone)
        a.tune[7] = (int32_t)((bits & 0xffffff00u) | (uint32_t)(pixelGate & 255));
    }
    const uint32_t ownedPixels = a.totalWork;
    if (ownedPixels == 0) {
        // a slice that owns no row (SliceOffset >= height): Execute returns for every index (JOBS/SampleBatchJob.cs:69-70) - nothing is
        // written, nothing is launched (a zero-sized grid is not a valid launch); the events still bracket "this batch"
        if (ctx->haveBatchDone) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->evBatchDone, 0), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipEventRecord(ctx->evStart, stream), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipEventRecord(ctx->evStop, stream), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipEventRecord(ctx->evBatchDone, stream), RTOW_ERROR_LAUNCH_FAILURE);
        ctx->haveBatchDone = true;
        ctx->haveTiming = true;
        return RTOW_SUCCESS;
    }
    a.groupsPerPixel = 1;
    a.xoroshiro = p->rngPolicy == RTOW_RNG_PER_SAMPLE_XOROSHIRO ? 1 : 0;
    if (p->rngPolicy != RTOW_RNG_REFERENCE) {
        // work units are (owned pixel, group of kSampleGroup samples); each leaves a record that fold_unit_records_kernel adds up
        uint32_t groups = (a.sampleCountMax > a.sampleCountMin ? a.sampleCountMax : a.sampleCountMin);
        groups = (groups + kSampleGroup - 1) / kSampleGroup;
        if (groups < 1) groups = 1;
        if ((uint64_t)ownedPixels * groups > 0x7fffffffull) return RTOW_ERROR_CAPACITY;
        a.groupsPerPixel = groups;
        a.totalWork = ownedPixels * groups;
        if ((size_t)a.totalWork > ctx->unitRecordCapacity) {
            if (ctx->dUnitRecords) (void)hipFree(ctx->dUnitRecords);
            ctx->dUnitRecords = nullptr;
            ctx->unitRecordCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dUnitRecords, (size_t)a.totalWork * 64u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->unitRecordCapacity = a.totalWork;
        }
        a.unitRecords = ctx->dUnitRecords;
    }
    // ---- nearest-hit ties under the rank rule: sphere kinds of more than 16 entities, reference stream - a pixel that meets two different spheres at
    // bit-identical distance at a nearest hit is listed instead of stored, and the exact-tie kernel of the same kind renders the list in a second, tiny launch
    // All-triangle scenes whose exact-tie kernels were chosen for their size alone (no triangle twice: SceneLayout.tieWatchOk) are watched too: the rank-rule kernels trace the
    // frame, the exact-tie kernels the marked pixels - unless the scene has shown that it ties often (triWatchOff: the flag below, or a list that overflowed)
    if (ctx->hCancel[3] != 0u) { ctx->hCancel[3] = 0u; if (triangleKind(ctx->scene.layout.sceneKind)) { ctx->triWatchOff = true; logf(ctx, 3, "rtow", "this scene's nearest hits tie often: exact-tie kernels from now on"); } }
    const bool sphereWatch = ctx->scene.layout.sceneKind <= SCENE_KIND_SPHERES_MOTION && !ctx->scene.layout.exactTies && ctx->scene.entityCount > 16;
    const bool triWatch = triangleKind(ctx->scene.layout.sceneKind) && ctx->scene.layout.exactTies && ctx->scene.layout.tieWatchOk && !ctx->triWatchOff &&
                          !(ctx->flags & RTOW_CONTEXT_EXACT_TIES_ALWAYS);
    const bool tieWatch = (sphereWatch || triWatch) && !(ctx->flags & RTOW_CONTEXT_EXACT_TIES_NEVER) && p->rngPolicy == RTOW_RNG_REFERENCE;
    if (tieWatch && triWatch) a.layout.exactTies = 0u;               // this launch goes through the rank-rule kernels of the kind; the fix-up launch below sets the bit again
    // in place: an output buffer that is also the input buffer (a chain's later batches always read the outputs, but they read what THIS launch stored: only batch 0's inputs count)
    const bool inPlace = in->color == out->color || in->normal == out->normal || in->albedo == out->albedo || in->sampleCountWeight == out->sampleCountWeight;
    if (tieWatch) {
        if (!ctx->dTieRedo) HIP_TRY(ctx, hipMalloc(&ctx->dTieRedo, (4u + (size_t)kTieRedoCapacity) * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
        const uint32_t most = (uint32_t)std::min<uint64_t>((uint64_t)ctx->scene.entityCount, listCapacity(ctx, false));
        const uint32_t entries = most > (uint32_t)kLocalHitEntries ? most - (uint32_t)kLocalHitEntries : 0u;
        if (entries > ctx->redoSpillEntries) {
            if (ctx->dRedoSpill) (void)hipFree(ctx->dRedoSpill);
            ctx->dRedoSpill = nullptr;
            ctx->redoSpillEntries = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dRedoSpill, (size_t)entries * kTieRedoBlocks * kBlockThreads * sizeof(uint4)), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->redoSpillEntries = entries;
        }
        const size_t framePixels = (size_t)a.width * (size_t)a.height;
        const size_t words = (framePixels + 31u) / 32u;
        if (words > ctx->tieBitsWords) {
            if (ctx->dTieBits) (void)hipFree(ctx->dTieBits);
            ctx->dTieBits = nullptr;
            ctx->tieBitsWords = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dTieBits, words * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->tieBitsWords = words;
        }
        if (inPlace && framePixels > ctx->tieInputPixels) {
            if (ctx->dTieInputs) (void)hipFree(ctx->dTieInputs);
            ctx->dTieInputs = nullptr;
            ctx->tieInputPixels = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dTieInputs, framePixels * 11u * sizeof(float)), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->tieInputPixels = framePixels;
        }
    }

    // ---- launch geometry: one persistent 1024-lane workgroup per CU (four waves per SIMD).  Smaller workgroups for launches that own about one pixel
    // per resident lane were built, measured and removed: results never depended on it.
    a.wideCodes = ctx->wideCodes ? 1 : 0;
    a.blockThreads = kBlockThreads;
    int blocks = (int)((a.totalWork + (uint32_t)a.blockThreads - 1) / (uint32_t)a.blockThreads);
    if (blocks > ctx->cuCount) blocks = ctx->cuCount; // persistent: one workgroup per CU
    if (blocks < 1) blocks = 1;

    // Batches of one context share its ticket counter, cost map, chunk order and candidate lists, and each one consumes what the
    // previous one produced: whatever stream this batch was given, it starts after everything the previous batch enqueued.
    if (ctx->haveBatchDone) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->evBatchDone, 0), RTOW_ERROR_LAUNCH_FAILURE);

    // ---- camera-ray candidate lists: one conservative beam walk per pixel, reused by all its samples (and by later batches of the same view) ----
    if (!(ctx->flags & RTOW_CONTEXT_NO_CAMERA_RAY_LISTS)) {
        const size_t pixels = (size_t)a.width * (size_t)a.height;
        const size_t listBytes = pixels * sizeof(uint4);      // 8 x 16-bit node codes per pixel; 4 x 32-bit with wide codes
        if (listBytes > ctx->pixCandCapacity) {
            if (ctx->dPixCand) (void)hipFree(ctx->dPixCand);
            ctx->dPixCand = nullptr;
            ctx->pixCandCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dPixCand, listBytes), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->pixCandCapacity = listBytes;
            ctx->pixCandValid = false;
        }
        const bool same = ctx->pixCandValid && ctx->pixCandScene == ctx->sceneSerial && memcmp(&ctx->pixCandView, &a.view, sizeof(RtowView)) == 0 &&
                          ctx->pixCandW == a.width && ctx->pixCandH == a.height && ctx->pixCandOff == a.sliceOffset && ctx->pixCandDiv == a.sliceDivider &&
                          ctx->pixCandJitter == (a.subPixelJitter ? 1 : 0);
        if (!same) {
            SampleKernelArgs perPixel = a;                       // the lists are per pixel whatever the work units are
            perPixel.totalWork = ownedPixels;
            HIP_TRY(ctx, launchPrimaryCandidates(perPixel, ctx->dPixCand, stream), RTOW_ERROR_LAUNCH_FAILURE);
            ctx->pixCandValid = true;
            ctx->pixCandScene = ctx->sceneSerial;
            ctx->pixCandView = a.view;
            ctx->pixCandW = a.width; ctx->pixCandH = a.height; ctx->pixCandOff = a.sliceOffset; ctx->pixCandDiv = a.sliceDivider;
            ctx->pixCandJitter = a.subPixelJitter ? 1 : 0;
        }
        a.pixelCandidates = ctx->dPixCand;
    }

    // ---- chunk order: most expensive 64-pixel chunks first, from the ray counts of the previous launch (or of a probe) ----
    a.chunkCount = (a.totalWork + 63u) / 64u;
    const bool wantOrder = a.chunkCount >= (uint32_t)(4 * ctx->cuCount) && !(ctx->flags & RTOW_CONTEXT_NO_CHUNK_ORDER);   // tiny frames: not worth it
    // ---- and which pixels share a chunk (a wave): the pixels of a super-tile of regroupSide x regroupSide tiles sorted by the ray counts of the previous launch and dealt out
    // 64 at a time (regroup_tickets_kernel), re-sorted behind every launch like the order.  Reference stream only (per-sample units are alike by construction).
    // schedulerTune[7]: 1 = no map; 3 = the tiles as they are, each tile's tickets most expensive first (regroupSide 1 below); 2 / 4 / 8 (+ 16 x mode) = super-tiles of that many tiles
    const unsigned knob = (unsigned)(ctx->regroupSide & 15);
    const unsigned regroupSide = knob == 3u ? 1u : (knob == 2u || knob == 4u || knob == 8u) ? knob : 0u;
    // (development: schedulerTune[7] = side + 16 * mode; mode 0 sorts by the ray count itself, 1 by sky / not sky, 2 by four classes of rays per sample - pixels of a class in tile order)
    const unsigned regroupMode = ((unsigned)ctx->regroupSide >> 4) & 3u;
    // batch groups: (chunk, batch) slots a wave reserves at a time (schedulerTune[7] bits 8 .. 11 override the default for A/B runs)
    a.slotBlock = (((unsigned)ctx->regroupSide >> 8) & 15u) ? (((unsigned)ctx->regroupSide >> 8) & 15u) : kGroupSlotBlock;
    if ((((unsigned)ctx->regroupSide >> 8) & 15u) == 0u) {
        // Reserving several (chunk, batch) slots per pull pays where a wave has dozens of them to work through (a whole frame's group: 79 slots per wave, +1.8 %); a launch that
        // owns a few slots per wave - the sub-batches of an 8-way row slice: 7.9 - must balance with single slots (profiles/r06j_partitions_c2.json: 10.9 ms per step against the
        // 7.9 of round 5's single slots)
        const uint64_t slotsPerWave = (uint64_t)a.chunkCount * a.chainCount / ((uint64_t)blocks * (uint64_t)(kBlockThreads / 64));
        if (slotsPerWave < 32u) a.slotBlock = 1u;
        else if (slotsPerWave < 64u && a.slotBlock > 2u) a.slotBlock = 2u;
    }
    auto regroupClasses = [&](unsigned floorCost, unsigned out[3]) {
        if (regroupSide == 1u) { out[0] = regroupMode == 0u ? 0u : (2u << regroupMode); out[1] = out[2] = 0u; return; }      // tile order: levels of the tile's cost range (mode 1 / 2 / 3: 4 / 8 / 16), 0 = by the ray count itself
        out[0] = regroupMode == 0u ? 0u : floorCost; out[1] = regroupMode >= 2u ? (floorCost * 11u) / 4u : 0xffffffffu; out[2] = regroupMode >= 2u ? floorCost * 4u : 0xffffffffu;
    };
    const bool wantMap = wantOrder && regroupSide >= 1u && a.tiledPixels != 0u && !a.unitRecords;
    const unsigned tileRows = a.tilesPerRow ? a.tiledPixels / (64u * a.tilesPerRow) : 0u;
    if (wantOrder) {
        if (a.chunkCount > ctx->chunkCapacity) {
            if (ctx->dChunkCost) { (void)hipFree(ctx->dChunkCost); (void)hipFree(ctx->dChunkOrder); (void)hipFree(ctx->dPixelCost); (void)hipFree(ctx->dTicketMap); }
            ctx->dChunkCost = ctx->dChunkOrder = nullptr;
            ctx->dPixelCost = nullptr;
            ctx->dTicketMap = nullptr;
            ctx->chunkCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dChunkCost, (2 * (size_t)a.chunkCount + kChunkOrderScratchWords) * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            HIP_TRY(ctx, hipMalloc(&ctx->dChunkOrder, a.chunkCount * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            HIP_TRY(ctx, hipMalloc(&ctx->dPixelCost, (size_t)a.chunkCount * 64 * sizeof(unsigned short)), RTOW_ERROR_MEMORY_ALLOCATION);
            HIP_TRY(ctx, hipMalloc(&ctx->dTicketMap, (size_t)a.chunkCount * 64 * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->chunkCapacity = a.chunkCount;
            ctx->orderValid = false;
        }
        if (ctx->orderW != a.width || ctx->orderH != a.height || ctx->orderOff != a.sliceOffset || ctx->orderDiv != a.sliceDivider || ctx->orderGroups != a.groupsPerPixel ||
            ctx->orderMapped != wantMap) ctx->orderValid = false;
        a.pixelCost = ctx->dPixelCost;
        a.ticketMap = wantMap ? ctx->dTicketMap : nullptr;
        bool haveOrder = true;
        if (!ctx->orderValid && a.unitRecords) {
            // per-sample units are small and alike: the first batch simply runs in natural order and records the map for the next
            HIP_TRY(ctx, hipMemsetAsync(ctx->dPixelCost, 0, (size_t)a.chunkCount * 64 * sizeof(unsigned short), stream), RTOW_ERROR_LAUNCH_FAILURE);
            haveOrder = false;
            ctx->orderValid = true;
            ctx->orderW = a.width; ctx->orderH = a.height; ctx->orderOff = a.sliceOffset; ctx->orderDiv = a.sliceDivider; ctx->orderGroups = a.groupsPerPixel;
            ctx->orderMapped = false;
        }
        if (!ctx->orderValid) {
            // no cost map yet for this frame configuration: a 1-sample-per-pixel probe of the same kernel (stores nothing else)
            SampleKernelArgs probe = a;
            probe.probeOnly = 1;
            probe.chunkOrder = nullptr;
            probe.ticketMap = nullptr;                           // the probe runs over the tiles themselves; the map starts from them
            if (wantMap) HIP_TRY(ctx, launchInitTicketMap(ctx->dTicketMap, a.tiledPixels, stream), RTOW_ERROR_LAUNCH_FAILURE);
            probe.cancelFlag = nullptr;
            probe.chainCount = 1;                                // one pass over the pixels, whatever the launch it prepares
            HIP_TRY(ctx, hipMemsetAsync(ctx->dWorkCounter, 0, sizeof(unsigned int), stream), RTOW_ERROR_LAUNCH_FAILURE);
            HIP_TRY(ctx, hipMemsetAsync(ctx->dPixelCost, 0, (size_t)a.chunkCount * 64 * sizeof(unsigned short), stream), RTOW_ERROR_LAUNCH_FAILURE);   // the last chunk's tail
            HIP_TRY(ctx, launchSampleBatch(probe, blocks, stream), RTOW_ERROR_LAUNCH_FAILURE);
            if (wantMap) { unsigned cls[3]; regroupClasses(1u, cls); HIP_TRY(ctx, launchRegroupTickets(ctx->dPixelCost, ctx->dTicketMap, a.tilesPerRow, tileRows, regroupSide, cls, stream), RTOW_ERROR_LAUNCH_FAILURE); }
            HIP_TRY(ctx, launchBuildChunkOrder(ctx->dPixelCost, ctx->dChunkCost, a.chunkCount, ctx->dChunkOrder, 0, stream), RTOW_ERROR_LAUNCH_FAILURE);
            ctx->orderValid = true;
            ctx->orderW = a.width; ctx->orderH = a.height; ctx->orderOff = a.sliceOffset; ctx->orderDiv = a.sliceDivider; ctx->orderGroups = a.groupsPerPixel;
            ctx->orderMapped = wantMap;
        }
        a.chunkOrder = haveOrder ? ctx->dChunkOrder : nullptr;

        // ---- stage thresholds: measured once per scene on this batch's own kernel, frame and view (see kTuneProbeSamples) ----
        // Never waited for: the probes are enqueued in front of a batch, timed with events, and a LATER call that finds the last event complete reads them
        // and switches the thresholds (scheduling only: no result depends on when that happens).  Until then the kernel kind's built-in values run.
        for (int k = 0; k < 7; k++) a.tune[k] = ctx->tune[k] < 1 ? 1 : ctx->tune[k];
        if (ctx->tunePending && ctx->tunePendingScene == ctx->sceneSerial && finishThresholdTuning(ctx, /*wait*/ false))
            for (int k = 0; k < 7; k++) a.tune[k] = ctx->tune[k] < 1 ? 1 : ctx->tune[k];
        ctx->sppSinceUpload += (uint64_t)a.chainCount * (a.sampleCountMax > a.sampleCountMin ? a.sampleCountMax : a.sampleCountMin);
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;                   // a stream that is being captured into a graph cannot carry the event pairs: no measurement then
        if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
        // worth it only where the scene is rendered for longer than the probes take (3 or 6 candidates x 3 repeats x 4 samples per pixel: 40 - 76 samples per
        // pixel): a one-shot render of a few samples per pixel (BASELINE.json configs[0]: 8) must not pay several times its own work first
        const bool worthIt = ctx->sppSinceUpload >= kTuneMinSamplesPerPixel;
        if (!ctx->userTune && !(ctx->flags & RTOW_CONTEXT_NO_THRESHOLD_TUNING) && ctx->tunedScene != ctx->sceneSerial && !ctx->tunePending && !a.unitRecords && haveOrder &&
            worthIt && capturing == hipStreamCaptureStatusNone) {
            constexpr int kFamilies = 3;
            static const int kSets[kFamilies][9] = {{RTOW_DEFAULT_TUNE}, {RTOW_GENERAL_TUNE}, {RTOW_GENERAL_TUNE_2}};     // (the ninth value, the walk slice, is set by rtowUploadScene)
            const bool volumes = a.layout.sceneKind == SCENE_KIND_VOLUMES || a.layout.sceneKind == SCENE_KIND_VOLUMES_TEXTURED;
            const int candidates = volumes ? 2 * kFamilies : kFamilies;                      // volume kinds: each family also with the volume stage from half of the live lanes
            const int launches = candidates * kTuneProbeRepeats;
            ctx->tuneEvents.assign((size_t)launches + 1, nullptr);
            bool ok = true;
            for (auto& e : ctx->tuneEvents) ok = ok && hipEventCreate(&e) == hipSuccess;
            if (!ctx->dProbeSink) ok = ok && hipMalloc(&ctx->dProbeSink, 64) == hipSuccess;
            SampleKernelArgs probe = a;
            probe.probeOnly = kTuneProbeSamples;
            probe.pixelCost = nullptr;                           // the cost map stays the cost probe's (or the previous batch's)
            probe.cancelFlag = nullptr;
            probe.overflowFlag = ctx->dProbeSink;                // a probe's ray beyond the hit-list capacity is not the batch's (which may trace fewer samples than a probe)
            probe.chainCount = 1;
            probe.chainIndependent = 0;
            // one untimed probe first: clocks, L2 and the instruction cache are warm before the first timed one
            for (int k = 0; k < 7; k++) probe.tune[k] = kSets[0][k];
            if (ok) ok = hipMemsetAsync(ctx->dWorkCounter, 0, sizeof(unsigned int), stream) == hipSuccess && launchSampleBatch(probe, blocks, stream) == hipSuccess;
            if (ok) ok = hipEventRecord(ctx->tuneEvents[0], stream) == hipSuccess;
            for (int l = 0; ok && l < launches; l++) {
                const int c = l % candidates;
                for (int k = 0; k < 7; k++) probe.tune[k] = kSets[c % kFamilies][k];
                if (c >= kFamilies) probe.tune[5] = 32;
                ok = hipMemsetAsync(ctx->dWorkCounter, 0, sizeof(unsigned int), stream) == hipSuccess && launchSampleBatch(probe, blocks, stream) == hipSuccess &&
                     hipEventRecord(ctx->tuneEvents[(size_t)l + 1], stream) == hipSuccess;
            }
            if (ok) {
                ctx->tunePending = true;
                ctx->tunePendingScene = ctx->sceneSerial;
                ctx->tuneCandidates = candidates;
                ctx->tuneBuiltin = a.layout.sceneKind <= SCENE_KIND_SPHERES_MOTION ? 0 : 1;       // what rtowUploadScene set for this kernel kind
            } else {
                (void)hipGetLastError();
                dropThresholdTuning(ctx);
                ctx->tunedCandidate = -1;
                ctx->tunedScene = ctx->sceneSerial;             // not measurable: do not try again for this scene
                logf(ctx, 2, "tune", "threshold probes failed; the per-kind values stay");
            }
        }
    }

    {
        // Plain or chained launches with paths deeper than 16 segments are bound by their slowest pixels (see "lanes in a hurry" above), not by lane occupancy: a static-sphere scene
        // that is whole in LDS runs them with REGEN from a quarter of the live lanes, HIT from 3/8, and the walk's hand-over at 4 candidates (profiles/r06u_deep_plain_launch_thresholds.json:
        // +7 ... +10 %; the same values cost batch groups 2.8 %, launches at depth <= 16 1.3 ... 3 %, moving spheres 2 %, and 10 000 spheres - next to the lanes in a hurry - 2 %)
        static const int kDefault[9] = {RTOW_DEFAULT_TUNE};
        if (!ctx->userTune && !a.chainIndependent && a.traceDepth > 16 && p->rngPolicy == RTOW_RNG_REFERENCE && a.layout.sceneKind == SCENE_KIND_SPHERES && a.ldsSceneBytes == a.layout.totalBytes &&
            a.tune[0] == kDefault[0] && a.tune[3] == kDefault[3] && a.tune[6] == kDefault[6]) {
            a.tune[0] = 16; a.tune[3] = 24; a.tune[6] = 4;
        }
    }
    if (a.chainCount > 1u && a.chainIndependent) {
        // a batch group: nothing is handed over between its batches; only the per-batch table (seed, diagnostics, outputs)
        if (!ctx->dChainBatches) HIP_TRY(ctx, hipMalloc(&ctx->dChainBatches, sizeof(ChainBatch) * kMaxChain), RTOW_ERROR_MEMORY_ALLOCATION);
        ChainBatch table[kMaxChain] = {};
        for (int b = 0; b < chain->count; b++) {
            table[b].seed = chain->seeds[b];
            table[b].diagnostics = chain->diags ? (uint8_t*)chain->diags[b] : nullptr;
            table[b].outColor = chain->outs[b].color; table[b].outNormal = chain->outs[b].normal; table[b].outAlbedo = chain->outs[b].albedo; table[b].outScw = chain->outs[b].sampleCountWeight;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dChainBatches, table, sizeof(ChainBatch) * (size_t)chain->count, hipMemcpyHostToDevice, stream), RTOW_ERROR_LAUNCH_FAILURE);
        a.chainBatches = ctx->dChainBatches;
    } else if (a.chainCount > 1u) {
        // per-chunk hand-off counters of the chain: pixels stored so far (all batches); batch b of a chunk waits for b x its pixels
        if (a.chunkCount > ctx->chunkDoneCapacity) {
            if (ctx->dChunkDone) (void)hipFree(ctx->dChunkDone);
            if (ctx->dXcdState) (void)hipFree(ctx->dXcdState);
            ctx->dChunkDone = nullptr;
            ctx->dXcdState = nullptr;
            ctx->chunkDoneCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dChunkDone, (size_t)a.chunkCount * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            HIP_TRY(ctx, hipMalloc(&ctx->dXcdState, sizeof(XcdState) + (size_t)kMaxXcds * a.chunkCount * sizeof(unsigned)), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->chunkDoneCapacity = a.chunkCount;
        }
        HIP_TRY(ctx, hipMemsetAsync(ctx->dChunkDone, 0, (size_t)a.chunkCount * sizeof(unsigned), stream), RTOW_ERROR_LAUNCH_FAILURE);
        a.chunkDone = ctx->dChunkDone;
        // chunk ownership per XCD: counters zero, list entries "not written yet" (the kernel indexes the lists with THIS launch's chunkCount)
        HIP_TRY(ctx, hipMemsetAsync(ctx->dXcdState, 0, sizeof(XcdState), stream), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemsetAsync(ctx->dXcdState + sizeof(XcdState), 0xff, (size_t)kMaxXcds * a.chunkCount * sizeof(unsigned), stream), RTOW_ERROR_LAUNCH_FAILURE);
        a.xcdState = reinterpret_cast<XcdState*>(ctx->dXcdState);
        // what differs between the chain's batches, indexed per lane by the kernel: a small table in device memory, written in stream order
        // (the previous chain's kernel may still be reading its own table: this copy is enqueued behind it)
        if (!ctx->dChainBatches) HIP_TRY(ctx, hipMalloc(&ctx->dChainBatches, sizeof(ChainBatch) * kMaxChain), RTOW_ERROR_MEMORY_ALLOCATION);
        ChainBatch table[kMaxChain] = {};
        for (int b = 0; b < chain->count; b++) { table[b].seed = chain->seeds[b]; table[b].diagnostics = chain->diags ? (uint8_t*)chain->diags[b] : nullptr; }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dChainBatches, table, sizeof(ChainBatch) * (size_t)chain->count, hipMemcpyHostToDevice, stream), RTOW_ERROR_LAUNCH_FAILURE);
        a.chainBatches = ctx->dChainBatches;
    }
    const float* redoIn[4] = {a.inColor, a.inNormal, a.inAlbedo, a.inScw};
    if (tieWatch) {
        const size_t framePixels = (size_t)a.width * (size_t)a.height;
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTieRedo, 0, 4u * sizeof(unsigned), stream), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTieBits, 0, ((framePixels + 31u) / 32u) * sizeof(unsigned), stream), RTOW_ERROR_LAUNCH_FAILURE);
        if (inPlace) {
            // the fix-up launch renders a marked pixel again from the launch's inputs, which an in-place launch overwrites: they are copied first (44 B per pixel through HBM,
            // ~0.05 ms at 1920 x 1080 against a batch's tens of milliseconds)
            static const size_t comps[4] = {4, 3, 3, 1};
            float* at = ctx->dTieInputs;
            const size_t rows = (size_t)ownedRows(p);                // the rows this launch writes (row % SliceDivider == SliceOffset), at their places in the frame
            for (int k = 0; k < 4; k++) {
                const size_t rowBytes = (size_t)a.width * comps[k] * sizeof(float), first = (size_t)a.sliceOffset * rowBytes, pitch = (size_t)a.sliceDivider * rowBytes;
                HIP_TRY(ctx, hipMemcpy2DAsync((uint8_t*)at + first, pitch, (const uint8_t*)redoIn[k] + first, pitch, rowBytes, rows, hipMemcpyDeviceToDevice, stream), RTOW_ERROR_LAUNCH_FAILURE);
                redoIn[k] = at;
                at += framePixels * comps[k];
            }
        }
        a.tieBits = ctx->dTieBits;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->dWorkCounter, 0, sizeof(unsigned int), stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipEventRecord(ctx->evStart, stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, launchSampleBatch(a, blocks, stream), RTOW_ERROR_LAUNCH_FAILURE);
    if (tieWatch) {
        // the fix-up: marked pixels -> list -> the exact-tie kernel of the same kind over the list (almost always empty: that kernel then leaves before it stages the scene).
        // A chain's pixel is listed once and carried through all its batches; a group's once per batch.
        const size_t framePixels = (size_t)a.width * (size_t)a.height;
        HIP_TRY(ctx, launchCollectTiedPixels(ctx->dTieBits, (unsigned)((framePixels + 31u) / 32u), ctx->dTieRedo, kTieRedoCapacity, a.chainIndependent ? a.chainCount : 1u, a.overflowFlag + 1,
                                             triWatch ? kTieWatchBusy : 0xffffffffu, stream),
                RTOW_ERROR_LAUNCH_FAILURE);
        SampleKernelArgs r = a;
        r.layout.exactTies = 1u;
        r.redoMode = 1;
        r.tune[7] &= 255;                      // (the exact-tie kernels have no lanes in a hurry: the pixel gate alone)
        r.tieBits = nullptr;
        r.tieRedo = ctx->dTieRedo;
        r.tieRedoCapacity = kTieRedoCapacity;
        r.inColor = redoIn[0]; r.inNormal = redoIn[1]; r.inAlbedo = redoIn[2]; r.inScw = redoIn[3];
        r.pixelCost = nullptr;
        r.chunkOrder = nullptr;
        r.pixelCandidates = a.pixelCandidates;
        r.hitSpill = ctx->redoSpillEntries ? ctx->dRedoSpill : nullptr;
        r.hitSpillEntries = ctx->redoSpillEntries;
        r.hitSpillStride = (uint32_t)kTieRedoBlocks * (uint32_t)kBlockThreads;
        HIP_TRY(ctx, hipMemsetAsync(ctx->dWorkCounter, 0, sizeof(unsigned int), stream), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, launchSampleBatch(r, kTieRedoBlocks < ctx->cuCount ? kTieRedoBlocks : ctx->cuCount, stream), RTOW_ERROR_LAUNCH_FAILURE);
    }
    if (a.unitRecords) HIP_TRY(ctx, launchFoldUnitRecords(a, stream), RTOW_ERROR_LAUNCH_FAILURE);   // inside the timed region: part of the batch
    HIP_TRY(ctx, hipEventRecord(ctx->evStop, stream), RTOW_ERROR_LAUNCH_FAILURE);
    // refresh the order for the next batch from what this one measured (same stream, after the timed kernel)
    if (wantMap) { unsigned cls[3]; regroupClasses(std::max(1u, a.sampleCountMin), cls); HIP_TRY(ctx, launchRegroupTickets(ctx->dPixelCost, ctx->dTicketMap, a.tilesPerRow, tileRows, regroupSide, cls, stream), RTOW_ERROR_LAUNCH_FAILURE); }
    // (development: schedulerTune[7] + 64 orders the chunks by their TOTAL ray count instead of by their most expensive pixel)
    if (wantOrder) HIP_TRY(ctx, launchBuildChunkOrder(ctx->dPixelCost, ctx->dChunkCost, a.chunkCount, ctx->dChunkOrder, (ctx->regroupSide & 64) ? 0 : 1, stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipEventRecord(ctx->evBatchDone, stream), RTOW_ERROR_LAUNCH_FAILURE);
    ctx->haveBatchDone = true;
    ctx->haveTiming = true;
    return RTOW_SUCCESS;
}

// Most surfaces one ray may meet where whole hit lists are kept: the caller's bound if it gave one, else a default that GROWS (growHitList) - the
// reference's list grows on the heap without bound (UTIL/HybridCollections.cs:22-36,65-71).
uint64_t listCapacity(const RtowContext_t* ctx, bool volumes)
{
    if (ctx->hitListCapacity) return ctx->hitListCapacity;
    return std::max<uint64_t>(volumes ? kDefaultHitListCapacity : kDefaultTieListCapacity, ctx->grownListCapacity);
}

// The spill area behind the lanes' own 24 entries, for a scene of this kind and size.  An entity yields at most two hits per ray (a volume hull's entry and exit,
// JOBS/SampleBatchJob.cs:457-469), one in scenes without volumes, so that bound - capped by listCapacity - is all a scene can need.
int sizeHitSpill(RtowContext ctx, uint32_t sceneKind, bool exactTies, int entityCount)
{
    const bool volumes = sceneKind == SCENE_KIND_VOLUMES || sceneKind == SCENE_KIND_VOLUMES_TEXTURED;
    uint64_t most = (volumes || exactTies) ? (uint64_t)entityCount * (volumes ? 2u : 1u) : 0u;
    const uint64_t cap = listCapacity(ctx, volumes);
    if (most > cap) most = cap;
    const uint32_t entries = most > (uint64_t)kLocalHitEntries ? (uint32_t)(most - kLocalHitEntries) : 0u;
    if (entries > ctx->hitSpillCapacity) {
        if (ctx->dHitSpill) (void)hipFree(ctx->dHitSpill);
        ctx->dHitSpill = nullptr;
        ctx->hitSpillCapacity = 0;
        ctx->hitSpillEntries = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dHitSpill, (size_t)entries * (size_t)ctx->cuCount * kBlockThreads * sizeof(uint4)), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->hitSpillCapacity = entries;
    }
    ctx->hitSpillEntries = entries;
    return RTOW_SUCCESS;
}

// A batch met a ray with more surfaces than the lists hold.  Unless the caller fixed the capacity, double it - up to what the scene can produce at all and to a
// quarter of the device's free memory - so that the batch, issued again, has room.  Called with every batch of the context finished (all callers have just waited).
bool growHitList(RtowContext ctx)
{
    if (ctx->hitListCapacity || !ctx->haveScene) return false;
    const uint32_t kind = ctx->scene.layout.sceneKind;
    const bool volumes = kind == SCENE_KIND_VOLUMES || kind == SCENE_KIND_VOLUMES_TEXTURED;
    const uint64_t bound = (uint64_t)ctx->scene.entityCount * (volumes ? 2u : 1u);
    const uint64_t now = listCapacity(ctx, volumes);
    if (now >= bound) return false;                                                      // the lists already hold everything the scene has
    const uint64_t next = std::min<uint64_t>(bound, now * 2u);
    if (volumes || ctx->scene.layout.exactTies) {
        size_t freeBytes = 0, totalBytes = 0;
        if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        const uint64_t held = (uint64_t)ctx->hitSpillCapacity * (uint64_t)ctx->cuCount * kBlockThreads * sizeof(uint4);
        const uint64_t need = (next - kLocalHitEntries) * (uint64_t)ctx->cuCount * kBlockThreads * sizeof(uint4);
        if (need > held && need > (freeBytes + held) / 4u) return false;
    }
    const uint32_t before = ctx->grownListCapacity;
    ctx->grownListCapacity = (uint32_t)next;
    if (sizeHitSpill(ctx, kind, ctx->scene.layout.exactTies != 0, ctx->scene.entityCount) != RTOW_SUCCESS) {
        (void)hipGetLastError();
        ctx->grownListCapacity = before;
        (void)sizeHitSpill(ctx, kind, ctx->scene.layout.exactTies != 0, ctx->scene.entityCount);
        return false;
    }
    return true;                                                                          // (the fix-up launch's own, small area follows at the next launch: launchSample)
}

// A ray met more surfaces than the context's hit-list capacity (volume scenes and exact-tie kernels keep every hit of a ray): the batch's result is not the
// reference's.  With RtowContextOptions.hitListCapacity == 0 the capacity has doubled by the time this returns (growHitList): the host-buffer calls then run the
// batch again themselves, a device-resident caller issues it again (from inputs the batch did not overwrite).
// The flag is sticky: it is set by the kernel and cleared only here, so it covers every batch enqueued since the last report
// (rtowSampleBatch, a cancellable rtowSampleBatchDevice, rtowGetBatchStatus, rtowSynchronize).
int takeOverflow(RtowContext ctx)
{
    if (ctx->hCancel[2] != 0u) {
        // not a hit list: a launch marked more pixel-batches for the tie fix-up pass than its list holds (kTieRedoCapacity = 2^20; a scene of coinciding spheres that
        // did not go to the exact-tie kernels).  Growing the hit lists would not help and running the batch again would overflow again: the error is final
        ctx->hCancel[2] = 0u;
        ctx->hCancel[1] = 0u;
        if (ctx->haveScene && triangleKind(ctx->scene.layout.sceneKind) && !ctx->triWatchOff) {
            // an all-triangle scene that ties over whole regions (coplanar layers): its exact-tie kernels need no list - the batch, issued again, runs on them
            ctx->triWatchOff = true;
            ctx->hCancel[3] = 0u;
            ctx->overflowGrew = true;
            logf(ctx, 3, "rtow", "more than %u pixel-batches of one launch met nearest-hit ties: results of this batch are invalid; the scene runs on the exact-tie kernels from now on", kTieRedoCapacity);
            return RTOW_ERROR_CAPACITY;
        }
        ctx->overflowGrew = false;
        logf(ctx, 2, "rtow", "more than %u pixel-batches of one launch met nearest-hit ties: results of this batch are invalid (use RTOW_CONTEXT_EXACT_TIES_ALWAYS for this scene)", kTieRedoCapacity);
        return RTOW_ERROR_CAPACITY;
    }
    if (ctx->hCancel[1] == 0u) return RTOW_SUCCESS;
    ctx->hCancel[1] = 0u;
    const bool volumes = ctx->scene.layout.sceneKind == SCENE_KIND_VOLUMES || ctx->scene.layout.sceneKind == SCENE_KIND_VOLUMES_TEXTURED;
    const uint32_t was = (uint32_t)listCapacity(ctx, volumes);
    ctx->overflowGrew = growHitList(ctx);
    if (ctx->overflowGrew) logf(ctx, 3, "rtow", "a ray met more than %u surfaces: results of this batch are invalid; the hit-list capacity is now %u", was, (uint32_t)listCapacity(ctx, volumes));
    else logf(ctx, 2, "rtow", "a ray met more surfaces than the hit-list capacity of this scene (%u; RtowContextOptions.hitListCapacity): results of this batch are invalid", was);
    return RTOW_ERROR_CAPACITY;
}

// Block until the stop event completes while mirroring the caller's cancellation byte into the device-visible flag.
int waitWithCancel(RtowContext ctx, const volatile uint8_t* cancel)
{
    bool cancelled = false;
    for (;;) {
        const hipError_t q = hipEventQuery(ctx->evStop);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            logf(ctx, 2, "hip", "sample kernel failed: %s", hipGetErrorString(q));
            ctx->hCancel[1] = 0u;
            ctx->hCancel[2] = 0u;
            return RTOW_ERROR_LAUNCH_FAILURE;
        }
        if (cancel && *cancel && !cancelled) {
            *ctx->hCancel = 1u;
            cancelled = true;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (cancel && *cancel) cancelled = true;
    if (cancelled) {
        ctx->hCancel[1] = 0u;          // the cancelled batch's outputs are discarded; its overflow must not be blamed on the next batch
        ctx->hCancel[2] = 0u;
        return RTOW_ERROR_CANCELLED;
    }
    return takeOverflow(ctx);
}

// device-visible address of caller memory [p, p + bytes) if it lies inside a range given to rtowRegisterHostBuffer, else null
uint8_t* mappedHost(RtowContext ctx, const void* p, size_t bytes)
{
    const uint8_t* q = (const uint8_t*)p;
    for (const RtowContext_t::HostRange& r : ctx->hostRanges)
        if (q >= r.base && q + bytes <= r.base + r.size) return r.device + (q - r.base);
    return nullptr;
}

int ensureStaging(RtowContext ctx, size_t pixels, size_t diagBytes)
{
    if (pixels > ctx->stagingPixels) {
        if (ctx->dColor) { (void)hipFree(ctx->dColor); (void)hipFree(ctx->dNormal); (void)hipFree(ctx->dAlbedo); (void)hipFree(ctx->dScw); }
        ctx->dColor = ctx->dNormal = ctx->dAlbedo = ctx->dScw = nullptr;
        ctx->stagingPixels = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dColor, pixels * 16), RTOW_ERROR_MEMORY_ALLOCATION);
        HIP_TRY(ctx, hipMalloc(&ctx->dNormal, pixels * 12), RTOW_ERROR_MEMORY_ALLOCATION);
        HIP_TRY(ctx, hipMalloc(&ctx->dAlbedo, pixels * 12), RTOW_ERROR_MEMORY_ALLOCATION);
        HIP_TRY(ctx, hipMalloc(&ctx->dScw, pixels * 4), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->stagingPixels = pixels;
    }
    if (diagBytes > ctx->stagingDiagBytes) {
        if (ctx->dDiag) (void)hipFree(ctx->dDiag);
        ctx->dDiag = nullptr;
        ctx->stagingDiagBytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dDiag, diagBytes), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->stagingDiagBytes = diagBytes;
    }
    return RTOW_SUCCESS;
}

// ---- RCCL, loaded on first use: hosts that drive one GPU never map it, and a process that already holds a copy (PyTorch ships its own
// librccl.so.1) shares that copy.  Only the point-to-point calls the row gather needs; types restated from <rccl/rccl.h> (ROCm 7.2:
// NCCL_UNIQUE_ID_BYTES 128, ncclFloat32 = 7, ncclSuccess = 0) so that the library has no link-time dependency on RCCL. ----
struct RcclUniqueId { char internal[128]; };
static_assert(sizeof(RcclUniqueId) == sizeof(RtowCommId), "RtowCommId carries an ncclUniqueId");
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return handle && GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString; }
};
constexpr int kRcclFloat32 = 7;

std::mutex gRcclMu;
std::string gRcclPath;        // rtowCommSetLibraryPath: the file to load instead of the default search
std::string gRcclLoadError;   // why the last load attempt failed (dlerror() is per thread and may be null by the time it is logged)
bool gRcclLoaded = false;

RcclApi* rccl()
{
    static RcclApi api;
    std::lock_guard<std::mutex> lock(gRcclMu);
    if (api.ok()) return &api;
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    if (!gRcclPath.empty()) {
        h = dlopen(gRcclPath.c_str(), RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;     // a copy this process already holds
        if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!h) {
        const char* e = dlerror();
        gRcclLoadError = e ? e : "dlopen failed";
        return nullptr;
    }
    api.handle = h;
    api.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    api.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    api.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!api.ok()) { gRcclLoadError = "the library does not export the nccl* entry points the row gather needs"; return nullptr; }
    gRcclLoaded = true;
    return &api;
}

#define RCCL_TRY(ctx, api, expr)                                                                       \
    do {                                                                                               \
        const int _r = (expr);                                                                         \
        if (_r != 0) {                                                                                 \
            logf(ctx, 2, "rccl", "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
            return RTOW_ERROR_LAUNCH_FAILURE;                                                          \
        }                                                                                              \
    } while (0)

// rows of the frame owned by `rank` under the reference's interlacing (row % divider == rank, JOBS/SampleBatchJob.cs:69-70)
unsigned rowsOwnedBy(int rank, int divider, int height) { return rank >= height ? 0u : (unsigned)((height - rank + divider - 1) / divider); }

// `count` successive batches (batch 0 reads `in`, every later one what its predecessor wrote to `out`), enqueued on `stream`: as ONE launch per
// group of up to kMaxChain batches when they differ in nothing but Seed, else one after the other (what the chain is defined to equal).
// The caller holds ctx->mu and has validated params / buffers.
int enqueueChain(RtowContext ctx, int count, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* out, void* const* diagnostics,
                 hipStream_t s, const volatile uint8_t* cancel)
{
    // One launch needs batches that differ in nothing but Seed (the reference's successive batches of a frame: UNITY/Raytracer.cs:656-661),
    // the reference RNG policy (per-sample units fold through records) and a frame of fewer than 2^27 padded pixels.
    bool fusable = params[0].rngPolicy == RTOW_RNG_REFERENCE && ctx->chainFusion;
    for (int b = 1; b < count && fusable; b++) {
        RtowSampleParams q = params[b];
        q.seed = params[0].seed;
        fusable = memcmp(&q, &params[0], sizeof(q)) == 0;
    }
    const uint64_t paddedPixels = ((uint64_t)ownedRows(&params[0]) * (uint64_t)(int)params[0].size.x + 63u) & ~63ull;
    if (paddedPixels >= (1ull << 27)) fusable = false;
    // (the tie fix-up list names FRAME pixels beside the batch number, batch << 27 | pixel: a sliced launch of a frame of 2^27 pixels or more runs batch by batch)
    if ((uint64_t)(int)params[0].size.x * (uint64_t)(int)params[0].size.y >= (1ull << 27)) fusable = false;
    // one launch is one kernel variant, and the variant follows the record format (launchByDiag: 16-byte FULL_DIAGNOSTICS records need the
    // counters compiled in): a chain in which only SOME batches carry a diagnostics buffer runs batch by batch, each with its own variant
    if (diagnostics) {
        int withDiag = 0;
        for (int b = 0; b < count; b++) withDiag += diagnostics[b] != nullptr ? 1 : 0;
        if (withDiag != 0 && withDiag != count) fusable = false;
    }
    int rc = RTOW_SUCCESS;
    for (int first = 0; first < count && rc == RTOW_SUCCESS;) {
        const int n = fusable ? std::min(count - first, (int)kMaxChain) : 1;
        const RtowAccumBuffers* src = first == 0 ? in : out;
        if (n == 1) {
            rc = launchSample(ctx, &params[first], src, out, diagnostics ? diagnostics[first] : nullptr, s, cancel != nullptr);
        } else {
            uint32_t seeds[kMaxChain];
            for (int b = 0; b < n; b++) seeds[b] = params[first + b].seed;
            const ChainSpec chain{n, seeds, diagnostics ? diagnostics + first : nullptr, nullptr};
            rc = launchSample(ctx, &params[first], src, out, nullptr, s, cancel != nullptr, &chain);
        }
        if (rc == RTOW_SUCCESS && cancel) rc = waitWithCancel(ctx, cancel);
        first += n;
    }
    return rc;
}

} // namespace

extern "C" {

RTOW_API int rtowGetApiVersion(void) { return RTOW_API_VERSION; }

RTOW_API const char* rtowErrorString(int result)
{
    switch (result) {
        case RTOW_SUCCESS: return "success";
        case RTOW_ERROR_INVALID_VALUE: return "invalid value";
        case RTOW_ERROR_MEMORY_ALLOCATION: return "memory allocation failed";
        case RTOW_ERROR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
        case RTOW_ERROR_NO_SCENE: return "no scene uploaded";
        case RTOW_ERROR_UNSUPPORTED: return "feature not built yet";
        case RTOW_ERROR_LAUNCH_FAILURE: return "kernel launch or stream failure";
        case RTOW_ERROR_CANCELLED: return "cancelled";
        case RTOW_ERROR_CAPACITY: return "capacity exceeded";
        case RTOW_ERROR_INTERNAL: return "internal error";
    }
    return "unknown error";
}

RTOW_API int rtowCreateContext(const RtowContextOptions* options, RtowContext* outContext)
{
    if (!outContext) return RTOW_ERROR_INVALID_VALUE;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RTOW_ERROR_NO_DEVICE;
    const int ordinal = options ? options->deviceOrdinal : 0;
    if (ordinal < 0 || ordinal >= count) return RTOW_ERROR_INVALID_VALUE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return RTOW_ERROR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RTOW_ERROR_NO_DEVICE; // the code object is gfx950 only
    if (hipSetDevice(ordinal) != hipSuccess) return RTOW_ERROR_NO_DEVICE;

    RtowContext ctx = new (std::nothrow) RtowContext_t();
    if (!ctx) return RTOW_ERROR_MEMORY_ALLOCATION;
    ctx->device = ordinal;
    ctx->cuCount = prop.multiProcessorCount;
    if (options) {
        ctx->logCb = options->logCallback; ctx->logData = options->logCallbackData; ctx->logLevel = options->logCallbackLevel;
        ctx->flags = options->flags;
        if ((ctx->flags & RTOW_CONTEXT_EXACT_TIES_ALWAYS) && (ctx->flags & RTOW_CONTEXT_EXACT_TIES_NEVER)) { delete ctx; return RTOW_ERROR_INVALID_VALUE; }
        if (options->ldsSceneBudgetBytes > 0) ctx->ldsSceneBudget = (uint32_t)options->ldsSceneBudgetBytes;
        if (options->hitListCapacity < 0) { delete ctx; return RTOW_ERROR_INVALID_VALUE; }
        if (options->sliceBlockThreads != 0 && options->sliceBlockThreads != 1024) { delete ctx; return RTOW_ERROR_INVALID_VALUE; }     // reserved (round 3: 256 / 512)
        ctx->hitListCapacity = (uint32_t)options->hitListCapacity;
        bool anyTune = false;
        for (int i = 0; i < 9; i++) anyTune = anyTune || (i != 7 && options->schedulerTune[i] != 0);
        if (options->schedulerTune[7] > 0) {                                                  // its own knob: says nothing about the thresholds
            // packed: side (low nibble: 1 none, 3 a tile's tickets most expensive first, 2 / 4 / 8 super-tiles) + 16 x mode (0 .. 3) + 64 x chunk order by total + 256 x queue slots per pull
            // + 4096 x lanes that wait for company at a pixel boundary.
            // A low nibble of 0 (only the upper fields given) keeps the default side; any other value would silently switch the ticket map off
            int v = options->schedulerTune[7];
            if ((v & 15) == 0) v |= RTOW_DEFAULT_REGROUP_SIDE & 15;
            const int side = v & 15;
            if (!(side == 1 || side == 2 || side == 3 || side == 4 || side == 8) || v >= (1 << 16)) {
                logf(ctx, 2, "rtow", "schedulerTune[7] = %d: the low four bits must be 1, 2, 3, 4 or 8 (or 0 for the default)", options->schedulerTune[7]);
                delete ctx;
                return RTOW_ERROR_INVALID_VALUE;
            }
            ctx->regroupSide = v;
        }
        if (anyTune) {
            // stage thresholds below 1 mean "any lane" (1); a zero hand-over count or walk slice means "the built-in value" (3; per scene at upload), as in API v6
            for (int i = 0; i < 9; i++) ctx->tune[i] = options->schedulerTune[i] < 1 ? 1 : options->schedulerTune[i];
            if (options->schedulerTune[6] < 1) ctx->tune[6] = 3;
            ctx->userSliceDefault = options->schedulerTune[8] < 1;
        }
        ctx->userTune = anyTune;
    }
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->evStart) == hipSuccess && hipEventCreate(&ctx->evStop) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->evBatchDone, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->evGatherDone, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->evMetricsDone, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&ctx->dWorkCounter, sizeof(unsigned int)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->dPartials, sizeof(MetricsPartial) * kMetricsBlocks) == hipSuccess;
    // the finalize pass may be given any stream later: the table is complete before the context exists for the caller
    ok = ok && hipMalloc(&ctx->dByteThresholds, kByteThresholdTableBytes) == hipSuccess;
    ok = ok && launchBuildByteThresholds(ctx->dByteThresholds, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    void* pinned = nullptr;
    ok = ok && hipHostMalloc(&pinned, 256, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    void* pinnedDevice = nullptr;
    ok = ok && hipHostGetDevicePointer(&pinnedDevice, pinned, 0) == hipSuccess;
    if (!ok) { if (pinned) (void)hipHostFree(pinned); rtowDestroyContext(ctx); return RTOW_ERROR_MEMORY_ALLOCATION; }
    ctx->hCancel = (volatile uint32_t*)pinned;
    ctx->hMetricsRecord = (volatile RtowMetrics*)((uint8_t*)pinned + 64);         // where the blocking metrics reduction has its record written by the device
    ctx->dMetricsRecord = (RtowMetrics*)((uint8_t*)pinnedDevice + 64);
    ctx->hCancel[0] = 0u;
    ctx->hCancel[1] = 0u;
    ctx->hCancel[2] = 0u;
    ctx->hCancel[3] = 0u;          // [3]: a watched launch of an all-triangle scene listed thousands of tied pixels (read at the next launch: triWatchOff)
    logf(ctx, 4, "rtow", "context on device %d (%s, %d CUs)", ordinal, prop.gcnArchName, ctx->cuCount);
    // chained launches hand accumulators over inside an XCD with plain stores + sc1 loads: measured on THIS device before it is relied on
    if (ctx->flags & RTOW_CONTEXT_NO_CHAIN_FUSION) {
        ctx->chainFusion = false;
    } else {
        // stale > 0: the hand-over is unsafe on this device.  timeouts > 0 with nothing stale: inconclusive - the litmus needs its 2 x CU workgroups resident at the same time,
        // which a device shared with other work may not grant within its bounded waits - so it runs once more before chains are given up (RtowSceneInfo does not say so; the log does)
        unsigned pairs = 0, stale = 0, timeouts = 0;
        hipError_t le = hipSuccess;
        for (int attempt = 0; attempt < 2; attempt++) {
            le = runXcdCoherenceLitmus(ctx->cuCount, ctx->stream, &pairs, &stale, &timeouts);
            if (le != hipSuccess || stale != 0 || (pairs > 0 && timeouts == 0)) break;
            logf(ctx, 3, "rtow", "same-XCD hand-over litmus inconclusive (%u pairs, %u timeouts): %s", pairs, timeouts, attempt == 0 ? "once more" : "giving up");
        }
        ctx->chainFusion = le == hipSuccess && pairs > 0 && stale == 0 && timeouts == 0;
        if (le != hipSuccess) (void)hipGetLastError();
        logf(ctx, ctx->chainFusion ? 4 : 3, "rtow", "same-XCD hand-over litmus: %u pairs, %u stale dwords, %u timeouts%s", pairs, stale, timeouts,
             ctx->chainFusion ? "" : stale ? " - unsafe here: chained batches will run one launch per batch" : " - inconclusive: chained batches will run one launch per batch");
    }
    *outContext = ctx;
    return RTOW_SUCCESS;
}

RTOW_API int rtowDestroyContext(RtowContext ctx)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->dScene) (void)hipFree(ctx->dScene);
    if (ctx->dWorkCounter) (void)hipFree(ctx->dWorkCounter);
    if (ctx->dChunkCost) { (void)hipFree(ctx->dChunkCost); (void)hipFree(ctx->dChunkOrder); (void)hipFree(ctx->dPixelCost); (void)hipFree(ctx->dTicketMap); }
    if (ctx->dChunkDone) (void)hipFree(ctx->dChunkDone);
    if (ctx->dXcdState) (void)hipFree(ctx->dXcdState);
    if (ctx->dChainBatches) (void)hipFree(ctx->dChainBatches);
    if (ctx->dPixCand) (void)hipFree(ctx->dPixCand);
    if (ctx->dCubemap) (void)hipFree(ctx->dCubemap);
    if (ctx->dBlueNoise) (void)hipFree(ctx->dBlueNoise);
    if (ctx->dStbNoise) (void)hipFree(ctx->dStbNoise);
    if (ctx->dTexBlob) (void)hipFree(ctx->dTexBlob);
    if (ctx->dRefTree) (void)hipFree(ctx->dRefTree);
    if (ctx->dHitSpill) (void)hipFree(ctx->dHitSpill);
    if (ctx->dUnitRecords) (void)hipFree(ctx->dUnitRecords);
    if (ctx->dPartials) (void)hipFree(ctx->dPartials);
    if (ctx->dByteThresholds) (void)hipFree(ctx->dByteThresholds);
    dropThresholdTuning(ctx);
    if (ctx->dProbeSink) (void)hipFree(ctx->dProbeSink);
    if (ctx->dTieRedo) (void)hipFree(ctx->dTieRedo);
    if (ctx->dTieBits) (void)hipFree(ctx->dTieBits);
    if (ctx->dTieInputs) (void)hipFree(ctx->dTieInputs);
    if (ctx->dHistSpill) (void)hipFree(ctx->dHistSpill);
    if (ctx->dRedoSpill) (void)hipFree(ctx->dRedoSpill);
    if (ctx->hCancel) (void)hipHostFree((void*)ctx->hCancel);
    if (ctx->dColor) { (void)hipFree(ctx->dColor); (void)hipFree(ctx->dNormal); (void)hipFree(ctx->dAlbedo); (void)hipFree(ctx->dScw); }
    if (ctx->dDiag) (void)hipFree(ctx->dDiag);
    if (ctx->comm) { if (RcclApi* api = rccl()) (void)api->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    if (ctx->dGatherSend) (void)hipFree(ctx->dGatherSend);
    if (ctx->dGatherRecv) (void)hipFree(ctx->dGatherRecv);
    for (const RtowContext_t::HostRange& r : ctx->hostRanges) (void)hipHostUnregister(r.base);
    ctx->hostRanges.clear();
    if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
    if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
    if (ctx->evBatchDone) (void)hipEventDestroy(ctx->evBatchDone);
    if (ctx->evGatherDone) (void)hipEventDestroy(ctx->evGatherDone);
    if (ctx->evMetricsDone) (void)hipEventDestroy(ctx->evMetricsDone);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RTOW_SUCCESS;
}

RTOW_API int rtowUploadScene(RtowContext ctx, const RtowSceneDesc* scene)
{
    if (!ctx || !scene) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    CompiledScene compiled;
    std::string err;
    // the library's own tree is always built to the LDS stack's depth; scene->maxBvhDepth (the host's tree) only decides the order of
    // hits at identical distances (rtow_reforder.h)
    const int rc = compileScene(scene, RTOW_STACK_CAPACITY, &compiled, &err);
    if (rc != RTOW_SUCCESS) {
        logf(ctx, 2, "scene", "%s", err.c_str());
        return rc;
    }
    HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);   // no batch (on whatever stream it was given) may still be reading the old scene
    if (compiled.blob.size() > ctx->dSceneCapacity) {
        if (ctx->dScene) (void)hipFree(ctx->dScene);
        ctx->dScene = nullptr;
        ctx->dSceneCapacity = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dScene, compiled.blob.size()), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->dSceneCapacity = compiled.blob.size();
    }
    HIP_TRY(ctx, hipMemcpy(ctx->dScene, compiled.blob.data(), compiled.blob.size(), hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    if (!compiled.texBlob.empty()) {
        if (compiled.texBlob.size() > ctx->texBlobCapacity) {
            if (ctx->dTexBlob) (void)hipFree(ctx->dTexBlob);
            ctx->dTexBlob = nullptr;
            ctx->texBlobCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dTexBlob, compiled.texBlob.size()), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->texBlobCapacity = compiled.texBlob.size();
        }
        HIP_TRY(ctx, hipMemcpy(ctx->dTexBlob, compiled.texBlob.data(), compiled.texBlob.size(), hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
        compiled.texBlob.clear();
        compiled.texBlob.shrink_to_fit();                                     // the host copy is not needed again
    }
    if (ctx->flags & RTOW_CONTEXT_REFERENCE_DIAGNOSTICS) {
        // the counters' walk keeps one stack entry per level of the reference tree (+1): 64 entries of scratch
        if (compiled.refTreeDepth > 62) { logf(ctx, 2, "scene", "RTOW_CONTEXT_REFERENCE_DIAGNOSTICS supports MaxBvhDepth <= 62"); return RTOW_ERROR_CAPACITY; }
        if (compiled.refTree.size() > ctx->refTreeCapacity) {
            if (ctx->dRefTree) (void)hipFree(ctx->dRefTree);
            ctx->dRefTree = nullptr;
            ctx->refTreeCapacity = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dRefTree, compiled.refTree.size()), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->refTreeCapacity = compiled.refTree.size();
        }
        HIP_TRY(ctx, hipMemcpy(ctx->dRefTree, compiled.refTree.data(), compiled.refTree.size(), hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    }
    compiled.refTree.clear();
    compiled.refTree.shrink_to_fit();
    HIP_TRY(ctx, launchPrepareMaterials(ctx->dScene, compiled.layout, ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, launchPrepareEntities(ctx->dScene, compiled.layout, ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    // rtowProbeNearestHit walks the HOST image of the scene: bring back what the device derived for the entities (inverse rotations / translations of the GpuPrim records)
    if (compiled.layout.sceneKind >= SCENE_KIND_GENERAL && compiled.entityCount > 0)
        HIP_TRY(ctx, hipMemcpy(compiled.blob.data() + compiled.layout.primOffset, ctx->dScene + compiled.layout.primOffset, (size_t)compiled.entityCount * sizeof(GpuPrim), hipMemcpyDeviceToHost),
                RTOW_ERROR_LAUNCH_FAILURE);
    if (ctx->flags & (RTOW_CONTEXT_EXACT_TIES_ALWAYS | RTOW_CONTEXT_EXACT_TIES_NEVER)) {   // exact-tie kernels for every scene without volumes (slower), or never
        const bool volumes = compiled.layout.sceneKind == SCENE_KIND_VOLUMES || compiled.layout.sceneKind == SCENE_KIND_VOLUMES_TEXTURED;
        if (!volumes) compiled.layout.exactTies = (ctx->flags & RTOW_CONTEXT_EXACT_TIES_ALWAYS) ? 1u : 0u;
    }
    {
        // Rays whose hit list outgrows a lane's own 24 entries (every hit is kept in volume scenes and by the exact-tie procedure) continue in HBM
        const int rc = sizeHitSpill(ctx, compiled.layout.sceneKind, compiled.layout.exactTies != 0, compiled.entityCount);
        if (rc != RTOW_SUCCESS) return rc;
    }
    // every scene kind has kernels with 32-bit codes (volume kinds included: a triangle-mesh scene with one fog volume among the meshes)
    const bool wide = compiled.entityCount > 65535 || compiled.layout.nodeCount > 65535u || (ctx->flags & RTOW_CONTEXT_FORCE_WIDE_CODES) != 0;
    ctx->wideCodes = wide;
    if (!ctx->userTune) {
        // Box-walk slice (node visits per trip).  16 for trees whose nodes come from LDS or L2 (with the hand-over at 3 candidates: cover 12 / 16 / 20
        // visits 9.34 / 9.48 / 9.23 Gsamples/s; 10 000 spheres, tree partly in LDS, 16 / 20 / 24: 8.00 / 7.81 / 7.45).  A tree of hundreds of
        // thousands of nodes is read from HBM at several times the latency per visit, and a ray visits twice as many nodes: longer slices amortise the
        // trip around them (250 882-triangle mesh, 24 / 32 / 40 visits: 2.04 / 1.97 / 1.86 Gsamples/s; gpurun_out/r03bc.  With walks that ran until the
        // candidate list was full the optima were 16 / 20 / 32: r03h, r03ap).
        static const int kDefault[9] = {RTOW_DEFAULT_TUNE}, kGeneral[9] = {RTOW_GENERAL_TUNE};
        const int* base = compiled.layout.sceneKind <= SCENE_KIND_SPHERES_MOTION ? kDefault : kGeneral;      // measured per family: see RTOW_DEFAULT_TUNE
        for (int k = 0; k < 9; k++) ctx->tune[k] = base[k];
        // (round 6: on the rank-rule triangle kernel that traces meshes now, 12 / 16 / 20 / 24 / 32 visits run 2 374 / 2 361 / 2 326 / 2 343 / 2 231 Msamples/s: the 24 of
        // round 3's exact-tie kernel is no better than the 16 everything else uses, profiles/r06n_mesh_scheduler_sweep.json)
    } else if (ctx->userSliceDefault) {
        ctx->tune[8] = 16;
    }
    std::lock_guard<std::mutex> sceneLock(ctx->sceneMu);          // rtowProbeNearestHit reads the host image under this lock only
    ctx->scene = std::move(compiled);
    // LDS of a launch: a traversal-stack row per inner level of THIS tree, then the scene image - whole, or the top of the node array (planLds, rtow_kernels.h)
    ctx->ldsPlan = planLds(wide, ctx->scene.layout, 0u, ctx->ldsSceneBudget);
    ctx->ldsSceneBytes = ctx->ldsPlan.sceneBytes;
    ctx->ldsNodeCount = ctx->ldsPlan.nodeCount;
    ctx->haveScene = true;
    ctx->triWatchOff = false;
    ctx->hCancel[3] = 0u;
    ctx->sceneSerial++;
    ctx->orderValid = false;
    dropThresholdTuning(ctx);                                   // (the device is idle: rtowUploadScene synchronised it above)
    ctx->sppSinceUpload = 0;
    ctx->sceneSignatureNow = sceneSignature(ctx->scene, wide);
    if (!ctx->userTune && !(ctx->flags & RTOW_CONTEXT_NO_THRESHOLD_TUNING))
        for (const RtowContext_t::TuneCacheEntry& e : ctx->tuneCache)
            if (e.signature == ctx->sceneSignatureNow) {
                // a scene like one this context has measured before (same kernel kind, entity / node / material counts): its thresholds, no new probes
                static const int kSets[3][9] = {{RTOW_DEFAULT_TUNE}, {RTOW_GENERAL_TUNE}, {RTOW_GENERAL_TUNE_2}};
                for (int k = 0; k < 8; k++) ctx->tune[k] = kSets[e.winner % 3][k];
                if (e.winner >= 3) ctx->tune[5] = 32;
                ctx->tunedCandidate = e.winner;
                ctx->tunedScene = ctx->sceneSerial;
            }
    logf(ctx, 4, "scene", "%d entities, %u BVH nodes, depth %u, %u bytes (%u in LDS)%s", ctx->scene.entityCount, ctx->scene.layout.nodeCount,
         ctx->scene.layout.bvhDepth, ctx->scene.layout.totalBytes, ctx->ldsSceneBytes,
         ctx->scene.layout.exactTies ? ", exact-tie kernels" : "");
    return RTOW_SUCCESS;
}

RTOW_API int rtowUploadSkyCubemap(RtowContext ctx, const RtowCubemapDesc* cubemap)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    if (!cubemap || !cubemap->faces) {                       // drop it: Cubemap.Sample returns default when the data pointer is null
        HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);
        if (ctx->dCubemap) (void)hipFree(ctx->dCubemap);
        ctx->dCubemap = nullptr;
        ctx->cubemapCapacity = 0;
        ctx->cubemap = RtowCubemapDesc{};
        return RTOW_SUCCESS;
    }
    if (cubemap->faceWidth <= 0 || cubemap->faceHeight <= 0 || cubemap->faceWidth > 32768 || cubemap->faceHeight > 32768) return RTOW_ERROR_INVALID_VALUE;
    if (cubemap->channelType != RTOW_CUBEMAP_UNSIGNED_BYTE && cubemap->channelType != RTOW_CUBEMAP_SIGNED_HALF) return RTOW_ERROR_INVALID_VALUE;
    const int minStride = cubemap->channelType == RTOW_CUBEMAP_SIGNED_HALF ? 6 : 3;            // r, g, b are read
    if (cubemap->pixelStride < minStride || cubemap->pixelStride > 64) return RTOW_ERROR_INVALID_VALUE;
    if (cubemap->channelType == RTOW_CUBEMAP_SIGNED_HALF && (cubemap->pixelStride & 1)) return RTOW_ERROR_INVALID_VALUE;
    const size_t bytes = (size_t)6 * (size_t)cubemap->faceWidth * (size_t)cubemap->faceHeight * (size_t)cubemap->pixelStride;
    if (bytes > 0x7fffffffull) return RTOW_ERROR_CAPACITY;                                      // Cubemap.Sample's strides are int32 here as in the reference (RT/Texture.cs:146-148)
    HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);                            // no batch may still be reading the old faces
    if (bytes > ctx->cubemapCapacity) {
        if (ctx->dCubemap) (void)hipFree(ctx->dCubemap);
        ctx->dCubemap = nullptr;
        ctx->cubemapCapacity = 0;
        ctx->cubemap = RtowCubemapDesc{};
        HIP_TRY(ctx, hipMalloc(&ctx->dCubemap, bytes), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->cubemapCapacity = bytes;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->dCubemap, cubemap->faces, bytes, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    ctx->cubemap = *cubemap;
    ctx->cubemap.faces = nullptr;
    logf(ctx, 3, "sky", "cubemap %dx%d, %s, stride %d (%zu bytes)", cubemap->faceWidth, cubemap->faceHeight,
         cubemap->channelType == RTOW_CUBEMAP_SIGNED_HALF ? "half" : "byte", cubemap->pixelStride, bytes);
    return RTOW_SUCCESS;
}

RTOW_API int rtowUploadBlueNoise(RtowContext ctx, const RtowBlueNoiseDesc* noise)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);
    if (ctx->dBlueNoise) (void)hipFree(ctx->dBlueNoise);
    ctx->dBlueNoise = nullptr;
    ctx->blueRowStride = ctx->blueTextureCount = 0;
    if (!noise || !noise->texels) return RTOW_SUCCESS;
    if (noise->rowStride == 0 || noise->rowStride > 16384 || noise->textureCount == 0 || noise->textureCount > 4096) return RTOW_ERROR_INVALID_VALUE;
    const size_t bytes = (size_t)noise->rowStride * noise->rowStride * noise->textureCount * 8u;      // half4 texels
    HIP_TRY(ctx, hipMalloc(&ctx->dBlueNoise, bytes), RTOW_ERROR_MEMORY_ALLOCATION);
    HIP_TRY(ctx, hipMemcpy(ctx->dBlueNoise, noise->texels, bytes, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    ctx->blueRowStride = noise->rowStride;
    ctx->blueTextureCount = noise->textureCount;
    return RTOW_SUCCESS;
}

RTOW_API int rtowUploadStbNoise(RtowContext ctx, const RtowStbNoiseDesc* noise)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipDeviceSynchronize(), RTOW_ERROR_LAUNCH_FAILURE);
    if (ctx->dStbNoise) (void)hipFree(ctx->dStbNoise);
    ctx->dStbNoise = nullptr;
    ctx->stbRowStride = ctx->stbTextureCount = 0;
    if (!noise) return RTOW_SUCCESS;
    if (!noise->scalar || !noise->vector2 || !noise->cosineUnitVector3 || !noise->unitVector2 || !noise->unitVector3) return RTOW_ERROR_INVALID_VALUE;
    if (noise->rowStride == 0 || noise->rowStride > 16384 || noise->textureCount == 0 || noise->textureCount > 4096) return RTOW_ERROR_INVALID_VALUE;
    const size_t all = (size_t)noise->rowStride * noise->rowStride * noise->textureCount;
    HIP_TRY(ctx, hipMalloc(&ctx->dStbNoise, all * 14u), RTOW_ERROR_MEMORY_ALLOCATION);                // 1 + 3 + 4 + 3 + 3 bytes per texel
    HIP_TRY(ctx, hipMemcpy(ctx->dStbNoise, noise->scalar, all, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipMemcpy(ctx->dStbNoise + all, noise->vector2, all * 3, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipMemcpy(ctx->dStbNoise + all * 4, noise->cosineUnitVector3, all * 4, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipMemcpy(ctx->dStbNoise + all * 8, noise->unitVector2, all * 3, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipMemcpy(ctx->dStbNoise + all * 11, noise->unitVector3, all * 3, hipMemcpyHostToDevice), RTOW_ERROR_LAUNCH_FAILURE);
    ctx->stbRowStride = noise->rowStride;
    ctx->stbTextureCount = noise->textureCount;
    return RTOW_SUCCESS;
}

RTOW_API int rtowGetSceneInfo(RtowContext ctx, RtowSceneInfo* info)
{
    if (!ctx || !info) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    if (ctx->tunePending && hipSetDevice(ctx->device) == hipSuccess) (void)finishThresholdTuning(ctx, /*wait*/ false);     // a measurement whose probes are done by now
    info->entityCount = ctx->scene.entityCount;
    info->materialCount = ctx->scene.materialCount;
    info->bvhNodeCount = (int32_t)ctx->scene.layout.nodeCount;
    info->bvhDepth = (int32_t)ctx->scene.layout.bvhDepth;
    info->ldsBytesScene = (int32_t)ctx->ldsSceneBytes;
    info->sceneInLds = ctx->ldsSceneBytes == ctx->scene.layout.totalBytes ? 1 : 0;
    info->sceneBytesDevice = ctx->scene.layout.totalBytes;
    info->hitSpillBytes = (uint64_t)ctx->hitSpillEntries * (uint64_t)ctx->cuCount * (uint64_t)kBlockThreads * sizeof(uint4);
    const bool keepsLists = ctx->scene.layout.exactTies || ctx->scene.layout.sceneKind == SCENE_KIND_VOLUMES || ctx->scene.layout.sceneKind == SCENE_KIND_VOLUMES_TEXTURED;
    // sphere scenes under the rank rule keep no lists in the fast kernel, but their tie fix-up launch (the exact-tie kernel over the marked pixels) does: its capacity is what a
    // RTOW_ERROR_CAPACITY of such a scene has just doubled (growHitList), and what a device-resident caller compares across the error
    const bool tieWatch = ctx->scene.layout.sceneKind <= SCENE_KIND_SPHERES_MOTION && !ctx->scene.layout.exactTies && ctx->scene.entityCount > 16 && !(ctx->flags & RTOW_CONTEXT_EXACT_TIES_NEVER);
    info->hitListCapacity = keepsLists ? (int32_t)(ctx->hitSpillEntries + (uint32_t)kLocalHitEntries)
                                       : tieWatch ? (int32_t)std::min<uint64_t>((uint64_t)ctx->scene.entityCount, listCapacity(ctx, false)) : 0;
    info->wideCodes = ctx->wideCodes ? 1 : 0;
    info->thresholdSet = ctx->tunedScene == ctx->sceneSerial ? ctx->tunedCandidate : -1;
    for (int k = 0; k < 9; k++) info->schedulerTune[k] = ctx->tune[k];
    info->schedulerTune[7] = ctx->regroupSide;
    return RTOW_SUCCESS;
}

RTOW_API int rtowSampleBatchDevice(RtowContext ctx, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                   void* diagnostics, void* stream, const volatile uint8_t* cancel)
{
    if (!ctx || !in || !out) return RTOW_ERROR_INVALID_VALUE;
    const int v = validateParams(params);
    if (v != RTOW_SUCCESS) return v;
    if (!in->color || !in->normal || !in->albedo || !in->sampleCountWeight || !out->color || !out->normal || !out->albedo || !out->sampleCountWeight)
        return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE); // called from a different worker thread each time
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    *ctx->hCancel = 0u;
    const int rc = launchSample(ctx, params, in, out, diagnostics, s, cancel != nullptr);
    if (rc != RTOW_SUCCESS) return rc;
    if (cancel) return waitWithCancel(ctx, cancel);
    return RTOW_SUCCESS;
}

RTOW_API int rtowSampleBatchChainDevice(RtowContext ctx, int32_t count, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                        void* const* diagnostics, void* stream, const volatile uint8_t* cancel)
{
    if (!ctx || !in || !out || !params || count < 1) return RTOW_ERROR_INVALID_VALUE;
    for (int b = 0; b < count; b++) {
        const int v = validateParams(&params[b]);
        if (v != RTOW_SUCCESS) return v;
    }
    if (!in->color || !in->normal || !in->albedo || !in->sampleCountWeight || !out->color || !out->normal || !out->albedo || !out->sampleCountWeight)
        return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    *ctx->hCancel = 0u;
    return enqueueChain(ctx, count, params, in, out, diagnostics, s, cancel);
}

RTOW_API int rtowSampleBatchGroupDevice(RtowContext ctx, int32_t count, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* outs,
                                        void* const* diagnostics, void* stream, const volatile uint8_t* cancel)
{
    if (!ctx || !in || !outs || !params || count < 1) return RTOW_ERROR_INVALID_VALUE;
    for (int b = 0; b < count; b++) {
        const int v = validateParams(&params[b]);
        if (v != RTOW_SUCCESS) return v;
        if (!outs[b].color || !outs[b].normal || !outs[b].albedo || !outs[b].sampleCountWeight) return RTOW_ERROR_INVALID_VALUE;
        for (int c = 0; c < b; c++)                                                 // every batch its own outputs (a batch may store over the shared inputs only if it is alone)
            if (outs[b].color == outs[c].color || outs[b].normal == outs[c].normal || outs[b].albedo == outs[c].albedo || outs[b].sampleCountWeight == outs[c].sampleCountWeight) return RTOW_ERROR_INVALID_VALUE;
        if (count > 1 && (outs[b].color == in->color || outs[b].normal == in->normal || outs[b].albedo == in->albedo || outs[b].sampleCountWeight == in->sampleCountWeight)) return RTOW_ERROR_INVALID_VALUE;
    }
    if (!in->color || !in->normal || !in->albedo || !in->sampleCountWeight) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    *ctx->hCancel = 0u;
    // one launch needs batches that differ in nothing but Seed, the reference RNG policy, fewer than 2^27 padded pixels, and diagnostics for all batches or for none
    bool fusable = params[0].rngPolicy == RTOW_RNG_REFERENCE;
    for (int b = 1; b < count && fusable; b++) {
        RtowSampleParams q = params[b];
        q.seed = params[0].seed;
        fusable = memcmp(&q, &params[0], sizeof(q)) == 0;
    }
    const uint64_t paddedPixels = ((uint64_t)ownedRows(&params[0]) * (uint64_t)(int)params[0].size.x + 63u) & ~63ull;
    if (paddedPixels >= (1ull << 27)) fusable = false;
    if ((uint64_t)(int)params[0].size.x * (uint64_t)(int)params[0].size.y >= (1ull << 27)) fusable = false;      // tie fix-up entries: batch << 27 | FRAME pixel (enqueueChain)
    if (diagnostics) {
        int withDiag = 0;
        for (int b = 0; b < count; b++) withDiag += diagnostics[b] != nullptr ? 1 : 0;
        if (withDiag != 0 && withDiag != count) fusable = false;
    }
    int rc = RTOW_SUCCESS;
    for (int first = 0; first < count && rc == RTOW_SUCCESS;) {
        const int n = fusable ? std::min(count - first, (int)kMaxChain) : 1;
        if (n == 1) {
            rc = launchSample(ctx, &params[first], in, &outs[first], diagnostics ? diagnostics[first] : nullptr, s, cancel != nullptr);
        } else {
            uint32_t seeds[kMaxChain];
            for (int b = 0; b < n; b++) seeds[b] = params[first + b].seed;
            const ChainSpec group{n, seeds, diagnostics ? diagnostics + first : nullptr, outs + first};
            rc = launchSample(ctx, &params[first], in, &outs[first], nullptr, s, cancel != nullptr, &group);
        }
        if (rc == RTOW_SUCCESS && cancel) rc = waitWithCancel(ctx, cancel);
        first += n;
    }
    return rc;
}

RTOW_API int rtowSampleBatchChain(RtowContext ctx, int32_t count, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                                  void* const* diagnostics, const volatile uint8_t* cancel)
{
    if (!ctx || !in || !out || !params || count < 1) return RTOW_ERROR_INVALID_VALUE;
    for (int b = 0; b < count; b++) {
        const int v = validateParams(&params[b]);
        if (v != RTOW_SUCCESS) return v;
        // one staging set serves the whole chain: the batches share the frame and the record format
        if ((int)params[b].size.x != (int)params[0].size.x || (int)params[b].size.y != (int)params[0].size.y || params[b].diagnosticsStride != params[0].diagnosticsStride ||
            params[b].sliceOffset != params[0].sliceOffset || params[b].sliceDivider != params[0].sliceDivider)
            return RTOW_ERROR_INVALID_VALUE;
    }
    if (!in->color || !in->normal || !in->albedo || !in->sampleCountWeight || !out->color || !out->normal || !out->albedo || !out->sampleCountWeight)
        return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    const int w = (int)params[0].size.x, h = (int)params[0].size.y;
    const size_t n = (size_t)w * (size_t)h;
    const size_t diagBytes = n * (size_t)params[0].diagnosticsStride;
    int rc = ensureStaging(ctx, n, diagnostics ? diagBytes * (size_t)count : 0);
    if (rc != RTOW_SUCCESS) return rc;
    hipStream_t s = ctx->stream;
    // the chain accumulates in place in the staging buffers (its batches read what their predecessors wrote there); only the final
    // accumulators and each batch's diagnostics travel back
    RtowAccumBuffers dev{ctx->dColor, ctx->dNormal, ctx->dAlbedo, ctx->dScw};
    std::vector<void*> devDiag((size_t)count, nullptr);
    if (diagnostics) for (int b = 0; b < count; b++) devDiag[(size_t)b] = diagnostics[b] ? ctx->dDiag + (size_t)b * diagBytes : nullptr;
    for (;;) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dColor, in->color, n * 16, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dNormal, in->normal, n * 12, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dAlbedo, in->albedo, n * 12, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dScw, in->sampleCountWeight, n * 4, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        *ctx->hCancel = 0u;
        ctx->overflowGrew = false;
        rc = enqueueChain(ctx, count, params, &dev, &dev, diagnostics ? devDiag.data() : nullptr, s, cancel);
        if (rc == RTOW_SUCCESS && !cancel) rc = waitWithCancel(ctx, nullptr);
        // a ray outgrew the hit lists and they have grown since (hitListCapacity 0): the caller's inputs are untouched - nothing has been copied back - so the chain runs again
        if (rc == RTOW_ERROR_CAPACITY && ctx->overflowGrew) continue;
        if (rc != RTOW_SUCCESS) return rc;
        break;
    }
    const int rows = ownedRows(&params[0]);
    if (rows > 0) {
        const size_t D = (size_t)params[0].sliceDivider, O = (size_t)params[0].sliceOffset;
        auto copyRows = [&](void* dst, const void* src, size_t bytesPerPixel) -> hipError_t {
            const size_t rowBytes = (size_t)w * bytesPerPixel;
            return hipMemcpy2DAsync((uint8_t*)dst + O * rowBytes, D * rowBytes, (const uint8_t*)src + O * rowBytes, D * rowBytes, rowBytes, (size_t)rows,
                                    hipMemcpyDeviceToHost, s);
        };
        HIP_TRY(ctx, copyRows(out->color, ctx->dColor, 16), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->normal, ctx->dNormal, 12), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->albedo, ctx->dAlbedo, 12), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->sampleCountWeight, ctx->dScw, 4), RTOW_ERROR_LAUNCH_FAILURE);
        if (diagnostics)
            for (int b = 0; b < count; b++)
                if (diagnostics[b]) HIP_TRY(ctx, copyRows(diagnostics[b], devDiag[(size_t)b], (size_t)params[0].diagnosticsStride), RTOW_ERROR_LAUNCH_FAILURE);
    }
    HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowSampleBatch(RtowContext ctx, const RtowSampleParams* params, const RtowAccumBuffers* in, const RtowAccumBuffers* out,
                             void* diagnostics, const volatile uint8_t* cancel)
{
    if (!ctx || !in || !out) return RTOW_ERROR_INVALID_VALUE;
    const int v = validateParams(params);
    if (v != RTOW_SUCCESS) return v;
    if (!in->color || !in->normal || !in->albedo || !in->sampleCountWeight || !out->color || !out->normal || !out->albedo || !out->sampleCountWeight)
        return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    const int w = (int)params->size.x, h = (int)params->size.y;
    const size_t n = (size_t)w * (size_t)h;
    const size_t diagBytes = diagnostics ? n * (size_t)params->diagnosticsStride : 0;
    int rc = ensureStaging(ctx, n, diagBytes);
    if (rc != RTOW_SUCCESS) return rc;
    hipStream_t s = ctx->stream;

    // outputs: when every output buffer (and the diagnostics) lies in registered host memory the kernel stores straight into it - each
    // pixel's 48 + stride bytes leave over PCIe when that pixel finishes, spread over the whole batch, and there is no copy-back at all.
    // Pixels skipped by the slice test are not touched either way (JOBS/SampleBatchJob.cs:69-70).
    RtowAccumBuffers dev{ctx->dColor, ctx->dNormal, ctx->dAlbedo, ctx->dScw}; // staging is read and (copy-back path) written in place: each lane reads its pixel before writing it
    RtowAccumBuffers direct{(float*)mappedHost(ctx, out->color, n * 16), (float*)mappedHost(ctx, out->normal, n * 12), (float*)mappedHost(ctx, out->albedo, n * 12),
                            (float*)mappedHost(ctx, out->sampleCountWeight, n * 4)};
    uint8_t* directDiag = diagnostics ? mappedHost(ctx, diagnostics, diagBytes) : nullptr;
    const bool zeroCopyOut = direct.color && direct.normal && direct.albedo && direct.sampleCountWeight && (!diagnostics || directDiag);
    // (stores that go straight into the caller's OUTPUT arrays leave its input arrays alone unless they are the same arrays: only then a batch cannot be run twice)
    const bool inputsSurvive = !zeroCopyOut || (in->color != out->color && in->normal != out->normal && in->albedo != out->albedo && in->sampleCountWeight != out->sampleCountWeight);
    for (;;) {
        // inputs: one DMA per buffer into the grow-only staging (pinned when the caller registered its pools, pageable otherwise)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dColor, in->color, n * 16, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dNormal, in->normal, n * 12, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dAlbedo, in->albedo, n * 12, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dScw, in->sampleCountWeight, n * 4, hipMemcpyHostToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
        *ctx->hCancel = 0u;
        ctx->overflowGrew = false;
        rc = launchSample(ctx, params, &dev, zeroCopyOut ? &direct : &dev, diagnostics ? (zeroCopyOut ? (void*)directDiag : (void*)ctx->dDiag) : nullptr, s, cancel != nullptr);
        if (rc != RTOW_SUCCESS) return rc;
        rc = waitWithCancel(ctx, cancel);
        // The reference's hit list grows without bound (UTIL/HybridCollections.cs:65-71).  Here a ray that outgrew the lists made the batch invalid and the lists twice
        // as long (hitListCapacity 0: growHitList): the batch runs again from the caller's inputs, until it fits or the scene's own bound / the memory is reached.
        if (rc == RTOW_ERROR_CAPACITY && ctx->overflowGrew && inputsSurvive) continue;
        if (rc != RTOW_SUCCESS) return rc;
        break;
    }

    // copy back ONLY the rows this slice owns: skipped pixels write nothing (JOBS/SampleBatchJob.cs:69-70)
    const int rows = ownedRows(params);
    if (rows > 0 && !zeroCopyOut) {
        const size_t D = (size_t)params->sliceDivider, O = (size_t)params->sliceOffset;
        auto copyRows = [&](void* dst, const void* src, size_t bytesPerPixel) -> hipError_t {
            const size_t rowBytes = (size_t)w * bytesPerPixel;
            return hipMemcpy2DAsync((uint8_t*)dst + O * rowBytes, D * rowBytes, (const uint8_t*)src + O * rowBytes, D * rowBytes, rowBytes, (size_t)rows,
                                    hipMemcpyDeviceToHost, s);
        };
        HIP_TRY(ctx, copyRows(out->color, ctx->dColor, 16), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->normal, ctx->dNormal, 12), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->albedo, ctx->dAlbedo, 12), RTOW_ERROR_LAUNCH_FAILURE);
        HIP_TRY(ctx, copyRows(out->sampleCountWeight, ctx->dScw, 4), RTOW_ERROR_LAUNCH_FAILURE);
        if (diagnostics) HIP_TRY(ctx, copyRows(diagnostics, ctx->dDiag, (size_t)params->diagnosticsStride), RTOW_ERROR_LAUNCH_FAILURE);
    }
    HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
    if (ctx->haveBatchDone) HIP_TRY(ctx, hipEventSynchronize(ctx->evBatchDone), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowRegisterHostBuffer(RtowContext ctx, void* pointer, size_t sizeInBytes)
{
    if (!ctx || !pointer || sizeInBytes == 0) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    for (const RtowContext_t::HostRange& r : ctx->hostRanges)
        if ((uint8_t*)pointer < r.base + r.size && r.base < (uint8_t*)pointer + sizeInBytes) return RTOW_ERROR_INVALID_VALUE;   // overlaps a live registration
    HIP_TRY(ctx, hipHostRegister(pointer, sizeInBytes, hipHostRegisterMapped | hipHostRegisterPortable), RTOW_ERROR_MEMORY_ALLOCATION);
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, pointer, 0) != hipSuccess || !dev) {
        (void)hipHostUnregister(pointer);
        return RTOW_ERROR_MEMORY_ALLOCATION;
    }
    ctx->hostRanges.push_back(RtowContext_t::HostRange{(uint8_t*)pointer, sizeInBytes, (uint8_t*)dev});
    return RTOW_SUCCESS;
}

RTOW_API int rtowUnregisterHostBuffer(RtowContext ctx, void* pointer)
{
    if (!ctx || !pointer) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    for (size_t i = 0; i < ctx->hostRanges.size(); i++)
        if (ctx->hostRanges[i].base == (uint8_t*)pointer) {
            if (ctx->haveBatchDone) (void)hipEventSynchronize(ctx->evBatchDone);      // no kernel may still be storing into it
            (void)hipStreamSynchronize(ctx->stream);
            HIP_TRY(ctx, hipHostUnregister(pointer), RTOW_ERROR_INVALID_VALUE);
            ctx->hostRanges.erase(ctx->hostRanges.begin() + (long)i);
            return RTOW_SUCCESS;
        }
    return RTOW_ERROR_INVALID_VALUE;
}

RTOW_API int rtowProbeNearestHit(RtowContext ctx, const RtowFloat3* origin, const RtowFloat3* direction, float time, float* distance, int32_t* entityIndex)
{
    if (!ctx || !origin || !direction) return RTOW_ERROR_INVALID_VALUE;
    // the host image of the scene has a lock of its own (held here and while rtowUploadScene swaps the image in): the blocking entry points hold ctx->mu for a whole
    // batch, and a probe from another thread must not wait for them
    std::lock_guard<std::mutex> lock(ctx->sceneMu);
    if (!ctx->haveScene) return RTOW_ERROR_NO_SCENE;
    const float o[3] = {origin->x, origin->y, origin->z}, d[3] = {direction->x, direction->y, direction->z};
    float t = 0.0f;
    int prim = -1;
    (void)probeNearestHitHost(ctx->scene.blob.data(), ctx->scene.layout, ctx->scene.entityOfPrim.empty() ? nullptr : ctx->scene.entityOfPrim.data(), o, d, time, &t, &prim);      // no device work: batches in flight are neither waited for nor disturbed
    if (distance) *distance = t;
    if (entityIndex) *entityIndex = prim;
    return RTOW_SUCCESS;
}

RTOW_API int rtowGetLastSampleKernelMs(RtowContext ctx, float* outMs)
{
    if (!ctx || !outMs) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->haveTiming) return RTOW_ERROR_INVALID_VALUE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipEventSynchronize(ctx->evStop), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipEventElapsedTime(outMs, ctx->evStart, ctx->evStop), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowReduceMetricsDevice(RtowContext ctx, int32_t pixelCount, const void* diagnostics, int32_t diagnosticsStride, const float* color,
                                     const float* sampleCountWeight, void* stream, RtowMetrics* outMetrics)
{
    if (!ctx || !diagnostics || !color || !sampleCountWeight || !outMetrics || pixelCount <= 0) return RTOW_ERROR_INVALID_VALUE;
    if (diagnosticsStride != 4 && diagnosticsStride != 16) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (ctx->haveMetricsDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evMetricsDone, 0), RTOW_ERROR_LAUNCH_FAILURE);      // an asynchronous reduction may still be folding the partials
    HIP_TRY(ctx, launchReduceMetrics(pixelCount, (const uint8_t*)diagnostics, diagnosticsStride, color, sampleCountWeight, ctx->dPartials, s),
            RTOW_ERROR_LAUNCH_FAILURE);
    // the per-block partials are folded on the device and the 40-byte record lands in pinned host memory by the time the stream is idle: no 8 KB copy of the partials, no
    // DMA of its own (round 4 copied and folded them on the host: 44.6 us per call at 1920 x 1080 against 18.6 us of kernels)
    HIP_TRY(ctx, launchFoldMetrics(ctx->dPartials, ctx->dMetricsRecord, s), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
    memcpy(outMetrics, (const void*)ctx->hMetricsRecord, sizeof(RtowMetrics));
    return RTOW_SUCCESS;
}

RTOW_API int rtowCombineDevice(RtowContext ctx, const RtowCombineParams* params, const float* inColor, const float* inNormal, const float* inAlbedo,
                               float* outColor, float* outNormal, float* outAlbedo, void* stream)
{
    if (!ctx || !params || !inColor || !inNormal || !inAlbedo || !outColor || !outNormal || !outAlbedo) return RTOW_ERROR_INVALID_VALUE;
    if (params->width <= 0 || params->height <= 0) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(ctx, launchCombine(*params, inColor, inNormal, inAlbedo, outColor, outNormal, outAlbedo, s), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowFinalizeDevice(RtowContext ctx, int32_t pixelCount, const float* inColor, const float* inNormal, const float* inAlbedo,
                                uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, void* stream)
{
    if (!ctx || pixelCount <= 0 || !inColor || !inNormal || !inAlbedo || !outColor || !outNormal || !outAlbedo) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(ctx, launchFinalize(pixelCount, inColor, inNormal, inAlbedo, outColor, outNormal, outAlbedo, ctx->dByteThresholds, s), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowAddAccumDevice(RtowContext ctx, int32_t pixelCount, const RtowAccumBuffers* dst, const RtowAccumBuffers* src, void* stream)
{
    if (!ctx || !dst || !src || pixelCount <= 0) return RTOW_ERROR_INVALID_VALUE;
    if (!dst->color || !dst->normal || !dst->albedo || !dst->sampleCountWeight || !src->color || !src->normal || !src->albedo || !src->sampleCountWeight)
        return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    float* const d[4] = {dst->color, dst->normal, dst->albedo, dst->sampleCountWeight};
    const float* const q[4] = {src->color, src->normal, src->albedo, src->sampleCountWeight};
    HIP_TRY(ctx, launchAddAccum((size_t)pixelCount, d, q, s), RTOW_ERROR_LAUNCH_FAILURE);      // one launch for the four buffers
    return RTOW_SUCCESS;
}

RTOW_API int rtowCombineFinalizeDevice(RtowContext ctx, const RtowCombineParams* params, const float* inColor, const float* inNormal, const float* inAlbedo,
                                       uint8_t* outColor, uint8_t* outNormal, uint8_t* outAlbedo, void* stream)
{
    if (!ctx || !params || !inColor || !inNormal || !inAlbedo || !outColor || !outNormal || !outAlbedo) return RTOW_ERROR_INVALID_VALUE;
    if (params->width <= 0 || params->height <= 0) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(ctx, launchCombineFinalize(*params, inColor, inNormal, inAlbedo, outColor, outNormal, outAlbedo, ctx->dByteThresholds, s), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowReduceMetricsDeviceAsync(RtowContext ctx, int32_t pixelCount, const void* diagnostics, int32_t diagnosticsStride, const float* color,
                                          const float* sampleCountWeight, void* stream, RtowMetrics* outMetrics)
{
    if (!ctx || !diagnostics || !color || !sampleCountWeight || !outMetrics || pixelCount <= 0) return RTOW_ERROR_INVALID_VALUE;
    if (diagnosticsStride != 4 && diagnosticsStride != 16) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // where the record goes: memory registered with rtowRegisterHostBuffer (the device writes it over PCIe), else a device pointer (any HIP allocation)
    RtowMetrics* target = (RtowMetrics*)mappedHost(ctx, outMetrics, sizeof(RtowMetrics));
    if (!target) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, outMetrics) != hipSuccess) { (void)hipGetLastError(); return RTOW_ERROR_INVALID_VALUE; }   // plain pageable host memory: the device cannot write it
        if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged) return RTOW_ERROR_INVALID_VALUE;
        target = attr.type == hipMemoryTypeHost && attr.devicePointer ? (RtowMetrics*)attr.devicePointer : outMetrics;
    }
    // the partials are one block per context: this reduction starts after the previous one's fold has read them
    if (ctx->haveMetricsDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evMetricsDone, 0), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, launchReduceMetrics(pixelCount, (const uint8_t*)diagnostics, diagnosticsStride, color, sampleCountWeight, ctx->dPartials, s), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, launchFoldMetrics(ctx->dPartials, target, s), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipEventRecord(ctx->evMetricsDone, s), RTOW_ERROR_LAUNCH_FAILURE);
    ctx->haveMetricsDone = true;
    return RTOW_SUCCESS;
}

RTOW_API int rtowCommSetLibraryPath(const char* path)
{
    std::lock_guard<std::mutex> lock(gRcclMu);
    if (gRcclLoaded) return RTOW_ERROR_INVALID_VALUE;              // loaded once per process: the choice comes before the first rtowComm* call
    gRcclPath = path ? path : "";
    return RTOW_SUCCESS;
}

RTOW_API int rtowCommGetUniqueId(RtowCommId* outId)
{
    if (!outId) return RTOW_ERROR_INVALID_VALUE;
    RcclApi* api = rccl();
    if (!api) return RTOW_ERROR_UNSUPPORTED;                       // no librccl.so in this process or on the loader path
    RcclUniqueId id;
    if (api->GetUniqueId(&id) != 0) return RTOW_ERROR_LAUNCH_FAILURE;
    memcpy(outId->bytes, id.internal, sizeof(id.internal));
    return RTOW_SUCCESS;
}

RTOW_API int rtowCommInit(RtowContext ctx, const RtowCommId* id, int32_t rank, int32_t worldSize)
{
    if (!ctx || !id || worldSize < 1 || rank < 0 || rank >= worldSize) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->comm) return RTOW_ERROR_INVALID_VALUE;                // one communicator per context; rtowCommDestroy first
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    RcclApi* api = rccl();
    if (!api) {
        std::lock_guard<std::mutex> l2(gRcclMu);
        logf(ctx, 2, "rccl", "the RCCL library could not be loaded: %s", gRcclLoadError.c_str());
        return RTOW_ERROR_UNSUPPORTED;
    }
    RcclUniqueId uid;
    memcpy(uid.internal, id->bytes, sizeof(uid.internal));
    void* comm = nullptr;
    RCCL_TRY(ctx, api, api->CommInitRank(&comm, worldSize, uid, rank));
    ctx->comm = comm;
    ctx->commRank = rank;
    ctx->commWorld = worldSize;
    logf(ctx, 4, "rccl", "rank %d of %d joined", rank, worldSize);
    return RTOW_SUCCESS;
}

RTOW_API int rtowCommDestroy(RtowContext ctx)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->comm) return RTOW_SUCCESS;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    (void)hipDeviceSynchronize();
    RcclApi* api = rccl();
    if (api) (void)api->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->commRank = 0;
    ctx->commWorld = 1;
    return RTOW_SUCCESS;
}

RTOW_API int rtowGatherRowsDevice(RtowContext ctx, int32_t width, int32_t height, int32_t sliceDivider, const RtowAccumBuffers* mine,
                                  const RtowAccumBuffers* frame, int32_t what, int32_t root, void* stream)
{
    if (!ctx || !mine || width <= 0 || height <= 0 || sliceDivider < 1 || (what & ~(RTOW_GATHER_ALL | RTOW_GATHER_NO_BATCH_WAIT | RTOW_GATHER_LOOPBACK)) || !(what & RTOW_GATHER_ALL)) return RTOW_ERROR_INVALID_VALUE;
    const bool waitForBatch = !(what & RTOW_GATHER_NO_BATCH_WAIT);
    const bool wantLoopback = (what & RTOW_GATHER_LOOPBACK) != 0;
    what &= RTOW_GATHER_ALL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int world = ctx->comm ? ctx->commWorld : 1, rank = ctx->comm ? ctx->commRank : 0;
    const bool loopback = wantLoopback && ctx->comm && world == 1;      // one rank sending its rows to itself through the transport (RTOW_GATHER_LOOPBACK)
    if (sliceDivider != world || root < 0 || root >= world) return RTOW_ERROR_INVALID_VALUE;   // rank g owns the rows of slice g: one slice per rank
    if (rank == root && !frame) return RTOW_ERROR_INVALID_VALUE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // the rows being gathered were written by the last sample batch, whatever stream that was enqueued on (RTOW_GATHER_NO_BATCH_WAIT: they were not)
    if (waitForBatch && ctx->haveBatchDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evBatchDone, 0), RTOW_ERROR_LAUNCH_FAILURE);

    static const int kComponents[4] = {4, 3, 3, 1};
    float* const mineBuf[4] = {mine->color, mine->normal, mine->albedo, mine->sampleCountWeight};
    float* const frameBuf[4] = {frame ? frame->color : nullptr, frame ? frame->normal : nullptr, frame ? frame->albedo : nullptr, frame ? frame->sampleCountWeight : nullptr};
    unsigned floatsPerPixel = 0;
    for (int b = 0; b < 4; b++)
        if (what & (1 << b)) {
            if (!mineBuf[b] || (rank == root && !frameBuf[b])) return RTOW_ERROR_INVALID_VALUE;
            floatsPerPixel += (unsigned)kComponents[b];
        }
    auto packedFloats = [&](int r) { return (size_t)rowsOwnedBy(r, world, height) * (size_t)width * floatsPerPixel; };

    if (loopback) {
        // the whole transport path of a peer AND of the root, against itself: pack -> {ncclSend, ncclRecv} to / from rank 0 in one group -> scatter
        RcclApi* api = rccl();
        if (!api) return RTOW_ERROR_UNSUPPORTED;
        if (ctx->haveGatherDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evGatherDone, 0), RTOW_ERROR_LAUNCH_FAILURE);
        const size_t need = packedFloats(0);
        if (need > ctx->gatherSendFloats || need > ctx->gatherRecvFloats) {
            HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
            if (ctx->dGatherSend) (void)hipFree(ctx->dGatherSend);
            if (ctx->dGatherRecv) (void)hipFree(ctx->dGatherRecv);
            ctx->dGatherSend = ctx->dGatherRecv = nullptr; ctx->gatherSendFloats = ctx->gatherRecvFloats = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dGatherSend, need * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->gatherSendFloats = need;
            HIP_TRY(ctx, hipMalloc(&ctx->dGatherRecv, need * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->gatherRecvFloats = need;
        }
        size_t at = 0;
        for (int b = 0; b < 4; b++)
            if (what & (1 << b)) {
                HIP_TRY(ctx, launchCopyRows(mineBuf[b], ctx->dGatherSend + at, (unsigned)(width * kComponents[b]), (unsigned)height, 0u, 1u, false, s), RTOW_ERROR_LAUNCH_FAILURE);
                at += (size_t)height * width * kComponents[b];
            }
        RCCL_TRY(ctx, api, api->GroupStart());
        int posted = api->Send(ctx->dGatherSend, need, kRcclFloat32, 0, ctx->comm, s);
        if (posted == 0) posted = api->Recv(ctx->dGatherRecv, need, kRcclFloat32, 0, ctx->comm, s);
        const int closed = api->GroupEnd();
        if (posted != 0 || closed != 0) {
            logf(ctx, 2, "rccl", "loop-back gather failed: ncclSend / ncclRecv %s, ncclGroupEnd %s", api->GetErrorString(posted), api->GetErrorString(closed));
            return RTOW_ERROR_LAUNCH_FAILURE;
        }
        at = 0;
        for (int b = 0; b < 4; b++)
            if (what & (1 << b)) {
                HIP_TRY(ctx, launchCopyRows(frameBuf[b], ctx->dGatherRecv + at, (unsigned)(width * kComponents[b]), (unsigned)height, 0u, 1u, true, s), RTOW_ERROR_LAUNCH_FAILURE);
                at += (size_t)height * width * kComponents[b];
            }
        HIP_TRY(ctx, hipEventRecord(ctx->evGatherDone, s), RTOW_ERROR_LAUNCH_FAILURE);
        ctx->haveGatherDone = true;
        return RTOW_SUCCESS;
    }
    if (world == 1 || rank == root) {
        // the root's own rows: already in place when frame == mine, else copied row by row on the device
        const unsigned rows = rowsOwnedBy(rank, world, height);
        for (int b = 0; b < 4; b++)
            if ((what & (1 << b)) && frameBuf[b] != mineBuf[b]) {
                const size_t rowBytes = (size_t)width * kComponents[b] * 4u;
                HIP_TRY(ctx, hipMemcpy2DAsync((uint8_t*)frameBuf[b] + (size_t)rank * rowBytes, (size_t)world * rowBytes, (const uint8_t*)mineBuf[b] + (size_t)rank * rowBytes,
                                              (size_t)world * rowBytes, rowBytes, rows, hipMemcpyDeviceToDevice, s), RTOW_ERROR_LAUNCH_FAILURE);
            }
        if (world == 1) return RTOW_SUCCESS;
    }
    RcclApi* api = rccl();
    if (!api) return RTOW_ERROR_UNSUPPORTED;
    // the packed-row staging is one block per context: a gather repacks it only after the previous gather's send / scatter is over,
    // whatever stream that one was given
    if (ctx->haveGatherDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evGatherDone, 0), RTOW_ERROR_LAUNCH_FAILURE);
    auto gatherEnds = [&]() -> int {
        HIP_TRY(ctx, hipEventRecord(ctx->evGatherDone, s), RTOW_ERROR_LAUNCH_FAILURE);
        ctx->haveGatherDone = true;
        return RTOW_SUCCESS;
    };

    if (rank != root) {
        // pack this rank's rows of the selected buffers back to back, one send to the root over this GPU's own xGMI link to it
        const size_t need = packedFloats(rank);
        if (need > ctx->gatherSendFloats) {
            HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);          // the old block may still be travelling
            if (ctx->dGatherSend) (void)hipFree(ctx->dGatherSend);
            ctx->dGatherSend = nullptr; ctx->gatherSendFloats = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dGatherSend, need * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->gatherSendFloats = need;
        }
        size_t at = 0;
        const unsigned rows = rowsOwnedBy(rank, world, height);
        for (int b = 0; b < 4; b++)
            if (what & (1 << b)) {
                HIP_TRY(ctx, launchCopyRows(mineBuf[b], ctx->dGatherSend + at, (unsigned)(width * kComponents[b]), rows, (unsigned)rank, (unsigned)world, false, s), RTOW_ERROR_LAUNCH_FAILURE);
                at += (size_t)rows * width * kComponents[b];
            }
        if (need) RCCL_TRY(ctx, api, api->Send(ctx->dGatherSend, need, kRcclFloat32, root, ctx->comm, s));
        return gatherEnds();
    }

    // root: one receive per peer into its own region of the staging block (posted as one group: all seven links run at once), then scatter
    size_t total = 0;
    std::vector<size_t> offset((size_t)world, 0);
    for (int r = 0; r < world; r++) { offset[(size_t)r] = total; if (r != root) total += packedFloats(r); }
    if (total > ctx->gatherRecvFloats) {
        HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
        if (ctx->dGatherRecv) (void)hipFree(ctx->dGatherRecv);
        ctx->dGatherRecv = nullptr; ctx->gatherRecvFloats = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dGatherRecv, total * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
        ctx->gatherRecvFloats = total;
    }
    RCCL_TRY(ctx, api, api->GroupStart());
    int posted = 0;                                       // a failed ncclRecv must not leave the communicator's group open: it is closed on every path
    for (int r = 0; r < world && posted == 0; r++)
        if (r != root && packedFloats(r)) posted = api->Recv(ctx->dGatherRecv + offset[(size_t)r], packedFloats(r), kRcclFloat32, r, ctx->comm, s);
    const int closed = api->GroupEnd();
    if (posted != 0 || closed != 0) {
        logf(ctx, 2, "rccl", "gather on the root failed: ncclRecv %s, ncclGroupEnd %s", api->GetErrorString(posted), api->GetErrorString(closed));
        return RTOW_ERROR_LAUNCH_FAILURE;
    }
    for (int r = 0; r < world; r++) {
        if (r == root) continue;
        size_t at = offset[(size_t)r];
        const unsigned rows = rowsOwnedBy(r, world, height);
        for (int b = 0; b < 4; b++)
            if (what & (1 << b)) {
                HIP_TRY(ctx, launchCopyRows(frameBuf[b], ctx->dGatherRecv + at, (unsigned)(width * kComponents[b]), rows, (unsigned)r, (unsigned)world, true, s), RTOW_ERROR_LAUNCH_FAILURE);
                at += (size_t)rows * width * kComponents[b];
            }
    }
    return gatherEnds();
}

RTOW_API int rtowHybridPlan(int32_t worldSize, int32_t rank, int32_t tileCount, uint32_t samplesPerBatch, uint32_t step, RtowHybridPlan* out)
{
    if (!out || worldSize < 1 || rank < 0 || rank >= worldSize || tileCount < 1 || worldSize % tileCount != 0 || step < 1u) return RTOW_ERROR_INVALID_VALUE;
    const int32_t groups = worldSize / tileCount;
    RtowHybridPlan p{};
    p.tileCount = tileCount;
    p.groupCount = groups;
    p.tile = rank % tileCount;
    p.group = rank / tileCount;
    p.sliceOffset = p.tile;
    p.sliceDivider = tileCount;
    p.samples = samplesPerBatch / (uint32_t)groups + ((uint32_t)p.group < samplesPerBatch % (uint32_t)groups ? 1u : 0u);
    p.seed = (step - 1u) * (uint32_t)groups + (uint32_t)p.group + 1u;
    *out = p;
    return RTOW_SUCCESS;
}

RTOW_API int rtowExchangeAccumDevice(RtowContext ctx, int32_t width, int32_t height, int32_t tileCount, const RtowAccumBuffers* partial, const RtowAccumBuffers* accum,
                                     int32_t what, void* stream)
{
    if (!ctx || !partial || !accum || width <= 0 || height <= 0 || tileCount < 1 || (what & ~(RTOW_GATHER_ALL | RTOW_GATHER_NO_BATCH_WAIT)) || !(what & RTOW_GATHER_ALL))
        return RTOW_ERROR_INVALID_VALUE;
    const bool waitForBatch = !(what & RTOW_GATHER_NO_BATCH_WAIT);
    what &= RTOW_GATHER_ALL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int world = ctx->comm ? ctx->commWorld : 1, rank = ctx->comm ? ctx->commRank : 0;
    if (world % tileCount != 0) return RTOW_ERROR_INVALID_VALUE;                      // G = T x B
    const int groups = world / tileCount, tile = rank % tileCount, own = rank / tileCount;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // the partial sums were written by the last sample batch, whatever stream that was enqueued on (RTOW_GATHER_NO_BATCH_WAIT: the caller ordered it)
    if (waitForBatch && ctx->haveBatchDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evBatchDone, 0), RTOW_ERROR_LAUNCH_FAILURE);

    static const int kComponents[4] = {4, 3, 3, 1};
    float* const partBuf[4] = {partial->color, partial->normal, partial->albedo, partial->sampleCountWeight};
    float* const accBuf[4] = {accum->color, accum->normal, accum->albedo, accum->sampleCountWeight};
    unsigned floatsPerPixel = 0;
    for (int b = 0; b < 4; b++)
        if (what & (1 << b)) {
            if (!partBuf[b] || !accBuf[b] || partBuf[b] == accBuf[b]) return RTOW_ERROR_INVALID_VALUE;      // the fold reads partial rows while it writes accum rows
            floatsPerPixel += (unsigned)kComponents[b];
        }
    // rank p folds the rows with row % G == p; they all lie in the tile p % T, i.e. in what every rank of that tile rendered
    auto packedFloats = [&](int r) { return (size_t)rowsOwnedBy(r, world, height) * (size_t)width * floatsPerPixel; };
    const unsigned myRows = rowsOwnedBy(rank, world, height);
    const size_t regionFloats = packedFloats(rank);                                     // what every peer of the tile sends here: this rank's rows of ITS partial

    if (groups > 1) {
        RcclApi* api = rccl();
        if (!api) return RTOW_ERROR_UNSUPPORTED;
        // staging (shared with rtowGatherRowsDevice, ordered by the same event): packed rows for every peer | one region per group for what arrives
        if (ctx->haveGatherDone) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->evGatherDone, 0), RTOW_ERROR_LAUNCH_FAILURE);
        size_t sendTotal = 0;
        std::vector<size_t> sendOffset((size_t)groups, 0);
        for (int g = 0; g < groups; g++) { sendOffset[(size_t)g] = sendTotal; if (g != own) sendTotal += packedFloats(tile + tileCount * g); }
        const size_t recvTotal = regionFloats * (size_t)groups;
        if (sendTotal > ctx->gatherSendFloats) {
            HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);          // the old block may still be travelling
            if (ctx->haveGatherDone) HIP_TRY(ctx, hipEventSynchronize(ctx->evGatherDone), RTOW_ERROR_LAUNCH_FAILURE);
            if (ctx->dGatherSend) (void)hipFree(ctx->dGatherSend);
            ctx->dGatherSend = nullptr; ctx->gatherSendFloats = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dGatherSend, sendTotal * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->gatherSendFloats = sendTotal;
        }
        if (recvTotal > ctx->gatherRecvFloats) {
            HIP_TRY(ctx, hipStreamSynchronize(s), RTOW_ERROR_LAUNCH_FAILURE);
            if (ctx->haveGatherDone) HIP_TRY(ctx, hipEventSynchronize(ctx->evGatherDone), RTOW_ERROR_LAUNCH_FAILURE);
            if (ctx->dGatherRecv) (void)hipFree(ctx->dGatherRecv);
            ctx->dGatherRecv = nullptr; ctx->gatherRecvFloats = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dGatherRecv, recvTotal * 4u), RTOW_ERROR_MEMORY_ALLOCATION);
            ctx->gatherRecvFloats = recvTotal;
        }
        for (int g = 0; g < groups; g++) {
            if (g == own) continue;
            const int peer = tile + tileCount * g;
            const unsigned rows = rowsOwnedBy(peer, world, height);
            size_t at = sendOffset[(size_t)g];
            for (int b = 0; b < 4; b++)
                if (what & (1 << b)) {
                    HIP_TRY(ctx, launchCopyRows(partBuf[b], ctx->dGatherSend + at, (unsigned)(width * kComponents[b]), rows, (unsigned)peer, (unsigned)world, false, s), RTOW_ERROR_LAUNCH_FAILURE);
                    at += (size_t)rows * width * kComponents[b];
                }
        }
        // one group: a send and a receive per peer of the tile, each pair on its own xGMI link.  The group is closed on every path.
        RCCL_TRY(ctx, api, api->GroupStart());
        int posted = 0;
        for (int g = 0; g < groups && posted == 0; g++) {
            if (g == own) continue;
            const int peer = tile + tileCount * g;
            if (packedFloats(peer)) posted = api->Send(ctx->dGatherSend + sendOffset[(size_t)g], packedFloats(peer), kRcclFloat32, peer, ctx->comm, s);
            if (posted == 0 && regionFloats) posted = api->Recv(ctx->dGatherRecv + (size_t)g * regionFloats, regionFloats, kRcclFloat32, peer, ctx->comm, s);
        }
        const int closed = api->GroupEnd();
        if (posted != 0 || closed != 0) {
            logf(ctx, 2, "rccl", "exchange of partial sums failed: ncclSend / ncclRecv %s, ncclGroupEnd %s", api->GetErrorString(posted), api->GetErrorString(closed));
            return RTOW_ERROR_LAUNCH_FAILURE;
        }
    }
    // the fold: this rank's rows, group order, own partial in place
    size_t at = 0;
    for (int b = 0; b < 4; b++)
        if (what & (1 << b)) {
            HIP_TRY(ctx, launchFoldRows(accBuf[b], partBuf[b], groups > 1 ? ctx->dGatherRecv + at : partBuf[b], regionFloats, (unsigned)(width * kComponents[b]), myRows, (unsigned)rank,
                                        (unsigned)world, (unsigned)groups, (unsigned)own, s), RTOW_ERROR_LAUNCH_FAILURE);
            at += (size_t)myRows * width * kComponents[b];
        }
    if (groups > 1) {
        HIP_TRY(ctx, hipEventRecord(ctx->evGatherDone, s), RTOW_ERROR_LAUNCH_FAILURE);
        ctx->haveGatherDone = true;
    }
    return RTOW_SUCCESS;
}

RTOW_API int rtowDeviceAlloc(RtowContext ctx, size_t sizeInBytes, void** outPointer)
{
    if (!ctx || !outPointer || sizeInBytes == 0) return RTOW_ERROR_INVALID_VALUE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipMalloc(outPointer, sizeInBytes), RTOW_ERROR_MEMORY_ALLOCATION);
    return RTOW_SUCCESS;
}

RTOW_API int rtowDeviceFree(RtowContext ctx, void* pointer)
{
    if (!ctx || !pointer) return RTOW_ERROR_INVALID_VALUE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipFree(pointer), RTOW_ERROR_INVALID_VALUE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowDeviceCopy(RtowContext ctx, const void* source, void* destination, size_t sizeInBytes, int kind)
{
    if (!ctx || !source || !destination) return RTOW_ERROR_INVALID_VALUE;
    hipMemcpyKind k;
    switch (kind) {
        case RTOW_MEMCPY_HOST_TO_HOST: k = hipMemcpyHostToHost; break;
        case RTOW_MEMCPY_HOST_TO_DEVICE: k = hipMemcpyHostToDevice; break;
        case RTOW_MEMCPY_DEVICE_TO_HOST: k = hipMemcpyDeviceToHost; break;
        case RTOW_MEMCPY_DEVICE_TO_DEVICE: k = hipMemcpyDeviceToDevice; break;
        default: return RTOW_ERROR_INVALID_VALUE;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipMemcpy(destination, source, sizeInBytes, k), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowDeviceMemset(RtowContext ctx, void* pointer, int value, size_t sizeInBytes)
{
    if (!ctx || !pointer) return RTOW_ERROR_INVALID_VALUE;
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipMemsetAsync(pointer, value, sizeInBytes, ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    return RTOW_SUCCESS;
}

RTOW_API int rtowGetBatchStatus(RtowContext ctx)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    // the last batch may have been enqueued on a caller's stream: its end is evBatchDone, not the end of ctx->stream
    if (ctx->haveBatchDone) HIP_TRY(ctx, hipEventSynchronize(ctx->evBatchDone), RTOW_ERROR_LAUNCH_FAILURE);
    return takeOverflow(ctx);
}

RTOW_API int rtowSynchronize(RtowContext ctx)
{
    if (!ctx) return RTOW_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device), RTOW_ERROR_NO_DEVICE);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream), RTOW_ERROR_LAUNCH_FAILURE);
    if (ctx->haveBatchDone) HIP_TRY(ctx, hipEventSynchronize(ctx->evBatchDone), RTOW_ERROR_LAUNCH_FAILURE);
    return takeOverflow(ctx);
}

} // extern "C"
