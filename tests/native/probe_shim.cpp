// Test-only C wrapper around the product's host-side ray probe (csrc/rtow_probe.hip: probeNearestHitHost, what rtowProbeNearestHit runs) over the product's own scene
// compiler, so that the CPU suite can hold it to the oracle's Raytracer.HitWorld without a GPU.  Only scene kinds whose host image is complete without the device
// (identity-rotation spheres, moving spheres, triangles: the derived inverse transforms of rotated / translated entities are computed on the device at upload).
// Built by tests/test_hit_world_oracle.py: rtow_probe.hip with hipcc --offload-host-only, the rest with g++.
#include <cstring>
#include <string>

#include "../../raytracing-in-one-weekend_amd/csrc/rtow_bvh.h"

namespace rtow {
bool probeNearestHitHost(const uint8_t* blob, const SceneLayout& L, const int32_t* entityOfPrim, const float origin[3], const float direction[3], float time, float* distance, int* entity);
}

static rtow::CompiledScene g_scene;

extern "C" int shim_probe_compile(const RtowSceneDesc* desc)
{
    std::string err;
    const int rc = rtow::compileScene(desc, RTOW_DEFAULT_MAX_BVH_DEPTH, &g_scene, &err);
    return rc != RTOW_SUCCESS ? -rc : (int)g_scene.layout.sceneKind;
}

extern "C" int shim_probe(const float* origin, const float* direction, float time, float* distance, int* entity)
{
    return rtow::probeNearestHitHost(g_scene.blob.data(), g_scene.layout, g_scene.entityOfPrim.empty() ? nullptr : g_scene.entityOfPrim.data(), origin, direction, time, distance, entity) ? 1 : 0;
}
