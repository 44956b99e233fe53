// Test-only C wrapper around the product's scene compiler (csrc/rtow_bvh.cpp: SAH builder, flat layout), so that the CPU suite can hold
// it to its contract without a GPU: depth bound, node count, breadth-first child order, boxes that enclose their subtrees, determinism.
// Built by tests/test_scene_compiler.py with g++.
#include <cstring>

#include "../../raytracing-in-one-weekend_amd/csrc/rtow_bvh.h"

extern "C" int shim_compile_scene(const RtowSceneDesc* desc, int maxDepth, uint8_t* blobOut, uint32_t blobCapacity, rtow::SceneLayout* layoutOut)
{
    rtow::CompiledScene cs;
    std::string err;
    const int rc = rtow::compileScene(desc, maxDepth, &cs, &err);
    if (rc != RTOW_SUCCESS) return rc;
    *layoutOut = cs.layout;
    if (cs.blob.size() > blobCapacity) return -1;
    memcpy(blobOut, cs.blob.data(), cs.blob.size());
    return 0;
}

// the reference's own tree as the product replays it (rtow_reforder.h RefTreeNode[]): 8 int32 / float words per node
extern "C" int shim_reference_tree(const RtowSceneDesc* desc, uint8_t* out, uint32_t capacityBytes)
{
    rtow::CompiledScene cs;
    std::string err;
    const int rc = rtow::compileScene(desc, 24, &cs, &err);
    if (rc != RTOW_SUCCESS) return -rc;
    if (cs.refTree.size() > capacityBytes) return -1000;
    memcpy(out, cs.refTree.data(), cs.refTree.size());
    return (int)(cs.refTree.size() / 32);
}
