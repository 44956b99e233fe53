// rtow_bvh.h - host-side scene compiler interface (see rtow_bvh.cpp).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rtow.h"
#include "rtow_scene.h"

// Upper bound on the number of inner nodes along any root->leaf path; equals the per-lane traversal stack depth the
// kernels reserve in LDS (16-bit entries).  2^24 leaves is far above the 32767-entity cap.
#define RTOW_STACK_CAPACITY 24
#define RTOW_DEFAULT_MAX_BVH_DEPTH 24

namespace rtow {

struct CompiledScene {
    std::vector<uint8_t> blob; // device image, see SceneLayout
    SceneLayout layout;
    int entityCount = 0;
    int materialCount = 0;
    std::vector<uint8_t> texBlob; // Image-texture blob (HBM only), see TexLayout; empty when the scene has no Image texture
    TexLayout texLayout{};
    std::vector<uint8_t> refTree; // the reference's own tree (rtow_reforder.h RefTreeNode[], node 0 = root): uploaded only for RTOW_CONTEXT_REFERENCE_DIAGNOSTICS
    int refTreeDepth = 0;         // its depth bound (the host's MaxBvhDepth)
    std::vector<int32_t> entityOfPrim; // all-triangle scenes number their primitives in leaf order: primitive -> the host's entity index (empty: the same number)
};

// Returns an RtowResult; *err describes failures.
int compileScene(const RtowSceneDesc* desc, int maxDepth, CompiledScene* out, std::string* err);

} // namespace rtow
