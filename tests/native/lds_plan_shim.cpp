// Test-only C wrapper around the per-launch LDS plan and the history-width rule of the sample kernel's launcher (csrc/rtow_kernels.h: planLds, historyWords), so that the
// CPU suite can hold them to their contract without a GPU.  Built by tests/test_lds_plan.py with hipcc --offload-host-only (the header pulls in hip_runtime.h).
#include "../../raytracing-in-one-weekend_amd/csrc/rtow_kernels.h"

extern "C" int shim_plan_lds(int wide, unsigned bvhDepth, unsigned totalBytes, unsigned nodeCount, unsigned histRows, unsigned budgetOverride, unsigned* out /* [8] */)
{
    rtow::SceneLayout L{};
    L.bvhDepth = bvhDepth; L.totalBytes = totalBytes; L.nodeCount = nodeCount;
    const rtow::LdsPlan p = rtow::planLds(wide != 0, L, histRows, budgetOverride);
    out[0] = p.stackRows; out[1] = p.histOffset; out[2] = p.histRows; out[3] = p.frontBytes; out[4] = p.sceneBytes; out[5] = p.nodeCount; out[6] = p.allLds ? 1u : 0u; out[7] = p.histSpillRows;
    return 0;
}
extern "C" int shim_history_words(int noise, int perSample, int wide, int ties, int fullDiag, int depth) { return rtow::historyWords(noise, perSample != 0, wide != 0, ties != 0, fullDiag != 0, depth); }
extern "C" int shim_constants(int* out /* [5] */) { out[0] = rtow::kLdsBytesMax; out[1] = rtow::kQueueBytes; out[2] = rtow::kCandCapacity; out[3] = rtow::kBlockThreads; out[4] = rtow::kHistoryInRegisters; return 0; }
