// Test-only C wrapper around the product's reference tie-order code (csrc/rtow_reforder.cpp), so that the CPU suite can
// check it against the oracle's restatement of the reference tree.  Built by tests/test_reference_tie_order.py with g++.
#include "../../raytracing-in-one-weekend_amd/csrc/rtow_reforder.h"

extern "C" void shim_leaf_ranks(const float* boxes, int n, int maxDepth, uint32_t* out)
{
    const std::vector<float> b(boxes, boxes + (size_t)n * 8);
    const std::vector<uint32_t> r = rtow::referenceLeafRanks(b, n, maxDepth);
    for (int i = 0; i < n; i++) out[i] = r[i];
}

extern "C" void shim_index_sort(uint32_t* idx, int length, const float* key) { rtow::referenceIndexSort(idx, length, key); }

extern "C" void shim_leaf_boxes(const float* boxes, int n, int maxDepth, float* out)
{
    const std::vector<float> b(boxes, boxes + (size_t)n * 8);
    std::vector<float> leaf;
    (void)rtow::referenceLeafRanks(b, n, maxDepth, &leaf);
    for (size_t i = 0; i < leaf.size(); i++) out[i] = leaf[i];
}
