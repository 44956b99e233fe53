// rtow_reforder.h - the order in which the reference enumerates entities that a ray hits at bit-identical distances.
//
// In the reference the hit list of a ray is the reversed candidate list (JOBS/SampleBatchJob.cs:450-475), candidates come out
// of its tree right-to-left (:403-447), so hits start out in the tree's left-to-right leaf order == the order of the
// re-ordered entity array (UNITY/BvhNodeData.cs:157-160), and are then put through NativeSortExtension.Sort, which is not
// stable.  Scenes with ProbabilisticVolume materials look at every hit of a ray, and coplanar surfaces (a fog box standing
// on the floor) produce exact ties, so that order is part of the result.  This file computes the leaf order (the rank of
// every entity) without building the reference tree; the kernel applies the same small-array sort to (distance) on hits
// pre-ordered by rank.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rtow {

// boxes: float[8] {min.xyz, -, max.xyz, -} per entity, the reference's own fp32 entity boxes (UNITY/BvhNodeData.cs:23-81).
// maxDepth: the host's MaxBvhDepth (leaves are forced at that depth).  Returns rank[entity] in 0..n-1.  leafBoxes (optional, same layout
// as boxes): the bounds of the reference leaf each entity ends up in - its own box, or the union of the boxes of a leaf forced at maxDepth.
// tree (optional): the reference tree itself, node 0 = root, for the FULL_DIAGNOSTICS counters that count ITS boxes and ITS leaves
// (JOBS/SampleBatchJob.cs:427-440): bounds, two child indices, or - for a leaf - the number of entities it holds.
struct RefTreeNode {
    float lo[3];
    int32_t left;      // >= 0: inner node, index of Left; < 0: leaf holding ~left entities
    float hi[3];
    int32_t right;     // inner node: index of Right
};
static_assert(sizeof(RefTreeNode) == 32, "RefTreeNode is read by the kernel as two float4");
std::vector<uint32_t> referenceLeafRanks(const std::vector<float>& boxes, int n, int maxDepth, std::vector<float>* leafBoxes = nullptr,
                                         std::vector<RefTreeNode>* tree = nullptr);

// The introsort of com.unity.collections 1.0.0-pre.6 (NativeSortExtension.Sort), on an index array with float keys and the
// comparer `(int) sign(key[l] - key[r])` (UNITY/BvhNodeData.cs:240-250).  Exposed for the unit tests.
void referenceIndexSort(uint32_t* idx, int length, const float* key);

} // namespace rtow
