// rtow_reforder.cpp - see rtow_reforder.h.
#include "rtow_reforder.h"

#include <cfloat>
#include <cstddef>
#include <utility>

namespace rtow {

namespace {

// Sorter over idx[] keyed by key[idx[i]].  Published algorithm (the .NET Core ArraySortHelper introsort that
// com.unity.collections' NativeSortExtension carries): ranges of up to 16 elements are finished directly - 2 and 3 elements
// by compare-exchange, 4..16 by insertion - larger ranges are split around a median-of-three pivot, the right part first;
// when the split budget 2*floor(log2 n) runs out the range is heap-sorted.
struct IndexSorter {
    uint32_t* idx;
    const float* key;

    int order(uint32_t l, uint32_t r) const
    {
        const float d = key[l] - key[r];
        return d > 0.0f ? 1 : d < 0.0f ? -1 : 0;
    }
    void exchangeIfAbove(int i, int j)
    {
        if (i != j && order(idx[i], idx[j]) > 0) std::swap(idx[i], idx[j]);
    }
    void insertion(int first, int last)
    {
        for (int k = first + 1; k <= last; k++) {
            const uint32_t moving = idx[k];
            int slot = k - 1;
            for (; slot >= first && order(moving, idx[slot]) < 0; slot--) idx[slot + 1] = idx[slot];
            idx[slot + 1] = moving;
        }
    }
    void siftDown(int node, int count, int first)     // 1-based heap positions inside [first, first + count)
    {
        const uint32_t moving = idx[first + node - 1];
        while (node <= count / 2) {
            int kid = 2 * node;
            if (kid < count && order(idx[first + kid - 1], idx[first + kid]) < 0) kid++;
            if (order(idx[first + kid - 1], moving) < 0) break;
            idx[first + node - 1] = idx[first + kid - 1];
            node = kid;
        }
        idx[first + node - 1] = moving;
    }
    void heap(int first, int last)
    {
        const int count = last - first + 1;
        for (int node = count / 2; node >= 1; node--) siftDown(node, count, first);
        for (int live = count; live > 1; live--) {
            std::swap(idx[first], idx[first + live - 1]);
            siftDown(1, live - 1, first);
        }
    }
    int split(int first, int last)
    {
        const int middle = first + (last - first) / 2;
        exchangeIfAbove(first, middle);
        exchangeIfAbove(first, last);
        exchangeIfAbove(middle, last);
        const uint32_t pivot = idx[middle];
        std::swap(idx[middle], idx[last - 1]);
        int up = first, down = last - 1;
        while (up < down) {
            do up++; while (order(pivot, idx[up]) > 0);
            do down--; while (order(pivot, idx[down]) < 0);
            if (up >= down) break;
            std::swap(idx[up], idx[down]);
        }
        std::swap(idx[up], idx[last - 1]);
        return up;
    }
    void run(int first, int last, int budget)
    {
        while (last > first) {
            const int count = last - first + 1;
            if (count <= 16) {
                if (count == 2) exchangeIfAbove(first, last);
                else if (count == 3) { exchangeIfAbove(first, last - 1); exchangeIfAbove(first, last); exchangeIfAbove(last - 1, last); }
                else insertion(first, last);
                return;
            }
            if (budget == 0) { heap(first, last); return; }
            budget--;
            const int p = split(first, last);
            run(p + 1, last, budget);
            last = p - 1;
        }
    }
};

} // namespace

void referenceIndexSort(uint32_t* idx, int length, const float* key)
{
    if (length < 2) return;
    int levels = 0;
    for (int v = length; v > 1; v >>= 1) levels++;
    IndexSorter{idx, key}.run(0, length - 1, 2 * levels);
}

std::vector<uint32_t> referenceLeafRanks(const std::vector<float>& boxes, int n, int maxDepth, std::vector<float>* leafBoxes, std::vector<RefTreeNode>* tree)
{
    if (leafBoxes) leafBoxes->assign((size_t)n * 8, 0.0f);
    if (tree) { tree->clear(); tree->push_back(RefTreeNode{}); }
    // The reference's recursion (UNITY/BvhNodeData.cs:122-213) sorts and splits sub-ranges of one array in place, left child =
    // the front part; the array it ends with IS the leaf order.  Only the range bookkeeping is repeated here.
    std::vector<uint32_t> order(n);
    for (int i = 0; i < n; i++) order[i] = (uint32_t)i;
    std::vector<float> key(n);
    struct Range { int begin, end, depth, sortedAxis, node; };
    std::vector<Range> todo;
    todo.push_back(Range{0, n, 0, -1, 0});
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        const int count = r.end - r.begin;
        if (count <= 0) continue;
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = r.begin; i < r.end; i++) {
            const float* b = &boxes[(size_t)order[i] * 8];
            for (int a = 0; a < 3; a++) {
                if (b[a] < lo[a]) lo[a] = b[a];
                if (b[4 + a] > hi[a]) hi[a] = b[4 + a];
            }
        }
        if (tree) {
            // BvhNodeData.Bounds: a leaf encloses its entities (:161-166), an inner node its two children (:198) - min / max, the same box either way
            RefTreeNode& t = (*tree)[(size_t)r.node];
            for (int a = 0; a < 3; a++) { t.lo[a] = lo[a]; t.hi[a] = hi[a]; }
        }
        int axis = -1;
        float widest = -FLT_MAX;
        for (int a = 0; a < 3; a++) {
            const float w = hi[a] - lo[a];
            if (w > widest) { widest = w; axis = a; }
        }
        if (axis >= 0 && axis != r.sortedAxis) {                       // :146-151
            for (int i = r.begin; i < r.end; i++) key[order[i]] = boxes[(size_t)order[i] * 8 + axis];
            referenceIndexSort(order.data() + r.begin, count, key.data());
        }
        if (r.depth == maxDepth || count <= 1) {                       // leaf (:155-167)
            // a leaf's entities are appended to the candidate list front to back and the hit loop pops that list from its end
            // (JOBS/SampleBatchJob.cs:436-441,452-455): inside one leaf the hits come out back to front
            for (int i = r.begin, j = r.end - 1; i < j; i++, j--) std::swap(order[i], order[j]);
            if (tree) { (*tree)[(size_t)r.node].left = ~count; (*tree)[(size_t)r.node].right = 0; }
            // the leaf's bounds (:161-166) guard every entity in it: a ray reaches an entity's exact test iff it passes THIS box
            if (leafBoxes)
                for (int i = r.begin; i < r.end; i++) {
                    float* g = &(*leafBoxes)[(size_t)order[i] * 8];
                    for (int a = 0; a < 3; a++) { g[a] = lo[a]; g[4 + a] = hi[a]; }
                }
            continue;
        }
        // :170-196: the front part ends with the first entity that starts beyond, or is itself wider than, half the range
        int front = 0;
        const float start = boxes[(size_t)order[r.begin] * 8 + axis];
        for (int i = r.begin; i < r.end; i++) {
            front++;
            const float* b = &boxes[(size_t)order[i] * 8];
            if (b[axis] - start > widest / 2 || b[4 + axis] - b[axis] > widest / 2) break;
        }
        if (front == count) front--;
        int leftNode = 0, rightNode = 0;
        if (tree) {
            leftNode = (int)tree->size();
            rightNode = leftNode + 1;
            tree->push_back(RefTreeNode{});
            tree->push_back(RefTreeNode{});
            (*tree)[(size_t)r.node].left = leftNode;
            (*tree)[(size_t)r.node].right = rightNode;
        }
        todo.push_back(Range{r.begin + front, r.end, r.depth + 1, axis, rightNode});
        todo.push_back(Range{r.begin, r.begin + front, r.depth + 1, axis, leftNode});
    }
    std::vector<uint32_t> rank(n);
    for (int i = 0; i < n; i++) rank[order[i]] = (uint32_t)i;
    return rank;
}

} // namespace rtow
