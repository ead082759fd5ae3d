// kd_build.h -- host kd-tree builder interface (see kd_build.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "scene_types.h"

namespace mipt {

struct KdTree {
    std::vector<KdNode> nodes;    // mesh-local indices (node.b = index of the child pair: left, right = left+1), local ref offsets; slot 1 is padding
    std::vector<float> boxes;     // 8 floats per node: minX,maxX,minY,maxY,minZ,maxZ,0,0
    std::vector<int32_t> refs;    // mesh-local triangle ids
    int64_t degenerate_nodes = 0; // nodes that fell back to a leaf (no split candidate / zero width)
    int64_t n_leaves = 0, max_leaf = 0;
    int max_depth = 0;
};

// pos: n_tris*9 floats (a,b,c world-space positions).  n_threads >= 1.
void build_kd_tree(const float *pos, int64_t n_tris, KdTree &out, int n_threads);

// A tree as a builder discovers it: nodes in any order, children linked by index (-1: leaf), a leaf's triangle list at
// [leaf_begin, leaf_begin + leaf_count) of a separate array.  assemble_kd_tree() gives it the flat numbering of KdTree
// (root in slot 0, slot 1 padding, the two children of a node adjacent, depth-first, right after their parent's subtree
// order of kd_build.cpp) -- the numbering is a function of the tree alone, so every builder ends in the same arrays.
struct BuildNode {
    float lo[3], hi[3];
    int axis = 0;
    float split = 0.f;
    int left = -1, right = -1;
    int leaf_begin = 0, leaf_count = 0;
    bool degenerate = false;
};
void assemble_kd_tree(const std::vector<BuildNode> &tree, const int32_t *leaf_refs, KdTree &out);

// The same tree built on a GPU, level by level (kd_build_gpu.hip).  Returns false and sets err on a HIP error.  d_pos: the
// same positions on `device` if the caller has uploaded them already (the packer keeps them for pack_gpu.hip), or null.
bool build_kd_tree_gpu(int device, const float *pos, int64_t n_tris, KdTree &out, std::string &err, const void *d_pos = nullptr);

}  // namespace mipt
