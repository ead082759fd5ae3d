// kd_build.h -- host kd-tree builder interface (see kd_build.cpp).
#pragma once
#include <cstdint>
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

}  // namespace mipt
