// kd_build.cpp -- host kd-tree builder + flattening into 8-byte nodes and triangle index lists.
//
// Produces exactly the tree of Mesh::getKdTree / Node::makeNodeTree (mesh.cu:32-155): same root
// box, same split planes (Box::half recomputes the plane as lo + width*pct, mesh.cu:5-30), same
// "a triangle belongs to a child iff at least one VERTEX is inside the child's box" membership with
// the asymmetric epsilons of Box::isVertexInBox (mesh.cuh:131-146), same leaf rule (<=50 triangles
// or depth > 15, mesh.cu:33) and the same median-walk split search (mesh.cu:72-108).
//
// It is organised differently so that 1M-10M triangle meshes build in seconds instead of making
// ~2 blocking device allocations per leaf (mesh.cu:35-36):
//   * during the split search only the child bound on the split axis changes, so every triangle is
//     reduced once per node to two keys (min admissible coordinate for the left child, max for the
//     right child); each of the <=20 search iterations is then a 2-compare count instead of up to
//     36 compares per triangle -- the counts are identical because the same fp32 values are compared;
//   * subtrees are built by a bounded pool of std::threads;
//   * nothing is uploaded here: the packer (scene.cpp) places nodes/refs into the scene arena.
#include "kd_build.h"

#include <algorithm>
#include <atomic>
#include <limits>
#include <memory>
#include <thread>

namespace mipt {
namespace {

struct TNode {
    float lo[3], hi[3];
    int axis = 0;
    float split = 0.f;
    std::unique_ptr<TNode> left, right;
    std::vector<int32_t> tris;
    bool degenerate = false;
};

struct Builder {
    const float *pos;  // n*9
    std::atomic<int> spare_threads{0};

    inline const float *vtx(int32_t tri, int v) const { return pos + (size_t)tri * 9 + 3 * v; }

    void build(TNode *nd, std::vector<int32_t> &&tris_in, int depth) {
        std::vector<int32_t> tris = std::move(tris_in);
        const int n = (int)tris.size();
        if (n <= kMaxLeafTris || depth > kMaxKdHeight) {  // mesh.cu:33
            nd->tris = std::move(tris);
            return;
        }
        const int a = nd->axis;
        const float blo = nd->lo[a], bhi = nd->hi[a];
        // split candidates: vertex coordinates strictly inside the box on this axis (mesh.cu:44-64)
        std::vector<float> cand;
        cand.reserve((size_t)n * 3);
        // per-triangle keys (see file header)
        std::vector<float> key_l((size_t)n), key_r((size_t)n);
        const float inf = std::numeric_limits<float>::infinity();
        for (int i = 0; i < n; i++) {
            float kl = inf, kr = -inf;
            for (int v = 0; v < 3; v++) {
                const float *p = vtx(tris[i], v);
                const float c = p[a];
                if (c > blo && c < bhi) cand.push_back(c);
                // Box::isVertexInBox against the PARENT bounds, per axis (mesh.cuh:131-139)
                bool lo_ok[3], hi_ok[3];
                lo_ok[0] = nd->lo[0] <= p[0];
                lo_ok[1] = nd->lo[1] <= p[1] + kEps;
                lo_ok[2] = nd->lo[2] <= p[2] + kEps;
                hi_ok[0] = p[0] <= nd->hi[0] + kEps;
                hi_ok[1] = p[1] <= nd->hi[1] + kEps;
                hi_ok[2] = p[2] <= nd->hi[2] + kEps;
                const int u = (a + 1) % 3, w = (a + 2) % 3;
                const bool other = lo_ok[u] && hi_ok[u] && lo_ok[w] && hi_ok[w];
                if (other && lo_ok[a]) kl = std::min(kl, c);                 // left child keeps lo[a], hi[a] := split
                if (other && hi_ok[a]) kr = std::max(kr, a == 0 ? c : c + kEps);  // right child keeps hi[a], lo[a] := split
            }
            key_l[i] = kl;
            key_r[i] = kr;
        }
        const int n_in = (int)cand.size();
        const float width = bhi - blo;
        if (n_in == 0 || !(width > 0.0f)) {  // reference is undefined here (reads v[0] / divides 0 by 0); become a leaf
            nd->tris = std::move(tris);
            nd->degenerate = true;
            return;
        }
        std::sort(cand.begin(), cand.end());  // mesh.cu:68
        auto plane_of = [&](int idx) { return blo + width * ((cand[idx] - blo) / width); };  // mesh.cu:76-78 + 19,23,27
        int idx = n_in / 2;
        for (int i = 2; i <= kSplitSearchDepth + 1; i++) {  // mesh.cu:73-108
            const float s = plane_of(idx);
            const float s_eps = s + kEps;
            int nl = 0, nr = 0;
            for (int k = 0; k < n; k++) {
                nl += key_l[k] <= s_eps;
                nr += s <= key_r[k];
            }
            const int diff = nl - nr;
            if (diff < kAllowedDiff && -kAllowedDiff < diff) break;
            if (nl > nr) {
                idx -= n_in / (2 * i);
                if (idx < 0) { idx = 0; break; }
            } else {
                idx += n_in / (2 * i);
                if (idx > n_in - 1) { idx = n_in - 1; break; }
            }
        }
        const float s = plane_of(idx);  // mesh.cu:112-114
        const float s_eps = s + kEps;
        nd->split = s;
        std::vector<int32_t> lt, rt;
        lt.reserve((size_t)n / 2 + 16);
        rt.reserve((size_t)n / 2 + 16);
        for (int k = 0; k < n; k++) {  // mesh.cu:123-130 (straddlers go to both)
            if (key_l[k] <= s_eps) lt.push_back(tris[k]);
            if (s <= key_r[k]) rt.push_back(tris[k]);
        }
        std::vector<int32_t>().swap(tris);
        std::vector<float>().swap(cand);
        std::vector<float>().swap(key_l);
        std::vector<float>().swap(key_r);
        nd->left.reset(new TNode());
        nd->right.reset(new TNode());
        for (int k = 0; k < 3; k++) {
            nd->left->lo[k] = nd->right->lo[k] = nd->lo[k];
            nd->left->hi[k] = nd->right->hi[k] = nd->hi[k];
        }
        nd->left->hi[a] = s;
        nd->right->lo[a] = s;
        nd->left->axis = nd->right->axis = (a + 1) % 3;  // mesh.cu:132-133
        TNode *l = nd->left.get(), *r = nd->right.get();
        bool spawned = false;
        std::thread th;
        if (lt.size() + rt.size() > 8192 && spare_threads.fetch_sub(1) > 0) {
            spawned = true;
            th = std::thread([this, l, depth](std::vector<int32_t> v) { build(l, std::move(v), depth + 1); }, std::move(lt));
        } else {
            if (lt.size() + rt.size() > 8192) spare_threads.fetch_add(1);  // undo the failed reservation
        }
        if (!spawned) build(l, std::move(lt), depth + 1);
        build(r, std::move(rt), depth + 1);
        if (spawned) {
            th.join();
            spare_threads.fetch_add(1);
        }
    }
};

// Flat layout: the two children of an interior node are ADJACENT (left at c, right at c + 1, c even), so one
// aligned 16-byte load issued at the start of a traversal step brings in both child nodes while the step's
// arithmetic runs.  Slot 0 is the root, slot 1 is padding that keeps every pair 16-byte aligned.
void flatten(const TNode *nd, size_t idx, KdTree &out, int depth) {
    const float b[8] = {nd->lo[0], nd->hi[0], nd->lo[1], nd->hi[1], nd->lo[2], nd->hi[2], 0.f, 0.f};
    std::copy(b, b + 8, out.boxes.begin() + 8 * idx);
    out.max_depth = std::max(out.max_depth, depth);
    if (nd->degenerate) out.degenerate_nodes++;
    if (!nd->left) {
        KdNode k;
        k.a = (uint32_t)out.refs.size();
        k.b = kLeafTag | (uint32_t)nd->tris.size();
        out.refs.insert(out.refs.end(), nd->tris.begin(), nd->tris.end());
        out.nodes[idx] = k;
        out.n_leaves++;
        out.max_leaf = std::max<int64_t>(out.max_leaf, (int64_t)nd->tris.size());
        return;
    }
    const size_t pair = out.nodes.size();
    out.nodes.resize(pair + 2, KdNode{0, kLeafTag});
    out.boxes.resize(8 * (pair + 2), 0.f);
    KdNode k;
    union { float f; uint32_t u; } cv;
    cv.f = nd->split;
    k.a = cv.u;
    k.b = ((uint32_t)nd->axis << 30) | (uint32_t)pair;
    out.nodes[idx] = k;
    flatten(nd->left.get(), pair, out, depth + 1);
    flatten(nd->right.get(), pair + 1, out, depth + 1);
}

void emit_flat(const std::vector<BuildNode> &tree, int i, size_t idx, const int32_t *leaf_refs, KdTree &out, int depth) {
    const BuildNode &nd = tree[(size_t)i];
    const float b[8] = {nd.lo[0], nd.hi[0], nd.lo[1], nd.hi[1], nd.lo[2], nd.hi[2], 0.f, 0.f};
    std::copy(b, b + 8, out.boxes.begin() + 8 * idx);
    out.max_depth = std::max(out.max_depth, depth);
    if (nd.degenerate) out.degenerate_nodes++;
    if (nd.left < 0) {
        KdNode k;
        k.a = (uint32_t)out.refs.size();
        k.b = kLeafTag | (uint32_t)nd.leaf_count;
        out.refs.insert(out.refs.end(), leaf_refs + nd.leaf_begin, leaf_refs + nd.leaf_begin + nd.leaf_count);
        out.nodes[idx] = k;
        out.n_leaves++;
        out.max_leaf = std::max<int64_t>(out.max_leaf, (int64_t)nd.leaf_count);
        return;
    }
    const size_t pair = out.nodes.size();
    out.nodes.resize(pair + 2, KdNode{0, kLeafTag});
    out.boxes.resize(8 * (pair + 2), 0.f);
    KdNode k;
    union { float f; uint32_t u; } cv;
    cv.f = nd.split;
    k.a = cv.u;
    k.b = ((uint32_t)nd.axis << 30) | (uint32_t)pair;
    out.nodes[idx] = k;
    emit_flat(tree, nd.left, pair, leaf_refs, out, depth + 1);
    emit_flat(tree, nd.right, pair + 1, leaf_refs, out, depth + 1);
}

}  // namespace

void assemble_kd_tree(const std::vector<BuildNode> &tree, const int32_t *leaf_refs, KdTree &out) {
    out = KdTree();
    out.nodes.assign(2, KdNode{0, kLeafTag});  // root + one padding slot (an empty leaf nobody references)
    out.boxes.assign(16, 0.f);
    emit_flat(tree, 0, 0, leaf_refs, out, 0);
}

void build_kd_tree(const float *pos, int64_t n_tris, KdTree &out, int n_threads) {
    out = KdTree();
    std::unique_ptr<TNode> root(new TNode());
    // root box = AABB of all vertices, axis X (mesh.cu:138-170)
    for (int k = 0; k < 3; k++) root->lo[k] = root->hi[k] = pos[k];
    for (int64_t i = 0; i < n_tris * 3; i++)
        for (int k = 0; k < 3; k++) {
            const float c = pos[i * 3 + k];
            if (c < root->lo[k]) root->lo[k] = c;
            if (c > root->hi[k]) root->hi[k] = c;
        }
    root->axis = 0;
    std::vector<int32_t> all((size_t)n_tris);
    for (int64_t i = 0; i < n_tris; i++) all[(size_t)i] = (int32_t)i;
    Builder b;
    b.pos = pos;
    b.spare_threads.store(std::max(0, n_threads - 1));
    b.build(root.get(), std::move(all), 0);
    out.nodes.assign(2, KdNode{0, kLeafTag});  // root + one padding slot (an empty leaf nobody references)
    out.boxes.assign(16, 0.f);
    flatten(root.get(), 0, out, 0);
}

}  // namespace mipt
