// kd_build_gpu.hip -- the kd-tree of Mesh::getKdTree / Node::makeNodeTree (shapes/mesh.cu:32-155) built ON THE GPU,
// level by level, bit-identical to the host builder (kd_build.cpp) and therefore to the reference's tree:
// same root box, same split planes (Box::half recomputes lo + width * pct, mesh.cu:5-30), same membership rule
// ("at least one VERTEX inside the child's box" with the asymmetric epsilons of Box::isVertexInBox, mesh.cuh:131-146),
// same leaf rule (mesh.cu:33) and the same 20-step median walk (mesh.cu:72-108).
//
// The reference recurses depth-first with by-value triangle vectors; here one LEVEL of the tree is in flight at a time:
//   k_keys     one lane per (node, triangle) reference: the triangle's two membership keys for this node (smallest
//              admissible coordinate for the left child, largest for the right child -- see kd_build.cpp's header) and its
//              up-to-three split candidates (vertex coordinates strictly inside the node on the split axis, else +inf)
//   kd_sort.h  ONE device-wide radix sort of the level's candidates as 64-bit keys (node slot << 32 | order-preserving
//              bits of the coordinate): every node's candidates end up contiguous and ascending (mesh.cu:68 sorts them per
//              node) however the references are spread over the nodes -- a segmented sort gives a 4 M-key root segment to
//              one workgroup (measured with the library's: 89 % of the build).  The library's radix_sort_keys until round 3;
//              now the build's own LSD sort (8-bit digits, ballot match-any ranking)
//   k_search   one workgroup per node: the median walk -- each of its <= 21 counting passes is a block-wide reduction
//              over the node's keys -- then the final plane and the sizes of the two children
//   (host)     256 bytes per node come back; the host decides which children are leaves (<= 50 triangles or depth > 15)
//              and where every child's triangle list goes (next level's buffer, or its final place in the leaf lists)
//   k_split    one workgroup per node: stable two-way partition of the node's list into its children (a triangle
//              straddling the plane goes to both, mesh.cu:123-130); order inside a list stays the input order
// Nothing depends on a thread count or on timing: the tree is the same tree, node for node, as kd_build.cpp's (asserted
// by tests/test_gpu_kd_build.py on 1.3 M triangles).  The flat numbering (children adjacent, depth-first) is applied by
// the same host routine as for the host build (assemble_kd_tree).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "kd_build.h"
#include "kd_sort.h"

namespace mipt {
namespace {

constexpr int kBuildBlock = 256;

#define KD_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            err = std::string("GPU kd build: ") + #expr + ": " + hipGetErrorString(e_);   \
            return false;                                                                 \
        }                                                                                 \
    } while (0)

struct DevBuf {  // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// One node of the level being split (uploaded per level).
struct LevelNode {
    float lo[3], hi[3];
    int axis;
    int begin, end;  // its references in the level's list
    int pad;
};
// What k_search finds for it.
struct SplitResult {
    float split;
    int n_left, n_right;
    int degenerate;  // no split candidate / zero width: the node stays a leaf (kd_build.cpp does the same)
};
// Where k_split puts the two children's lists (destination arrays are chosen by the host per child).
struct SplitDest {
    int left_to_leaf, right_to_leaf;    // 1: the child is a leaf, its list goes to the final leaf-reference array
    int left_offset, right_offset;      // offset in that array / in the next level's list
    int left_slot, right_slot;          // next level's node slot (owner id) when the child is split further
};

// fp32 -> uint32 whose unsigned order is the float order (-0 just below +0: equal coordinates either way), and back
__device__ __forceinline__ uint32_t ord_of(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ __forceinline__ int block_sum(int v, int *lds) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    int s = 0;
    for (int w = 0; w < kBuildBlock / 64; w++) s += lds[w];
    return s;
}

// keys + split candidates of every reference of the level (kd_build.cpp:58-83, same fp32 operations)
__global__ __launch_bounds__(kBuildBlock) void k_keys(const float *__restrict__ pos, const LevelNode *__restrict__ nodes,
                                                       const int *__restrict__ tris, const int *__restrict__ owner, int n_refs,
                                                       float *__restrict__ key_l, float *__restrict__ key_r, unsigned long long *__restrict__ cand) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n_refs) return;
    const unsigned long long own = (unsigned long long)(unsigned int)owner[i] << 32;
    const LevelNode nd = nodes[owner[i]];
    const int a = nd.axis;
    const float blo = nd.lo[a], bhi = nd.hi[a];
    const float *p0 = pos + (size_t)tris[i] * 9;
    const float inf = __builtin_inff();
    float kl = inf, kr = -inf;
    for (int v = 0; v < 3; v++) {
        const float p[3] = {p0[3 * v], p0[3 * v + 1], p0[3 * v + 2]};
        const float c = p[a];
        cand[3 * (size_t)i + v] = own | ord_of((c > blo && c < bhi) ? c : inf);  // strictly inside (inBoundsOnAxis, mesh.cuh:180-192)
        bool lo_ok[3], hi_ok[3];  // Box::isVertexInBox against the parent's bounds, per axis (mesh.cuh:131-139)
        lo_ok[0] = nd.lo[0] <= p[0];
        lo_ok[1] = nd.lo[1] <= p[1] + kEps;
        lo_ok[2] = nd.lo[2] <= p[2] + kEps;
        hi_ok[0] = p[0] <= nd.hi[0] + kEps;
        hi_ok[1] = p[1] <= nd.hi[1] + kEps;
        hi_ok[2] = p[2] <= nd.hi[2] + kEps;
        const int u = (a + 1) % 3, w = (a + 2) % 3;
        const bool other = lo_ok[u] && hi_ok[u] && lo_ok[w] && hi_ok[w];
        if (other && lo_ok[a]) kl = fminf(kl, c);
        if (other && hi_ok[a]) kr = fmaxf(kr, a == 0 ? c : c + kEps);
    }
    key_l[i] = kl;
    key_r[i] = kr;
}

// the median walk of one node (mesh.cu:72-114); cand_sorted: the level's keys; the node's 3 * n candidates ascending, +inf last
__global__ __launch_bounds__(kBuildBlock) void k_search(const LevelNode *__restrict__ nodes, const float *__restrict__ key_l,
                                                         const float *__restrict__ key_r, const unsigned long long *__restrict__ cand_sorted,
                                                         SplitResult *__restrict__ out) {
    __shared__ int lds[kBuildBlock / 64];
    __shared__ int s_nin;
    const LevelNode nd = nodes[blockIdx.x];
    const int a = nd.axis, n = nd.end - nd.begin;
    const float blo = nd.lo[a], bhi = nd.hi[a];
    const unsigned long long *cs = cand_sorted + 3 * (size_t)nd.begin;
    if (threadIdx.x == 0) {  // n_in = the finite prefix of the sorted candidates
        int lo = 0, hi = 3 * n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)cs[mid] < ord_of(__builtin_inff())) lo = mid + 1; else hi = mid;
        }
        s_nin = lo;
    }
    __syncthreads();
    const int n_in = s_nin;
    const float width = bhi - blo;
    if (n_in == 0 || !(width > 0.0f)) {
        if (threadIdx.x == 0) out[blockIdx.x] = SplitResult{0.0f, 0, 0, 1};
        return;
    }
    const float *kl = key_l + nd.begin, *kr = key_r + nd.begin;
    auto plane_of = [&](int idx) { return blo + width * ((float_of((uint32_t)cs[idx]) - blo) / width); };  // mesh.cu:76-78 + Box::half :19,23,27
    auto count = [&](float s, int &nl, int &nr) {
        const float s_eps = s + kEps;
        int cl = 0, cr = 0;
        for (int k = threadIdx.x; k < n; k += kBuildBlock) {
            cl += kl[k] <= s_eps;
            cr += s <= kr[k];
        }
        nl = block_sum(cl, lds);
        nr = block_sum(cr, lds);
    };
    int idx = n_in / 2;
    for (int i = 2; i <= kSplitSearchDepth + 1; i++) {  // every value below is block-uniform
        int nl, nr;
        count(plane_of(idx), nl, nr);
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
    const float s = plane_of(idx);
    int nl, nr;
    count(s, nl, nr);
    if (threadIdx.x == 0) out[blockIdx.x] = SplitResult{s, nl, nr, 0};
}

// stable two-way partition of one node's list (mesh.cu:121-130)
__global__ __launch_bounds__(kBuildBlock) void k_split(const LevelNode *__restrict__ nodes, const SplitResult *__restrict__ res,
                                                        const SplitDest *__restrict__ dest, const int *__restrict__ tris,
                                                        const float *__restrict__ key_l, const float *__restrict__ key_r,
                                                        int *__restrict__ next_tris, int *__restrict__ next_owner,
                                                        int *__restrict__ leaf_refs) {
    __shared__ int wsum[2][kBuildBlock / 64];
    const LevelNode nd = nodes[blockIdx.x];
    const SplitResult r = res[blockIdx.x];
    if (r.degenerate) return;
    const SplitDest d = dest[blockIdx.x];
    const float s = r.split, s_eps = s + kEps;
    const int n = nd.end - nd.begin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int base_l = 0, base_r = 0;  // block-uniform running counts
    for (int c0 = 0; c0 < n; c0 += kBuildBlock) {
        const int k = c0 + threadIdx.x;
        const bool in = k < n;
        const int tri = in ? tris[nd.begin + k] : 0;
        const bool fl = in && key_l[nd.begin + k] <= s_eps;
        const bool fr = in && s <= key_r[nd.begin + k];
        const unsigned long long ml = __ballot(fl), mr = __ballot(fr);
        const unsigned long long below = (1ull << lane) - 1ull;
        __syncthreads();
        if (lane == 0) { wsum[0][wave] = __popcll(ml); wsum[1][wave] = __popcll(mr); }
        __syncthreads();
        int off_l = base_l, off_r = base_r, tot_l = 0, tot_r = 0;
        for (int w = 0; w < kBuildBlock / 64; w++) {
            if (w < wave) { off_l += wsum[0][w]; off_r += wsum[1][w]; }
            tot_l += wsum[0][w];
            tot_r += wsum[1][w];
        }
        if (fl) {
            const int at = d.left_offset + off_l + __popcll(ml & below);
            if (d.left_to_leaf) leaf_refs[at] = tri;
            else { next_tris[at] = tri; next_owner[at] = d.left_slot; }
        }
        if (fr) {
            const int at = d.right_offset + off_r + __popcll(mr & below);
            if (d.right_to_leaf) leaf_refs[at] = tri;
            else { next_tris[at] = tri; next_owner[at] = d.right_slot; }
        }
        base_l += tot_l;
        base_r += tot_r;
    }
}

// the list of a node that stays a leaf although it was meant to be split (degenerate): copied out as it is
__global__ __launch_bounds__(kBuildBlock) void k_copy_list(const int *__restrict__ src, int n, int *__restrict__ dst) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(kBuildBlock) void k_iota(int *__restrict__ tris, int *__restrict__ owner, int n) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i < n) { tris[i] = i; owner[i] = 0; }
}

}  // namespace

bool build_kd_tree_gpu(int device, const float *pos, int64_t n_tris, KdTree &out, std::string &err, const void *d_pos_in) {
    if (n_tris <= 0 || n_tris > (int64_t)(1u << 28)) { err = "GPU kd build: triangle count out of range"; return false; }
    KD_TRY(hipSetDevice(device));
    hipStream_t st;
    KD_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};

    // root box = AABB of all vertices, axis X (mesh.cu:138-151); a plain min / max, done where the vertices are
    BuildNode root;
    for (int k = 0; k < 3; k++) root.lo[k] = root.hi[k] = pos[k];
    for (int64_t i = 0; i < n_tris * 3; i++)
        for (int k = 0; k < 3; k++) {
            const float c = pos[i * 3 + k];
            if (c < root.lo[k]) root.lo[k] = c;
            if (c > root.hi[k]) root.hi[k] = c;
        }
    root.axis = 0;
    std::vector<BuildNode> tree;  // breadth-first as it is discovered; children linked by index
    tree.push_back(root);
    if (n_tris <= kMaxLeafTris) {  // a single leaf: nothing to split
        tree[0].leaf_begin = 0;
        tree[0].leaf_count = (int)n_tris;
        std::vector<int32_t> ids((size_t)n_tris);
        for (int64_t i = 0; i < n_tris; i++) ids[(size_t)i] = (int32_t)i;
        assemble_kd_tree(tree, ids.data(), out);
        return true;
    }

    DevBuf d_pos, d_tris[2], d_owner[2], d_kl, d_kr, d_cand, d_cand_sorted, d_nodes, d_res, d_dest, d_leaf, d_temp;
    const float *pos_dev = static_cast<const float *>(d_pos_in);
    if (!pos_dev) {
        KD_TRY(d_pos.reserve((size_t)n_tris * 36));
        KD_TRY(hipMemcpyAsync(d_pos.p, pos, (size_t)n_tris * 36, hipMemcpyHostToDevice, st));
        pos_dev = d_pos.as<float>();
    }
    // a reference list never holds a triangle twice, a level's lists together can: sized as they grow (DevBuf::reserve)
    size_t leaf_cap = (size_t)n_tris * 2 + 1024;
    KD_TRY(d_leaf.reserve(leaf_cap * 4));
    int64_t leaf_used = 0;

    struct Cur { int tree_index; int begin, end; };
    std::vector<Cur> level{{0, 0, (int)n_tris}};
    int64_t level_refs = n_tris;
    int cur = 0;
    KD_TRY(d_tris[0].reserve((size_t)n_tris * 4));
    KD_TRY(d_owner[0].reserve((size_t)n_tris * 4));
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((n_tris + kBuildBlock - 1) / kBuildBlock)), dim3(kBuildBlock), 0, st,
                       d_tris[0].as<int>(), d_owner[0].as<int>(), (int)n_tris);

    std::vector<LevelNode> h_nodes;
    std::vector<SplitResult> h_res;
    std::vector<SplitDest> h_dest;
    for (int depth = 0; !level.empty(); depth++) {
        const int n_nodes = (int)level.size();
        const int n_refs = (int)level_refs;
        h_nodes.resize((size_t)n_nodes);
        for (int j = 0; j < n_nodes; j++) {
            const BuildNode &b = tree[(size_t)level[(size_t)j].tree_index];
            LevelNode &ln = h_nodes[(size_t)j];
            for (int k = 0; k < 3; k++) { ln.lo[k] = b.lo[k]; ln.hi[k] = b.hi[k]; }
            ln.axis = b.axis;
            ln.begin = level[(size_t)j].begin;
            ln.end = level[(size_t)j].end;
            ln.pad = 0;
        }
        KD_TRY(d_nodes.reserve((size_t)n_nodes * sizeof(LevelNode)));
        KD_TRY(d_res.reserve((size_t)n_nodes * sizeof(SplitResult)));
        KD_TRY(d_dest.reserve((size_t)n_nodes * sizeof(SplitDest)));
        KD_TRY(d_kl.reserve((size_t)n_refs * 4));
        KD_TRY(d_kr.reserve((size_t)n_refs * 4));
        KD_TRY(d_cand.reserve((size_t)n_refs * 24));
        KD_TRY(d_cand_sorted.reserve((size_t)n_refs * 24));
        KD_TRY(hipMemcpyAsync(d_nodes.p, h_nodes.data(), (size_t)n_nodes * sizeof(LevelNode), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_keys, dim3((unsigned)((n_refs + kBuildBlock - 1) / kBuildBlock)), dim3(kBuildBlock), 0, st,
                           pos_dev, d_nodes.as<LevelNode>(), d_tris[cur].as<int>(), d_owner[cur].as<int>(), n_refs,
                           d_kl.as<float>(), d_kr.as<float>(), d_cand.as<unsigned long long>());
        const unsigned long long *cand_sorted;
        {
            int owner_bits = 1;
            while ((1 << owner_bits) < n_nodes) owner_bits++;
            KD_TRY(d_temp.reserve(kd_sort_scratch_bytes(3 * (int64_t)n_refs)));
            cand_sorted = launch_kd_sort(st, d_cand.as<unsigned long long>(), d_cand_sorted.as<unsigned long long>(), 3 * (int64_t)n_refs,
                                         32 + owner_bits, d_temp.p);
            KD_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_search, dim3((unsigned)n_nodes), dim3(kBuildBlock), 0, st, d_nodes.as<LevelNode>(), d_kl.as<float>(),
                           d_kr.as<float>(), cand_sorted, d_res.as<SplitResult>());
        h_res.resize((size_t)n_nodes);
        KD_TRY(hipMemcpyAsync(h_res.data(), d_res.p, (size_t)n_nodes * sizeof(SplitResult), hipMemcpyDeviceToHost, st));
        KD_TRY(hipStreamSynchronize(st));

        // the host's part of a level: children, leaves, destinations
        std::vector<Cur> next;
        int64_t next_refs = 0;
        h_dest.assign((size_t)n_nodes, SplitDest{0, 0, 0, 0, 0, 0});
        for (int j = 0; j < n_nodes; j++) {
            const SplitResult r = h_res[(size_t)j];
            const int ti = level[(size_t)j].tree_index;
            const int n = level[(size_t)j].end - level[(size_t)j].begin;
            if (r.degenerate) {  // stays a leaf; its list is copied out below
                tree[(size_t)ti].degenerate = true;
                tree[(size_t)ti].leaf_begin = (int)leaf_used;
                tree[(size_t)ti].leaf_count = n;
                leaf_used += n;
                continue;
            }
            const int a = tree[(size_t)ti].axis;
            tree[(size_t)ti].split = r.split;
            BuildNode l = tree[(size_t)ti], rr = tree[(size_t)ti];
            l.left = l.right = rr.left = rr.right = -1;
            l.hi[a] = r.split;
            rr.lo[a] = r.split;
            l.axis = rr.axis = (a + 1) % 3;  // mesh.cu:132-133
            const int li = (int)tree.size();
            tree[(size_t)ti].left = li;
            tree[(size_t)ti].right = li + 1;
            SplitDest &d = h_dest[(size_t)j];
            const int cnt[2] = {r.n_left, r.n_right};
            BuildNode *child[2] = {&l, &rr};
            for (int c = 0; c < 2; c++) {
                const bool leaf = cnt[c] <= kMaxLeafTris || depth + 1 > kMaxKdHeight;  // mesh.cu:33 for the child
                int off, slot = 0;
                if (leaf) {
                    off = (int)leaf_used;
                    child[c]->leaf_begin = off;
                    child[c]->leaf_count = cnt[c];
                    leaf_used += cnt[c];
                } else {
                    off = (int)next_refs;
                    slot = (int)next.size();
                    next.push_back(Cur{li + c, off, off + cnt[c]});
                    next_refs += cnt[c];
                }
                if (c == 0) { d.left_to_leaf = leaf; d.left_offset = off; d.left_slot = slot; }
                else { d.right_to_leaf = leaf; d.right_offset = off; d.right_slot = slot; }
            }
            tree.push_back(l);
            tree.push_back(rr);
        }
        if (leaf_used > (int64_t)INT32_MAX || next_refs > (int64_t)INT32_MAX / 3) { err = "GPU kd build: reference lists too long"; return false; }
        if ((size_t)leaf_used > leaf_cap) {  // grow the leaf array, keeping what is already in it
            DevBuf bigger;
            const size_t new_cap = (size_t)leaf_used * 2;
            KD_TRY(bigger.reserve(new_cap * 4));
            KD_TRY(hipMemcpyAsync(bigger.p, d_leaf.p, leaf_cap * 4, hipMemcpyDeviceToDevice, st));
            KD_TRY(hipStreamSynchronize(st));
            std::swap(bigger.p, d_leaf.p);
            std::swap(bigger.cap, d_leaf.cap);
            leaf_cap = new_cap;
        }
        KD_TRY(d_tris[cur ^ 1].reserve((size_t)std::max<int64_t>(next_refs, 1) * 4));
        KD_TRY(d_owner[cur ^ 1].reserve((size_t)std::max<int64_t>(next_refs, 1) * 4));
        KD_TRY(hipMemcpyAsync(d_dest.p, h_dest.data(), (size_t)n_nodes * sizeof(SplitDest), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_split, dim3((unsigned)n_nodes), dim3(kBuildBlock), 0, st, d_nodes.as<LevelNode>(), d_res.as<SplitResult>(),
                           d_dest.as<SplitDest>(), d_tris[cur].as<int>(), d_kl.as<float>(), d_kr.as<float>(), d_tris[cur ^ 1].as<int>(),
                           d_owner[cur ^ 1].as<int>(), d_leaf.as<int>());
        for (int j = 0; j < n_nodes; j++)
            if (h_res[(size_t)j].degenerate) {
                const int n = level[(size_t)j].end - level[(size_t)j].begin;
                hipLaunchKernelGGL(k_copy_list, dim3((unsigned)((n + kBuildBlock - 1) / kBuildBlock)), dim3(kBuildBlock), 0, st,
                                   d_tris[cur].as<int>() + level[(size_t)j].begin, n,
                                   d_leaf.as<int>() + tree[(size_t)level[(size_t)j].tree_index].leaf_begin);
            }
        KD_TRY(hipGetLastError());
        level.swap(next);
        level_refs = next_refs;
        cur ^= 1;
    }
    std::vector<int32_t> leaf_refs((size_t)leaf_used);
    KD_TRY(hipMemcpyAsync(leaf_refs.data(), d_leaf.p, (size_t)leaf_used * 4, hipMemcpyDeviceToHost, st));
    KD_TRY(hipStreamSynchronize(st));
    assemble_kd_tree(tree, leaf_refs.data(), out);
    return true;
}

}  // namespace mipt
