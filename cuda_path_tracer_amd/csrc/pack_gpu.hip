// pack_gpu.hip -- the scene packer ON THE GPU: what replaces copyToDevice (copyToDevice/copyToDevice.h:8-55; the per-leaf
// uploads of shapes/mesh.cu:35-36 and the node array of mesh.cu:173-201) for scenes with large meshes.  The host packer
// (scene.cpp) derives, per kd node and per leaf reference, the records the closest-hit kernel reads -- global nodes with
// their recheck flags, cull boxes (leaf: AABB of the leaf's triangles within the leaf box; interior: union of the
// children's), the 64-byte step records, leaf-ordered planes / vertices / candidate spheres, padded triangle arrays -- on
// host threads and uploads the finished arena (349 MB for 1.31 M triangles): after the kd build went to the GPU that was the
// larger part of a commit.  Every one of those records is independent per node or per reference, so here the INPUTS go up
// (positions, normals, the flat tree: 0.3 x the arena) and the records are made where they are used, in the device arena.
//
// The arena must come out BYTE-IDENTICAL to the host packer's (tests/test_gpu_kd_build.py): every kernel below evaluates the
// host's expressions in the host's order -- fp32 without contraction for the plane records, IEEE double (add, mul, div, sqrt:
// all correctly rounded on gfx950) for the candidate spheres, std::min / std::max written out as the comparisons they are
// (they are not symmetric in their arguments for +-0), nextafterf for the outward rounding of the step records.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pack_gpu.h"

namespace mipt {
namespace {

constexpr int kPackBlock = 256;

#define PK_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            err = std::string("GPU pack: ") + #expr + ": " + hipGetErrorString(e_);       \
            return false;                                                                 \
        }                                                                                 \
    } while (0)

// std::min(a, b) / std::max(a, b) as the library defines them (first argument on ties and for +-0)
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }

struct MeshArgs {
    const float *pos, *nrm;      // n_tris * 9 each
    const KdNode *nodes;         // flat tree, mesh-local indices
    const float *boxes;          // 8 floats per node
    const int32_t *refs;         // mesh-local triangle ids
    const uint8_t *depth;        // per node
    int32_t n_nodes, n_refs, n_tris;
    uint32_t node_base, ref_base, tri_base;
    int force_recheck;
};
struct ArenaPtrs {
    KdNode *nodes;
    float *boxes, *tight, *tight_pairs;
    int32_t *refs;
    float *leaf_recs, *leaf_spheres, *tri_pos, *tri_nrm;
};

// nodes (global indices + recheck flag), boxes, the leaves' cull boxes, and which leaf owns every reference
__global__ __launch_bounds__(kPackBlock) void k_pack_nodes(MeshArgs m, ArenaPtrs a, int32_t *__restrict__ ref_leaf) {
    const int i = blockIdx.x * kPackBlock + threadIdx.x;
    if (i >= m.n_nodes) return;
    KdNode k = m.nodes[i];
    const bool leaf = (k.b >> 30) == 3u;
    const float *nb = m.boxes + 8 * (size_t)i;
    KdNode g = k;
    if (leaf) g.a += m.ref_base;
    else {
        g.b = (k.b & ~kIdxMask) | ((k.b & kIdxMask) + m.node_base);
        const uint32_t ax = k.b >> 30;
        const float sv = __uint_as_float(k.a);
        if (m.force_recheck || !(nb[2 * ax] <= sv && sv <= nb[2 * ax + 1])) g.b |= kRecheckBit;
    }
    a.nodes[m.node_base + i] = g;
    float *bo = a.boxes + 8 * ((size_t)m.node_base + i), *tb = a.tight + 8 * ((size_t)m.node_base + i);
    for (int c = 0; c < 8; c++) { bo[c] = nb[c]; tb[c] = nb[c]; }
    if (!leaf) return;
    const int cnt = (int)(k.b & kIdxMask);
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int r = 0; r < cnt; r++) {
        ref_leaf[k.a + r] = i;
        const float *v = m.pos + (size_t)m.refs[k.a + r] * 9;
        for (int c = 0; c < 9; c++) {
            lo[c % 3] = std_min(lo[c % 3], v[c]);
            hi[c % 3] = std_max(hi[c % 3], v[c]);
        }
    }
    for (int c = 0; c < 3; c++) {  // empty leaf: lo = +inf, hi = -inf -> every ray is culled
        tb[2 * c] = std_max(nb[2 * c], lo[c]);
        tb[2 * c + 1] = std_min(nb[2 * c + 1], hi[c]);
    }
}

// interior cull boxes, one tree level per launch, deepest first (children carry larger indices than their parent)
__global__ __launch_bounds__(kPackBlock) void k_pack_interior(MeshArgs m, ArenaPtrs a, int level) {
    const int i = blockIdx.x * kPackBlock + threadIdx.x;
    if (i >= m.n_nodes || m.depth[i] != level) return;
    const KdNode k = m.nodes[i];
    if ((k.b >> 30) == 3u) return;
    const float *l = a.tight + 8 * ((size_t)m.node_base + (k.b & kIdxMask)), *r = l + 8;
    float *tb = a.tight + 8 * ((size_t)m.node_base + i);
    for (int c = 0; c < 3; c++) {
        tb[2 * c] = std_min(l[2 * c], r[2 * c]);
        tb[2 * c + 1] = std_max(l[2 * c + 1], r[2 * c + 1]);
    }
}

__device__ __forceinline__ void grow_box(float lo, float hi, float &glo, float &ghi) {
    const float inf = __builtin_inff();
    if (!(lo <= hi)) { glo = inf; ghi = -inf; return; }
    const float mg = 2.44140625e-4f * std_max(std_max(fabsf(lo), fabsf(hi)), 1.0f);
    glo = nextafterf(lo - mg, -inf);
    ghi = nextafterf(hi + mg, inf);
}

// the 64-byte step record of every sibling pair: both cull boxes grown and interleaved, + the children's global nodes
__global__ __launch_bounds__(kPackBlock) void k_pack_pairs(MeshArgs m, ArenaPtrs a) {
    const int i = blockIdx.x * kPackBlock + threadIdx.x;
    if (i >= m.n_nodes) return;
    const KdNode k = m.nodes[i];
    if ((k.b >> 30) == 3u) return;
    const size_t lc = (size_t)(k.b & kIdxMask);
    const float *l = a.tight + 8 * ((size_t)m.node_base + lc), *r = l + 8;
    float *rec = a.tight_pairs + 16 * (((size_t)m.node_base + lc) >> 1);
    const float inf = __builtin_inff();
    const bool l_empty = !(l[0] <= l[1]) || !(l[2] <= l[3]) || !(l[4] <= l[5]);
    const bool r_empty = !(r[0] <= r[1]) || !(r[2] <= r[3]) || !(r[4] <= r[5]);
    for (int c = 0; c < 3; c++) {
        grow_box(l_empty ? inf : l[2 * c], l_empty ? -inf : l[2 * c + 1], rec[4 * c + 0], rec[4 * c + 2]);
        grow_box(r_empty ? inf : r[2 * c], r_empty ? -inf : r[2 * c + 1], rec[4 * c + 1], rec[4 * c + 3]);
    }
    const KdNode nl = a.nodes[m.node_base + lc], nr = a.nodes[m.node_base + lc + 1];
    rec[12] = __uint_as_float(nl.a); rec[13] = __uint_as_float(nl.b);
    rec[14] = __uint_as_float(nr.a); rec[15] = __uint_as_float(nr.b);
}

// per leaf reference: global triangle id, plane record (mesh.cuh:40-42,52), vertices, candidate sphere
__global__ __launch_bounds__(kPackBlock) void k_pack_refs(MeshArgs m, ArenaPtrs a, const int32_t *__restrict__ ref_leaf) {
    const int ri = blockIdx.x * kPackBlock + threadIdx.x;
    if (ri >= m.n_refs) return;
    const size_t gi = (size_t)m.ref_base + ri;
    const int32_t tri = m.refs[ri];
    a.refs[gi] = tri + (int32_t)m.tri_base;
    const float *v = m.pos + (size_t)tri * 9;
    {
        const f3 pa = mk3(v[0], v[1], v[2]), pb = mk3(v[3], v[4], v[5]), pc = mk3(v[6], v[7], v[8]);
        const f3 N = cross3(pb - pa, pc - pa);
        float *rec = a.leaf_recs + 16 * gi;  // 64 B: plane, a, b, c
        rec[0] = N.x; rec[1] = N.y; rec[2] = N.z;
        rec[3] = -dot3(N, pa);
        for (int q = 0; q < 3; q++) {
            for (int c = 0; c < 3; c++) rec[4 + 4 * q + c] = v[3 * q + c];
            rec[4 + 4 * q + 3] = 0.0f;
        }
        rec[7] = __int_as_float(tri + (int32_t)m.tri_base);  // a.w: the global triangle id (scene.cpp)
    }
    // candidate sphere (scene.cpp: the smaller of the triangle's smallest enclosing sphere and the sphere around
    // AABB(triangle) within the leaf's cull box, grown by its margin) -- the host's double arithmetic, operation for operation
    const float *tb = a.tight + 8 * ((size_t)m.node_base + ref_leaf[ri]);
    double P[3][3], mag = 1.0;
    for (int q = 0; q < 3; q++)
        for (int c = 0; c < 3; c++) { P[q][c] = v[3 * q + c]; mag = std_max(mag, fabs(P[q][c])); }
    auto dist2 = [](const double *x, const double *y) {
        return (x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) + (x[2] - y[2]) * (x[2] - y[2]);
    };
    double cen[3], rad2;
    {
        int e0 = 0, e1 = 1, eo = 2;
        double best = dist2(P[0], P[1]);
        if (dist2(P[1], P[2]) > best) { best = dist2(P[1], P[2]); e0 = 1; e1 = 2; eo = 0; }
        if (dist2(P[2], P[0]) > best) { best = dist2(P[2], P[0]); e0 = 2; e1 = 0; eo = 1; }
        for (int c = 0; c < 3; c++) cen[c] = 0.5 * (P[e0][c] + P[e1][c]);
        rad2 = 0.25 * best;
        if (dist2(cen, P[eo]) > rad2) {
            const double *pa = P[0], *pb = P[1], *pc = P[2];
            double ab[3], ac[3], n[3];
            for (int q = 0; q < 3; q++) { ab[q] = pb[q] - pa[q]; ac[q] = pc[q] - pa[q]; }
            n[0] = ab[1] * ac[2] - ab[2] * ac[1]; n[1] = ab[2] * ac[0] - ab[0] * ac[2]; n[2] = ab[0] * ac[1] - ab[1] * ac[0];
            const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
            const double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], ac2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
            if (n2 > 0.0) {
                double u[3], w[3];
                u[0] = n[1] * ab[2] - n[2] * ab[1]; u[1] = n[2] * ab[0] - n[0] * ab[2]; u[2] = n[0] * ab[1] - n[1] * ab[0];
                w[0] = ac[1] * n[2] - ac[2] * n[1]; w[1] = ac[2] * n[0] - ac[0] * n[2]; w[2] = ac[0] * n[1] - ac[1] * n[0];
                double cc[3];
                for (int q = 0; q < 3; q++) cc[q] = pa[q] + (ac2 * u[q] + ab2 * w[q]) / (2.0 * n2);
                const double r2 = std_max(dist2(cc, pa), std_max(dist2(cc, pb), dist2(cc, pc)));
                if (r2 <= 2.0 * rad2) {
                    for (int q = 0; q < 3; q++) cen[q] = cc[q];
                    rad2 = r2;
                } else {
                    rad2 = std_max(rad2, dist2(cen, P[eo]));
                }
            } else {
                rad2 = std_max(rad2, dist2(cen, P[eo]));
            }
        }
    }
    {
        double lo[3], hi[3];
        bool empty = false;
        for (int c = 0; c < 3; c++) {
            lo[c] = std_max((double)tb[2 * c], std_min(P[0][c], std_min(P[1][c], P[2][c])));
            hi[c] = std_min((double)tb[2 * c + 1], std_max(P[0][c], std_max(P[1][c], P[2][c])));
            mag = std_max(mag, std_max(fabs((double)tb[2 * c]), fabs((double)tb[2 * c + 1])));
            if (!(lo[c] <= hi[c])) empty = true;
        }
        if (!empty) {
            double bc[3], br2 = 0.0;
            for (int c = 0; c < 3; c++) { bc[c] = 0.5 * (lo[c] + hi[c]); br2 += 0.25 * (hi[c] - lo[c]) * (hi[c] - lo[c]); }
            if (br2 < rad2) { rad2 = br2; for (int c = 0; c < 3; c++) cen[c] = bc[c]; }
        }
    }
    if (!isfinite(mag)) mag = 1.0;
    const double R = sqrt(rad2) * (1.0 + 1.0 / 1024.0) + 4e-5 + mag / 65536.0;
    float R2 = (float)(R * R);
    R2 = nextafterf(R2, __builtin_inff());
    if (!(R2 >= 0.0f) || !isfinite(R2) || !isfinite(cen[0]) || !isfinite(cen[1]) || !isfinite(cen[2])) {
        cen[0] = cen[1] = cen[2] = 0.0;
        R2 = __builtin_inff();
    }
    float *rec = a.leaf_spheres + 8 * (gi >> 1) + (gi & 1);
    rec[0] = (float)cen[0]; rec[2] = (float)cen[1]; rec[4] = (float)cen[2];
    rec[6] = R2;
}

// triangle arrays: 3 x float4 per triangle (w = 0), positions and normals
__global__ __launch_bounds__(kPackBlock) void k_pack_tris(MeshArgs m, ArenaPtrs a) {
    const int64_t i = (int64_t)blockIdx.x * kPackBlock + threadIdx.x;  // vertex index
    if (i >= (int64_t)m.n_tris * 3) return;
    float *tp = a.tri_pos + 4 * ((size_t)m.tri_base * 3 + (size_t)i), *tn = a.tri_nrm + 4 * ((size_t)m.tri_base * 3 + (size_t)i);
    for (int c = 0; c < 3; c++) { tp[c] = m.pos[3 * i + c]; tn[c] = m.nrm[3 * i + c]; }
    tp[3] = 0.0f; tn[3] = 0.0f;
}

// slots no reference owns (padding) must never pass: R^2 = -inf
__global__ __launch_bounds__(kPackBlock) void k_init_spheres(float *ls, size_t n_pairs) {
    const size_t i = (size_t)blockIdx.x * kPackBlock + threadIdx.x;
    if (i < n_pairs) ls[8 * i + 6] = ls[8 * i + 7] = -__builtin_inff();
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace

void *gpu_upload(int device, const void *src, size_t bytes, std::string &err) {
    void *p = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&p, bytes > 0 ? bytes : 16);
    if (e == hipSuccess && bytes > 0) e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        err = std::string("GPU upload: ") + hipGetErrorString(e);
        if (p) (void)hipFree(p);
        return nullptr;
    }
    return p;
}

void gpu_free(int device, void *p) {
    if (!p) return;
    (void)hipSetDevice(device);
    (void)hipFree(p);
}

bool gpu_arena_write(int device, void *d_arena, size_t off, const void *src, size_t bytes, std::string &err) {
    if (bytes == 0) return true;
    PK_TRY(hipSetDevice(device));
    PK_TRY(hipMemcpy(static_cast<uint8_t *>(d_arena) + off, src, bytes, hipMemcpyHostToDevice));
    return true;
}

bool gpu_arena_read(int device, const void *d_arena, size_t bytes, uint8_t *dst, std::string &err) {
    if (bytes == 0) return true;
    PK_TRY(hipSetDevice(device));
    PK_TRY(hipMemcpy(dst, d_arena, bytes, hipMemcpyDeviceToHost));
    return true;
}

bool gpu_pack_scene(int device, const std::vector<GpuPackMesh> &meshes, const ArenaLayout &L, int64_t total_refs, bool force_recheck,
                    void **d_arena_out, float *root_tight, std::string &err) {
    *d_arena_out = nullptr;
    PK_TRY(hipSetDevice(device));
    Buf arena;
    PK_TRY(hipMalloc(&arena.p, L.total));
    uint8_t *A = static_cast<uint8_t *>(arena.p);
    PK_TRY(hipMemsetAsync(A, 0, L.meshes, nullptr));  // the big sections; the small ones behind them are written by the host
    ArenaPtrs ap;
    ap.nodes = reinterpret_cast<KdNode *>(A + L.nodes);
    ap.boxes = reinterpret_cast<float *>(A + L.boxes);
    ap.tight = reinterpret_cast<float *>(A + L.tight);
    ap.tight_pairs = reinterpret_cast<float *>(A + L.tight_pairs);
    ap.refs = reinterpret_cast<int32_t *>(A + L.refs);
    ap.leaf_recs = reinterpret_cast<float *>(A + L.leaf_recs);
    ap.leaf_spheres = reinterpret_cast<float *>(A + L.leaf_spheres);
    ap.tri_pos = reinterpret_cast<float *>(A + L.tri_pos);
    ap.tri_nrm = reinterpret_cast<float *>(A + L.tri_nrm);
    {
        const size_t n_pairs = (size_t)total_refs / 2 + 4;
        hipLaunchKernelGGL(k_init_spheres, dim3((unsigned)((n_pairs + kPackBlock - 1) / kPackBlock)), dim3(kPackBlock), 0, nullptr, ap.leaf_spheres, n_pairs);
    }
    auto grid = [](int64_t n) { return dim3((unsigned)((n + kPackBlock - 1) / kPackBlock)); };
    for (size_t mi = 0; mi < meshes.size(); mi++) {
        const GpuPackMesh &gm = meshes[mi];
        const KdTree &t = *gm.tree;
        const int64_t n_nodes = (int64_t)t.nodes.size(), n_refs = (int64_t)t.refs.size();
        if ((gm.node_base & 1) != 0) { err = "internal: odd node base"; return false; }
        // depth of every node: children come after their parent in the flat order, so one forward sweep does it
        std::vector<uint8_t> depth((size_t)n_nodes, 0);
        int max_depth = 0;
        for (int64_t i = 0; i < n_nodes; i++) {
            const KdNode k = t.nodes[(size_t)i];
            if ((k.b >> 30) == 3u) continue;
            const size_t lc = (size_t)(k.b & kIdxMask);
            depth[lc] = depth[lc + 1] = (uint8_t)(depth[(size_t)i] + 1);
            if (depth[lc] > max_depth) max_depth = depth[lc];
        }
        Buf d_pos, d_nrm, d_nodes, d_boxes, d_refs, d_depth, d_ref_leaf;
        const float *pos_dev = static_cast<const float *>(gm.d_pos);
        if (!pos_dev) {
            PK_TRY(hipMalloc(&d_pos.p, (size_t)gm.n_tris * 36));
            PK_TRY(hipMemcpyAsync(d_pos.p, gm.pos, (size_t)gm.n_tris * 36, hipMemcpyHostToDevice, nullptr));
            pos_dev = static_cast<const float *>(d_pos.p);
        }
        PK_TRY(hipMalloc(&d_nrm.p, (size_t)gm.n_tris * 36));
        PK_TRY(hipMalloc(&d_nodes.p, (size_t)n_nodes * sizeof(KdNode)));
        PK_TRY(hipMalloc(&d_boxes.p, (size_t)n_nodes * 32));
        PK_TRY(hipMalloc(&d_refs.p, (size_t)std::max<int64_t>(n_refs, 1) * 4));
        PK_TRY(hipMalloc(&d_depth.p, (size_t)n_nodes));
        PK_TRY(hipMalloc(&d_ref_leaf.p, (size_t)std::max<int64_t>(n_refs, 1) * 4));
        PK_TRY(hipMemcpyAsync(d_nrm.p, gm.nrm, (size_t)gm.n_tris * 36, hipMemcpyHostToDevice, nullptr));
        PK_TRY(hipMemcpyAsync(d_nodes.p, t.nodes.data(), (size_t)n_nodes * sizeof(KdNode), hipMemcpyHostToDevice, nullptr));
        PK_TRY(hipMemcpyAsync(d_boxes.p, t.boxes.data(), (size_t)n_nodes * 32, hipMemcpyHostToDevice, nullptr));
        if (n_refs > 0) PK_TRY(hipMemcpyAsync(d_refs.p, t.refs.data(), (size_t)n_refs * 4, hipMemcpyHostToDevice, nullptr));
        PK_TRY(hipMemcpyAsync(d_depth.p, depth.data(), (size_t)n_nodes, hipMemcpyHostToDevice, nullptr));
        MeshArgs ma;
        ma.pos = pos_dev; ma.nrm = static_cast<const float *>(d_nrm.p);
        ma.nodes = static_cast<const KdNode *>(d_nodes.p); ma.boxes = static_cast<const float *>(d_boxes.p);
        ma.refs = static_cast<const int32_t *>(d_refs.p); ma.depth = static_cast<const uint8_t *>(d_depth.p);
        ma.n_nodes = (int32_t)n_nodes; ma.n_refs = (int32_t)n_refs; ma.n_tris = (int32_t)gm.n_tris;
        ma.node_base = (uint32_t)gm.node_base; ma.ref_base = (uint32_t)gm.ref_base; ma.tri_base = (uint32_t)gm.tri_base;
        ma.force_recheck = force_recheck ? 1 : 0;
        int32_t *ref_leaf = static_cast<int32_t *>(d_ref_leaf.p);
        hipLaunchKernelGGL(k_pack_nodes, grid(n_nodes), dim3(kPackBlock), 0, nullptr, ma, ap, ref_leaf);
        for (int level = max_depth - 1; level >= 0; level--)
            hipLaunchKernelGGL(k_pack_interior, grid(n_nodes), dim3(kPackBlock), 0, nullptr, ma, ap, level);
        hipLaunchKernelGGL(k_pack_pairs, grid(n_nodes), dim3(kPackBlock), 0, nullptr, ma, ap);
        if (n_refs > 0) hipLaunchKernelGGL(k_pack_refs, grid(n_refs), dim3(kPackBlock), 0, nullptr, ma, ap, ref_leaf);
        hipLaunchKernelGGL(k_pack_tris, grid((int64_t)gm.n_tris * 3), dim3(kPackBlock), 0, nullptr, ma, ap);
        PK_TRY(hipGetLastError());
        PK_TRY(hipMemcpyAsync(root_tight + 8 * mi, ap.tight + 8 * (size_t)gm.node_base, 32, hipMemcpyDeviceToHost, nullptr));
        PK_TRY(hipStreamSynchronize(nullptr));  // the inputs (and `depth`) go away with this iteration
    }
    PK_TRY(hipDeviceSynchronize());
    *d_arena_out = arena.p;
    arena.p = nullptr;
    return true;
}

}  // namespace mipt
