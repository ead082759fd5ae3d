// scene_types.h -- flat SoA scene layout shared by the host packer and the HIP kernels.
//
// This layout replaces the reference's pointer-linked device objects (80-byte fat kd nodes with
// per-leaf 72-byte triangle copies, mesh.cuh:204-220, mesh.cu:35-36) and the ~2 allocations per
// leaf made through copyToDevice (copyToDevice.h:8-55).  The whole scene lives in ONE device arena:
//
//   nodes       uint2[n_nodes]      8-byte kd node.  interior: x = bits(split plane), y = axis<<30 | recheck<<29 | left child
//                                   (right child = left + 1, pair 16-byte aligned); leaf: x = first ref, y = 3<<30 | count.
//                                   Read for mesh roots and stack pops; the node steps read the copy in tight_pairs
//   boxes       float4[2*n_nodes]   the node's box (minX,maxX,minY,maxY) (minZ,maxZ,-,-): root and stack pops only
//   tight       float4[2*n_nodes]   cull box, same layout: leaf = AABB(leaf triangles) within the leaf box, interior = union
//                                   of the children's; root cull / whole-mesh skip, reference-walk counting kernel
//   tight_pairs float4[4*(n/2+1)]   THE STEP RECORD, 64 B per sibling pair (left child index >> 1): the two cull boxes grown by
//                                   their margin, component-interleaved (loL,loR,hiL,hiR) per axis (48 B) + the two children's
//                                   8-byte nodes (16 B): one half cache line per node step
//   refs        int32[n_refs]       leaf triangle index lists (global triangle ids, leaf order = input order); read once per
//                                   leaf scan that found a hit (reference position -> triangle id)
//   leaf_spheres SpherePair[n_refs/2+4] 32 B per PAIR of references: conservative candidate spheres, streamed by pass 1 of the
//                                   leaf scan (packed fp32)
//   leaf_recs   float4[4*n_refs]    64 B per reference, leaf order: plane record (N, -N.a) + the three vertex positions again
//                                   (a.w = the global triangle id as int bits, b.w = c.w = 0)
//                                   (no reference -> triangle hop): pass 2, marked candidates -- one line per candidate
//   tri_pos     float4[3*n_tris]    vertex positions a,b,c (w unused), triangle order: shading (surface_at)
//   tri_nrm     float4[3*n_tris]    vertex normals, fetched once for the winning triangle
//   meshes / mesh_order / spheres / planes / lights / materials: small tables; wave-uniform reads go through ldk4() below
//   env         6 cube faces, raw RGBE (4 B/texel) or fp32 RGB
#pragma once
#include <stdint.h>
#include "pt_math.h"

namespace mipt {

constexpr int kMaxLeafTris = 50;       // constants.cuh:12
constexpr int kMaxKdHeight = 15;       // constants.cuh:13
constexpr int kAllowedDiff = 20;       // constants.cuh:14
constexpr int kSplitSearchDepth = 20;  // constants.cuh:15
constexpr int kStackDepth = 18;        // >= 16 interior levels + root (reference: int[15], Q9)
constexpr int kMaxPathDepth = 16;      // accepted cam_depth / light_depth

constexpr uint32_t kLeafTag = 3u << 30;
constexpr uint32_t kIdxMask = (1u << 29) - 1;
// Interior nodes only (set by the packer): the split plane lies outside [lo, hi] of the node on its axis (possible
// by one rounding of Box::half's lo + width*pct), so a child is not a subset of its parent and the incremental
// plane-validity update of descend_fast() must be replaced by a full re-evaluation for the children.
constexpr uint32_t kRecheckBit = 1u << 29;

struct KdNode {
    uint32_t a, b;
};

struct alignas(16) KdNodePair {  // the two children of an interior node: left, right
    KdNode l, r;
};

struct DMesh {
    int32_t root;      // global node index of the mesh's root
    int32_t tri_base;  // first global triangle id
    int32_t n_tris;
    int32_t material;
};

struct alignas(16) f4 {
    float x, y, z, w;
};

// Candidate spheres of leaf references 2j and 2j+1, component-interleaved so that one packed fp32 instruction
// (v_pk_*_f32) works on both candidates: (x0,x1) (y0,y1) (z0,z1) (R0^2,R1^2).
typedef float v2f __attribute__((ext_vector_type(2)));
struct alignas(32) SpherePair {
    v2f x, y, z, r2;
};

// Loads from the scene's small read-only tables (planes, spheres, lights, mesh roots and their cull boxes).  The arena is
// never written while a kernel runs, so on the device these go through the CONSTANT address space: a wave-uniform
// address becomes a scalar load (s_load_dwordx4/x8 into SGPRs through the scalar data cache, no VGPRs, hoistable out of
// loops) instead of a vector load per lane of the same 16 bytes followed by a wait on the vector-memory counter; a
// divergent address still compiles to an ordinary vector load.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float __attribute__((address_space(4))) *kfloat_p;
typedef const int32_t __attribute__((address_space(4))) *kint_p;
PT_HD f4 ldk4(const f4 *base, size_t idx) {
    kfloat_p p = (kfloat_p)(uintptr_t)(base + idx);
    f4 r;
    r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    return r;
}
PT_HD int32_t ldk_i32(const int32_t *p) { return *(kint_p)(uintptr_t)p; }
#else
PT_HD f4 ldk4(const f4 *base, size_t idx) { return base[idx]; }
PT_HD int32_t ldk_i32(const int32_t *p) { return *p; }
#endif

// Device view of the packed scene; passed to kernels by value (kernarg -> scalar loads).
struct DScene {
    const KdNode *nodes;
    const f4 *boxes;
    const f4 *tight;        // per node, same layout as boxes: leaf cull box (triangle AABB within the leaf box)
    const f4 *tight_pairs;  // per sibling pair (left child index >> 1), 64 B: the two cull boxes component-interleaved (48 B) + the two children's nodes (16 B)
                            // (loxL,loxR,hixL,hixR) (loyL,loyR,hiyL,hiyR) (lozL,lozR,hizL,hizR) (pad) for packed fp32
    const int32_t *refs;
    const f4 *leaf_recs;    // per leaf reference, in leaf order, ONE 64-byte record (half a cache line): (N.xyz, -dot(N,a)) of the
                            // triangle, then its vertex positions a, b, c -- a marked candidate's plane test and edge tests read
                            // one line (round 2 kept planes and vertices in two arrays: two to three lines per candidate)
    const SpherePair *leaf_spheres;  // per PAIR of leaf references (global reference index >> 1): spheres that contain
                                     // every hit point the leaf loop could accept for that reference
    const f4 *tri_pos;
    const f4 *tri_nrm;
    const DMesh *meshes;
    const int32_t *mesh_order;  // [3][n_meshes] ROOT node of the meshes sorted by the centre of their root cull box along
                                // x, y, z: the extend kernel opens meshes near-to-far along the ray's dominant axis
    const f4 *spheres;    // 2 per sphere: (center, radius) (mat bits, -, -, -)
    const f4 *planes;     // 2 per plane:  (point, mat bits) (normal, -)
    const f4 *lights;     // 2 per light:  (location, -) (power, -)
    const f4 *materials;  // 2 per material: (diffuse, -) (mirror, -)
    const uint32_t *env_rgbe;  // 6*w*h texels R|G<<8|B<<16|E<<24, or null
    const float *env_f32;      // 6*w*h*3 floats, or null
    const float *env_marg;     // the map as a light (use_envmap == 2): CDF over the 6*h rows, + 1 entry: 1.0f = usable
    const float *env_cond;     // CDF inside each row, [6*h][w]
    int32_t n_meshes, n_spheres, n_planes, n_lights, n_materials, total_tris;
    int32_t env_w, env_h;
    float cam_eye[3], cam_lookat[3], cam_right[3], cam_up[3];
};

}  // namespace mipt
