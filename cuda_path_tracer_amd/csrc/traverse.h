// traverse.h -- kd-tree closest hit over the packed scene (device code; plain C++ so that a
// test-side harness can also instantiate it on the host against the host copy of the arena).
//
// Semantics are those of Scene::firstIntersect (kernel.cu:145-160), Mesh::intersect
// (mesh.cuh:253-305), Box::intersect / isVertexInBox (mesh.cuh:131-170), Triangle::intersect
// (mesh.cuh:36-90), Plane::intersect (plane.cuh:18-30) and Sphere::intersect (sphere.cuh:19-37),
// reproduced bit for bit in fp32, including the quirks catalogued in SURVEY.md 8.1 (Q3-Q8).
//
// What is different is the machine mapping (layout: scene_types.h):
//  * 8-byte nodes.  A child's box differs from its parent's in ONE plane (Box::half, mesh.cu:5-30),
//    so the current node's six bounds and the ray parameter of each of its six slab planes are
//    carried in registers; descending costs ONE division (the split plane) instead of two 80-byte
//    nodes and twelve divisions, and one 64-byte step record (tight_pairs: both children's cull boxes
//    and 8-byte nodes).  The values are the same fp32 values the reference computes, because they come
//    from the same expressions.
//  * The per-node 32-byte box is re-read only for the root and after a stack pop.
//  * Subtrees whose cull box (tight / tight_pairs) the ray misses are neither entered nor stacked.
//  * The traversal stack lives in LDS (one column per lane, bank-conflict free).
//  * In a leaf, pass 1 streams conservative candidate spheres (leaf_spheres, packed fp32); only the
//    marked candidates get the reference's evaluation from leaf-ordered plane and vertex records
//    (leaf_recs: one 64-byte record each); the vertex normals are never touched here.
#pragma once
#include "scene_types.h"

namespace mipt {

struct TravCounters {
    unsigned int nodes, pops, refs, tris;  // units of the reference's traversal
    unsigned int culled;                   // candidates of those that the leaf cull skipped
};

// What the production walk reads (k_extend<kCountWork>, MIPT_FLAG_WORK_COUNTERS): see mipt_stats.work_* in mipt.h.
struct WorkCounters {
    unsigned int steps, pops, sphere_pairs, planes, verts, leaves;
};

// Developer build only (make PHASE_STATS=1): per-phase SIMD-lane use of the persistent extend kernel.
// exec[i] counts wave-level executions of phase i, lanes[i] the lanes active in them.
#ifdef MIPT_PHASE_STATS
constexpr int kPhases = 12;
struct PhaseStats { unsigned int exec[kPhases], lanes[kPhases]; };
#define PSTAT(ps, i)                                                                                     \
    do {                                                                                                 \
        (ps).lanes[i]++;                                                                                 \
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(1)) - 1) (ps).exec[i]++;             \
    } while (0)
#define PSTAT_ARG , PhaseStats &ps
#define PSTAT_PASS , ps
#define PSTAT_LOCAL PhaseStats ps = {};
#else
#define PSTAT(ps, i)
#define PSTAT_ARG
#define PSTAT_PASS
#define PSTAT_LOCAL
#endif

PT_HD float as_f32(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}
PT_HD uint32_t as_u32(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

// One-instruction min / max for operands that are never NaN (fminf/fmaxf add a canonicalising v_max x,x per operand
// the compiler cannot prove quiet; a compare + select is two instructions).
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD float fmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_HD float fmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_HD float fmin3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// v_rcp_f32, 1 ulp.  (__frcp_rn and 1.0f / x are a 13-instruction IEEE division under
// -fhip-fp32-correctly-rounded-divide-sqrt; only the conservative cull tests may use this one.)
PT_HD float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
#else
PT_HD float fmin_raw(float a, float b) { return a < b ? a : b; }
PT_HD float fmax_raw(float a, float b) { return a > b ? a : b; }
PT_HD float fmin3_raw(float a, float b, float c) { return fmin_raw(fmin_raw(a, b), c); }
PT_HD float rcp_approx(float x) { return 1.0f / x; }
#endif

// Box::isVertexInBox (mesh.cuh:131-139): min-X has no epsilon, every other bound has +eps slack (Q5)
PT_HD bool point_in_box(f3 p, float lox, float hix, float loy, float hiy, float loz, float hiz) {
    return (lox <= p.x) & (p.x <= hix + kEps) & (loy <= p.y + kEps) & (p.y <= hiy + kEps) & (loz <= p.z + kEps) &
           (p.z <= hiz + kEps);
}

struct SlabBox {
    float lo[3], hi[3];    // bounds
    float tlo[3], thi[3];  // ray parameter of each bounding plane: (plane - o)/d   (mesh.cuh:151-160)
};

// Box::intersect (mesh.cuh:148-170): smallest t > 0 among the six planes whose point lies in the box, else -1
PT_HD float slab_hit(const SlabBox &b, f3 o, f3 d) {
    const float inf = __builtin_inff();
    float best = inf;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t0 = b.tlo[k], t1 = b.thi[k];
        const f3 p0 = ray_at(o, d, t0), p1 = ray_at(o, d, t1);
        const bool ok0 = (t0 > 0.0f) & point_in_box(p0, b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2]);
        const bool ok1 = (t1 > 0.0f) & point_in_box(p1, b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2]);
        const float c0 = ok0 ? t0 : inf, c1 = ok1 ? t1 : inf;
        best = (c0 < best) ? c0 : best;
        best = (c1 < best) ? c1 : best;
    }
    return (best < inf) ? best : -1.0f;
}

PT_HD void load_slab(const DScene &sc, int node, f3 o, f3 d, SlabBox &b) {
    const f4 b0 = sc.boxes[2 * (size_t)node], b1 = sc.boxes[2 * (size_t)node + 1];
    b.lo[0] = b0.x; b.hi[0] = b0.y; b.lo[1] = b0.z; b.hi[1] = b0.w; b.lo[2] = b1.x; b.hi[2] = b1.y;
    b.tlo[0] = (b.lo[0] - o.x) / d.x;
    b.thi[0] = (b.hi[0] - o.x) / d.x;
    b.tlo[1] = (b.lo[1] - o.y) / d.y;
    b.thi[1] = (b.hi[1] - o.y) / d.y;
    b.tlo[2] = (b.lo[2] - o.z) / d.z;
    b.thi[2] = (b.hi[2] - o.z) / d.z;
}

// Triangle::intersect (mesh.cuh:36-90) split in two.  Part 1 needs only the plane record (N, dd) that the
// packer precomputed with the reference's own expressions (mesh.cuh:40-42,52): plane distance, the
// acceptance conditions of Mesh::intersect (mesh.cuh:293-296: t > 0, t < best, point inside the leaf box, Q8)
// and the parallel test (Q3).  Returns the candidate's t, or -1 when it cannot be accepted.
PT_HD float plane_candidate(f4 pl, f3 o, f3 d, const SlabBox &box, float best) {
    const f3 N = mk3(pl.x, pl.y, pl.z);
    const float nd = dot3(N, d);
    if (fabsf(nd) < kEps) return -1.0f;  // Q3
    const float t = -(dot3(N, o) + pl.w) / nd;
    if (!(t > 0.0f)) return -1.0f;                 // t<0 -> raw negative, t==0 -> fails "hit.t > 0"
    if (!(best < 0.0f || t < best)) return -1.0f;  // cannot replace the current best
    const f3 P = ray_at(o, d, t);
    if (!point_in_box(P, box.lo[0], box.hi[0], box.lo[1], box.hi[1], box.lo[2], box.hi[2])) return -1.0f;
    return t;
}
// Part 2: the three inside-outside edge tests (mesh.cuh:61-88) for a surviving candidate.
PT_HD bool inside_triangle(f4 pl, f3 a, f3 b, f3 c, f3 o, f3 d, float t) {
    const f3 N = mk3(pl.x, pl.y, pl.z);
    const f3 P = ray_at(o, d, t);
    f3 C = cross3(b - a, P - a);
    if (dot3(N, C) < 0.0f) return false;
    C = cross3(c - b, P - b);
    if (dot3(N, C) < 0.0f) return false;
    C = cross3(a - c, P - c);
    if (dot3(N, C) < 0.0f) return false;
    return true;
}

PT_HD bool leaf_culled(const DScene &sc, int node, f3 o, f3 d);
PT_HD bool box_culled(f4 b0, f4 b1, f3 o, f3 d);

// Inner loop of Mesh::intersect (mesh.cuh:265-288): walk from `node` (whose 8 bytes are in `nd`) down towards a
// leaf, pushing the far child when both children are hit (Q7).  `cur` (bounds + plane parameters of `node`)
// is updated in place.  Both children sit in one aligned 16-byte pair; its load is issued first and is only
// consumed after the step's slab arithmetic, which hides most of the node-fetch latency.
// Stops at a leaf or after `max_steps` steps (then `nd` is still interior).  With CULL, returns true as soon as the
// ray misses the cull box of the subtree it just stepped into (see leaf_culled): nothing can be found below.
template <class Stack, bool COUNT, bool CULL = false>
PT_HD bool descend_to_leaf(const DScene &sc, f3 o, f3 d, SlabBox &cur, int &node, KdNode &nd, int &sp, Stack &stk,
                           TravCounters &ctr, int max_steps PSTAT_ARG) {
    for (int step = 0; (nd.b >> 30) != 3u && step < max_steps; step++) {
        PSTAT(ps, 9);
        const uint32_t ax = nd.b >> 30;
        const int left = (int)(nd.b & kIdxMask), right = left + 1;
        const KdNodePair pair = *reinterpret_cast<const KdNodePair *>(sc.nodes + left);
        const float s = as_f32(nd.a);
        const float oa = ax == 0 ? o.x : (ax == 1 ? o.y : o.z);
        const float da = ax == 0 ? d.x : (ax == 1 ? d.y : d.z);
        const float ts = (s - oa) / da;
        SlabBox rb = cur, lb = cur;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool on = ax == (uint32_t)k;
            rb.lo[k] = on ? s : cur.lo[k];
            rb.tlo[k] = on ? ts : cur.tlo[k];
            lb.hi[k] = on ? s : cur.hi[k];
            lb.thi[k] = on ? ts : cur.thi[k];
        }
        const float rt = slab_hit(rb, o, d);  // mesh.cuh:266-269
        const float lt = slab_hit(lb, o, d);
        const bool rf = rt < lt;              // Q7: a miss (-1) sorts first
        const float small_t = rf ? rt : lt, big_t = rf ? lt : rt;
        const bool go_right = (small_t > 0.0f) ? rf : !rf;  // mesh.cuh:281
        if ((small_t > 0.0f) & (big_t > 0.0f)) {            // mesh.cuh:285
            if (sp < kStackDepth) stk.push(sp, rf ? left : right);
            sp++;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cur.lo[k] = go_right ? rb.lo[k] : cur.lo[k];
            cur.tlo[k] = go_right ? rb.tlo[k] : cur.tlo[k];
            cur.hi[k] = go_right ? cur.hi[k] : lb.hi[k];
            cur.thi[k] = go_right ? cur.thi[k] : lb.thi[k];
        }
        node = go_right ? right : left;
        nd.a = go_right ? pair.r.a : pair.l.a;
        nd.b = go_right ? pair.r.b : pair.l.b;
        if (COUNT) ctr.nodes++;
        if (CULL && leaf_culled(sc, node, o, d)) return true;
    }
    return false;
}

// ---- the same walk, incrementally (the kernel's fast path; results identical to descend_to_leaf) ----------------
//
// Box::intersect is "smallest t > 0 among the six plane points that pass isVertexInBox".  Call plane j of a box
// VALID when t_j > 0 and its point o + d t_j passes the six bound tests.  A child box is its parent with ONE bound
// replaced by the split plane, so for the five planes it inherits (same t_j, same point):
//       valid_child(j) = valid_parent(j)  AND  (the point passes the new bound),
// because the replaced parent bound is implied by the tighter child bound (fp add is monotonic; the packer flags the
// rare node whose split lies outside [lo, hi] with kRecheckBit and the children of such a node are re-evaluated in
// full).  The new bound only involves the point's coordinate ALONG THE SPLIT AXIS: one fused multiply-add per plane instead of
// the whole point, one compare instead of six.  Only the split plane's own point is evaluated in full.
// A plane that is not valid never becomes valid further down, so its t is never needed again: the slab state of the
// fast path stores +inf in its place ("masked" t), and Box::intersect is a plain minimum.  (For a masked entry the
// coordinate below is +-inf or NaN; whatever the compare says, the entry stays +inf.)
PT_HD void mask_slab(SlabBox &b, f3 o, f3 d) {
    const float inf = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t0 = b.tlo[k], t1 = b.thi[k];
        const f3 p0 = ray_at(o, d, t0), p1 = ray_at(o, d, t1);
        const bool ok0 = (t0 > 0.0f) & point_in_box(p0, b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2]);
        const bool ok1 = (t1 > 0.0f) & point_in_box(p1, b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2]);
        b.tlo[k] = ok0 ? t0 : inf;
        b.thi[k] = ok1 ? t1 : inf;
    }
}
// Box::intersect of a masked slab (== slab_hit of the same box)
PT_HD float masked_min(const float tlo[3], const float thi[3]) {
    const float best = fmin_raw(fmin3_raw(tlo[0], thi[0], tlo[1]), fmin3_raw(thi[1], tlo[2], thi[2]));
    return (best < __builtin_inff()) ? best : -1.0f;
}

// box_culled() for the two children of a node at once: their cull boxes come component-interleaved (scene_types.h),
// so turning them into ray parameters is packed fp32 work on (left, right) pairs.  The packer already grew these boxes
// by 2^-12 x max(|lo|, |hi|, 1).  That margin covers the 1e-5 box slack, the rounding of the edge tests and the
// rounding of the reference's P = o + d t (<= ~2.5 ulp of max(|o|, |P|) per axis) as long as |o| is within ~2000 box
// scales; for origins farther out the hit parameter t is itself ~|o|/|d| on every axis the ray is not parallel to,
// and the rounding of P moves the slab parameters by ~1e-7 t, which the 2^-12 (|tf| + 1) widening of the interval
// test absorbs.  (Parallel axes: P_k = o_k exactly, the box margin alone decides.)
PT_HD void pair_culled(f4 bx, f4 by, f4 bz, f3 o, f3 d, bool &cull_l, bool &cull_r, float &near_l, float &near_r) {
    const f4 bb[3] = {bx, by, bz};
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float tn_l = 0.0f, tf_l = __builtin_inff(), tn_r = 0.0f, tf_r = __builtin_inff();
    bool out_l = false, out_r = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const v2f l = {bb[k].x, bb[k].y}, h = {bb[k].z, bb[k].w};
        if (fabsf(dd[k]) < 1e-30f) {
            out_l |= (oo[k] < l.x) | (oo[k] > h.x);
            out_r |= (oo[k] < l.y) | (oo[k] > h.y);
        } else {
            // a 1-ulp reciprocal is plenty: the test is conservative by 2^-12.  (Three v_rcp_f32 per step look expensive and are
            //  not: the reciprocals kept per ray in registers cost C2 4 % / the incoherent batch 6 % in round 3 as in round 2
            //  -- 9 spilled dwords instead of 3 -- and kept in LDS rows next to the stack 5 % / 10 %.)
            const float inv = rcp_approx(dd[k]);
            const v2f t0 = (l - oo[k]) * inv, t1 = (h - oo[k]) * inv;
            tn_l = fmax_raw(tn_l, fmin_raw(t0.x, t1.x));
            tf_l = fmin_raw(tf_l, fmax_raw(t0.x, t1.x));
            tn_r = fmax_raw(tn_r, fmin_raw(t0.y, t1.y));
            tf_r = fmin_raw(tf_r, fmax_raw(t0.y, t1.y));
        }
    }
    // an empty box is stored as (+inf, -inf): its own half turns into NaN / inf and is decided by the last term
    cull_l = out_l | (tn_l > tf_l + 2.44140625e-4f * (fabsf(tf_l) + 1.0f)) | !(bx.x <= bx.z);
    cull_r = out_r | (tn_r > tf_r + 2.44140625e-4f * (fabsf(tf_r) + 1.0f)) | !(bx.y <= bx.w);
    near_l = tn_l - 2.44140625e-4f * (tn_l + 1.0f);  // lower bounds (tn >= 0) of the parameter of any accepted hit inside
    near_r = tn_r - 2.44140625e-4f * (tn_r + 1.0f);
}

// Walks down from `node` for at most max_steps steps; `cur` is a MASKED slab.  Every node this function stands on
// has passed its cull test (see box_culled), and so has every node it pushes: a subtree the ray cannot hit anything
// in is neither entered nor stacked -- visiting it is a no-op in the reference, whatever the order (Q7).  Returns
// true when nothing below `node` is left to visit (dead end: the caller continues with the stack), false when it
// stopped on `node` (a leaf, or an interior node if the steps ran out).
//
// Distance pruning (scenes with several meshes; t_prune = +inf switches it off): a child whose cull box the ray enters
// strictly beyond t_prune = the closest answer of the objects evaluated so far can only hold hits that
// Scene::firstIntersect would not use, so it is treated like a culled child -- but unlike a culled child it may hold
// the accepted hit at which the reference ENDS this mesh's search (Q8).  `taint_bit` of `flags` records that such a
// child was dropped; if the search then still comes up with a hit that would be used, the caller repeats the mesh without
// pruning (rare: it takes a leaf order that is not front to back).
template <class Stack, bool WORK = false>
PT_HD bool descend_fast(const DScene &sc, f3 o, f3 d, SlabBox &cur, int &node, KdNode &nd, int &sp, Stack &stk,
                        int max_steps, bool prune, float t_prune, uint32_t &flags, uint32_t taint_bit, WorkCounters &wc PSTAT_ARG) {
    const float inf = __builtin_inff();
    for (int step = 0; (nd.b >> 30) != 3u && step < max_steps; step++) {
        PSTAT(ps, 9);
        if (WORK) wc.steps++;
        const uint32_t ax = nd.b >> 30;
        const bool recheck = (nd.b & kRecheckBit) != 0u;
        const int left = (int)(nd.b & kIdxMask), right = left + 1;
        // ONE 64-byte record per sibling pair (pair index = left / 2): the two cull boxes and the children's 8-byte nodes.
        // (A step used to read the nodes from their own array: two 128-byte lines per step for an incoherent ray.)
        const f4 *tb = sc.tight_pairs + 2 * (size_t)left;
        const f4 tbx = tb[0], tby = tb[1], tbz = tb[2], tbn = tb[3];
        KdNodePair pair;
        pair.l.a = as_u32(tbn.x); pair.l.b = as_u32(tbn.y); pair.r.a = as_u32(tbn.z); pair.r.b = as_u32(tbn.w);
        // (Touching the step records of BOTH children one step ahead -- a dummy load each, consumed only at the end of the
        //  walk -- was measured: coherent closest hit 3690 -> 3390 Mrays/s, incoherent 2240 -> 1885, 1080p frame -8 %.)
        bool cull_l, cull_r;
        float near_l, near_r;
        pair_culled(tbx, tby, tbz, o, d, cull_l, cull_r, near_l, near_r);
        if (prune) {  // wave-uniform
            const bool far_l = near_l > t_prune, far_r = near_r > t_prune;
            flags |= ((far_l & !cull_l) | (far_r & !cull_r)) ? taint_bit : 0u;
            cull_l |= far_l;
            cull_r |= far_r;
        }
        if (cull_l & cull_r) return true;

        const float s = as_f32(nd.a);
        const float oa = ax == 0 ? o.x : (ax == 1 ? o.y : o.z);
        const float da = ax == 0 ? d.x : (ax == 1 ? d.y : d.z);
        const float ts = (s - oa) / da;
        const float s_eps = s + kEps;
        const float e_ax = ax == 0 ? 0.0f : kEps;  // Q5: min-X has no slack
        // inherited planes: the new bound on the point's split-axis coordinate
        float l_lo[3], l_hi[3], r_lo[3], r_hi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float p0 = fma_at(oa, da, cur.tlo[k]), p1 = fma_at(oa, da, cur.thi[k]);
            l_lo[k] = (p0 <= s_eps) ? cur.tlo[k] : inf;
            l_hi[k] = (p1 <= s_eps) ? cur.thi[k] : inf;
            r_lo[k] = (s <= p0 + e_ax) ? cur.tlo[k] : inf;
            r_hi[k] = (s <= p1 + e_ax) ? cur.thi[k] : inf;
        }
        // the split plane's own point against the parent's bounds, then the one bound each child replaces
        const f3 P = ray_at(o, d, ts);
        const float Pc[3] = {P.x, P.y, P.z};
        // (logical operators on values that are all computed already: the compiler keeps them as lane masks and combines
        //  them with scalar instructions; the bitwise forms went through 0/1 integers in vector registers)
        const bool in_lo0 = cur.lo[0] <= Pc[0], in_lo1 = cur.lo[1] <= Pc[1] + kEps, in_lo2 = cur.lo[2] <= Pc[2] + kEps;
        const bool in_hi0 = Pc[0] <= cur.hi[0] + kEps, in_hi1 = Pc[1] <= cur.hi[1] + kEps, in_hi2 = Pc[2] <= cur.hi[2] + kEps;
        const bool on0 = ax == 0u, on1 = ax == 1u, on2 = ax == 2u;
        const float Pa = on0 ? Pc[0] : (on1 ? Pc[1] : Pc[2]);
        const bool lo_all = in_lo0 && in_lo1 && in_lo2, hi_all = in_hi0 && in_hi1 && in_hi2;
        const bool hi_off = (in_hi0 || on0) && (in_hi1 || on1) && (in_hi2 || on2);  // left box: hi[ax] is the split plane
        const bool lo_off = (in_lo0 || on0) && (in_lo1 || on1) && (in_lo2 || on2);  // right box: lo[ax] is the split plane
        const bool split_l = (ts > 0.0f) && (Pa <= s_eps) && lo_all && hi_off;
        const bool split_r = (ts > 0.0f) && (s <= Pa + e_ax) && hi_all && lo_off;
        const float ts_l = split_l ? ts : inf, ts_r = split_r ? ts : inf;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool on = ax == (uint32_t)k;
            l_hi[k] = on ? ts_l : l_hi[k];
            r_lo[k] = on ? ts_r : r_lo[k];
        }
        // Box::intersect of the two children (mesh.cuh:266-269): the minimum of each masked slab, a miss (-1) when nothing is
        // valid.  Every valid entry is a finite t > 0, so "hit" is "minimum < inf" and the reference's rules -- rt < lt with
        // a miss sorting first (Q7), enter the nearer hit child (mesh.cuh:281), stack the other when both are hit
        // (mesh.cuh:285) -- are logic on three compares (kept as lane masks: scalar instructions, not vector selects):
        //   rf = rt < lt            = l_hit && (!r_hit || rmin < lmin)
        //   go_right = small_t > 0 ? rf : !rf   = both ? rmin < lmin : !l_hit
        const float lmin = fmin_raw(fmin3_raw(l_lo[0], l_hi[0], l_lo[1]), fmin3_raw(l_hi[1], l_lo[2], l_hi[2]));
        const float rmin = fmin_raw(fmin3_raw(r_lo[0], r_hi[0], r_lo[1]), fmin3_raw(r_hi[1], r_lo[2], r_hi[2]));
        const bool l_hit = lmin < inf, r_hit = rmin < inf, r_nearer = rmin < lmin;
        const bool both = l_hit && r_hit;
        bool go_right = (both && r_nearer) || !l_hit;
        const bool first_dead = (go_right && cull_r) || (!go_right && cull_l);
        const bool other_dead = (go_right && cull_l) || (!go_right && cull_r);
        if (first_dead) {
            if (!both || other_dead) return true;  // nothing to find in the child the reference enters, nothing stacked
            go_right = !go_right;                  // the fruitless visit is skipped: continue with the stacked child
        } else if (both && !other_dead) {
            if (sp < kStackDepth) stk.push(sp, go_right ? left : right);
            sp++;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool on = ax == (uint32_t)k;
            cur.lo[k] = (go_right & on) ? s : cur.lo[k];
            cur.hi[k] = (!go_right & on) ? s : cur.hi[k];
            cur.tlo[k] = go_right ? r_lo[k] : l_lo[k];
            cur.thi[k] = go_right ? r_hi[k] : l_hi[k];
        }
        node = go_right ? right : left;
        nd.a = go_right ? pair.r.a : pair.l.a;
        nd.b = go_right ? pair.r.b : pair.l.b;
        if (recheck) {  // rare: rebuild the child's slab from its stored box
            load_slab(sc, node, o, d, cur);
            mask_slab(cur, o, d);
        }
    }
    return false;
}

// Leaf loop of Mesh::intersect (mesh.cuh:290-299): leaf_culled() first, then leaf_scan().
//
// Leaf cull (no counterpart in the reference; it changes no result): an accepted hit point lies on a triangle of
// the leaf and inside the leaf box, hence inside sc.tight[node] = AABB(leaf triangles) ∩ leaf box.  The ray is
// tested against that box grown by a margin of 2^-12 x (largest coordinate magnitude in play) -- thousands of ulps,
// far beyond the rounding of P = o + d*t, of the fp32 edge tests and of the 1e-5 box slack -- with the standard
// interval test.  A ray that misses the grown box cannot produce an accepted candidate, so the loop is skipped.
PT_HD bool box_culled(f4 b0, f4 b1, f3 o, f3 d) {
    const float lo[3] = {b0.x, b0.z, b1.x}, hi[3] = {b0.y, b0.w, b1.y};
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float tn = 0.0f, tf = __builtin_inff();
    bool out = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mg = 2.44140625e-4f * fmaxf(fmaxf(fabsf(lo[k]), fabsf(hi[k])), fmaxf(fabsf(oo[k]), 1.0f));
        const float l = lo[k] - mg, h = hi[k] + mg;
        if (fabsf(dd[k]) < 1e-30f) {
            out |= (oo[k] < l) | (oo[k] > h);
        } else {
            const float inv = rcp_approx(dd[k]);  // a 1-ulp reciprocal is plenty: the test is conservative by 2^-12
            const float t0 = (l - oo[k]) * inv, t1 = (h - oo[k]) * inv;
            tn = fmaxf(tn, fminf(t0, t1));
            tf = fminf(tf, fmaxf(t0, t1));
        }
    }
    // widen the interval itself by a relative margin as well (rounding of the divisions)
    return out | (tn > tf + 2.44140625e-4f * (fabsf(tf) + 1.0f)) | !(lo[0] <= hi[0]);
}
PT_HD bool leaf_culled(const DScene &sc, int node, f3 o, f3 d) {
    return box_culled(sc.tight[2 * (size_t)node], sc.tight[2 * (size_t)node + 1], o, d);
}
// Whole-mesh skip with a distance bound (no counterpart in the reference; it changes no result): besides missing the
// root cull box, a ray also cannot be given a USED hit by a mesh whose root cull box it enters strictly beyond t_max =
// the closest answer so far (earlier meshes, spheres, planes) -- every accepted hit of that mesh lies inside the box,
// hence at t > t_max, and Scene::firstIntersect (kernel.cu:147-158) keeps the earlier object on ties.  Only a WHOLE
// mesh may be dropped this way: inside a mesh the first leaf with an accepted hit ends the search (Q8), so pruning
// subtrees by distance could expose a later, nearer leaf the reference never reaches.
PT_HD bool mesh_skippable(const DScene &sc, int root, f3 o, f3 d, float t_max) {
    const f4 b0 = sc.tight[2 * (size_t)root], b1 = sc.tight[2 * (size_t)root + 1];
    const float lo[3] = {b0.x, b0.z, b1.x}, hi[3] = {b0.y, b0.w, b1.y};
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float tn = 0.0f, tf = __builtin_inff();
    bool out = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mg = 2.44140625e-4f * fmaxf(fmaxf(fabsf(lo[k]), fabsf(hi[k])), fmaxf(fabsf(oo[k]), 1.0f));
        const float l = lo[k] - mg, h = hi[k] + mg;
        if (fabsf(dd[k]) < 1e-30f) {
            out |= (oo[k] < l) | (oo[k] > h);
        } else {
            const float inv = rcp_approx(dd[k]);
            const float t0 = (l - oo[k]) * inv, t1 = (h - oo[k]) * inv;
            tn = fmaxf(tn, fminf(t0, t1));
            tf = fminf(tf, fmaxf(t0, t1));
        }
    }
    const bool beyond = tn - 2.44140625e-4f * (tn + 1.0f) > t_max;  // tn >= 0; same relative widening as the interval test
    return out | beyond | (tn > tf + 2.44140625e-4f * (fabsf(tf) + 1.0f)) | !(lo[0] <= hi[0]);
}

template <bool WORK>
PT_HD void leaf_scan(const DScene &sc, int first, int count, f3 o, f3 d, const SlabBox &cur, float &bt, int &bi, WorkCounters &wc PSTAT_ARG);

template <bool COUNT>
PT_HD void leaf_closest(const DScene &sc, int node, KdNode nd, f3 o, f3 d, const SlabBox &cur, float &bt, int &bi,
                        TravCounters &ctr) {
    const int count = (int)(nd.b & kIdxMask);
    bt = -1.0f;
    bi = -1;
    if (COUNT) { ctr.refs += count; ctr.tris += count; }
    if (leaf_culled(sc, node, o, d)) {
        if (COUNT) ctr.culled += count;
        return;
    }
    PSTAT_LOCAL
    WorkCounters wc = {0u, 0u, 0u, 0u, 0u, 0u};
    leaf_scan<false>(sc, (int)nd.a, count, o, d, cur, bt, bi, wc PSTAT_PASS);
}

// Candidate pre-test (no counterpart in the reference; it changes no result).  A SpherePair holds, for two
// consecutive leaf references, a sphere (centre, R^2) that contains every hit point the leaf loop could accept for
// that triangle, with a margin far above any rounding in play (scene.cpp).  A ray whose LINE passes the centre at
// more than R cannot produce one.  |d| <= 1 by the Ray constructor rule (ray.cuh:10), so |(c-o) x d| underestimates
// the distance -- the safe side; the 2^-26 |c-o|^2 term covers the rounding of the cross product (and of
// P = o + d t in the reference) for origins far from the triangle.  Both candidates go through the same packed
// fp32 instructions (fused multiply-adds are fine here: the test only has to be conservative).
PT_HD v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
PT_HD uint32_t sphere_pair_hit(const SpherePair s, f3 o, f3 d) {
    const v2f cx = s.x - o.x, cy = s.y - o.y, cz = s.z - o.z;
    const v2f qx = vfma(cy, (v2f)(d.z), -(cz * d.y));
    const v2f qy = vfma(cz, (v2f)(d.x), -(cx * d.z));
    const v2f qz = vfma(cx, (v2f)(d.y), -(cy * d.x));
    const v2f q2 = vfma(qz, qz, vfma(qy, qy, qx * qx));
    const v2f D2 = vfma(cz, cz, vfma(cy, cy, cx * cx));
    const v2f thr = vfma(D2, (v2f)(1.4901161e-8f), s.r2);
    return (q2.x <= thr.x ? 1u : 0u) | (q2.y <= thr.y ? 2u : 0u);
}

// The candidate loop proper: `count` references starting at global reference `first`, leaf box in `cur`.  Two
// passes per 64 references.  Pass 1 streams the sphere pairs (two 32-byte records = four candidates per step; pairs
// are indexed by global reference >> 1, so the first/last pair may straddle a neighbouring leaf -- those bits are
// masked off, and the array is padded so every load is in bounds) and marks the candidates whose sphere the ray
// passes: a few per leaf.  Pass 2 gives the marked ones the reference's evaluation in leaf order (the sequential
// "t < best" rule keeps the earliest of equal hits); the lanes of a wave run it together instead of each branching
// out of pass 1 on its own.
template <bool WORK>
PT_HD void leaf_scan(const DScene &sc, int first, int count, f3 o, f3 d, const SlabBox &cur, float &bt, int &bi, WorkCounters &wc PSTAT_ARG) {
    bt = -1.0f;
    bi = -1;
    if (WORK) wc.leaves++;
    const int end = first + count;
    for (int base = first & ~1; base < end; base += 64) {
        const int stop = (end - base < 64) ? end - base : 64;  // bits [max(first-base,0), stop) belong to this leaf
        const SpherePair *sp = sc.leaf_spheres + (base >> 1);
        unsigned long long marked = 0ull;
        for (int p = 0; 2 * p < stop; p += 2) {
            PSTAT(ps, 5);
            const SpherePair s0 = sp[p], s1 = sp[p + 1];
            if (WORK) wc.sphere_pairs += 2;
            marked |= (unsigned long long)(sphere_pair_hit(s0, o, d) | (sphere_pair_hit(s1, o, d) << 2)) << (2 * p);
        }
        marked &= (stop < 64 ? (1ull << stop) - 1ull : ~0ull) & ~((1ull << (first > base ? first - base : 0)) - 1ull);
        while (marked != 0ull) {
            PSTAT(ps, 6);
            const int r = base + (__builtin_ffsll((long long)marked) - 1);
            marked &= marked - 1ull;
            const f4 *rec = sc.leaf_recs + 4 * (size_t)r;
            const f4 pl = rec[0];
            const float t = plane_candidate(pl, o, d, cur, bt);
            if (WORK) wc.planes++;
            if (t > 0.0f) {
                if (WORK) wc.verts++;
                const f4 pa = rec[1], pb = rec[2], pc = rec[3];
                if (inside_triangle(pl, mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z), mk3(pc.x, pc.y, pc.z), o, d, t)) {
                    bt = t;
                    bi = r;
                }
            }
        }
    }
    if (bi >= 0) bi = (int)as_u32(sc.leaf_recs[4 * (size_t)bi + 1].w);  // reference position -> triangle id, once per leaf (rides in a.w)
}

// Mesh::intersect (mesh.cuh:253-305), one ray start to finish.  Stack: push(sp, v) / pop(sp) on a per-lane column.
template <class Stack, bool COUNT>
PT_HD void mesh_closest(const DScene &sc, int root, f3 o, f3 d, Stack &stk, float &best_t, int &best_tri,
                        TravCounters &ctr) {
    best_t = -1.0f;
    best_tri = -1;
    SlabBox cur;
    load_slab(sc, root, o, d, cur);
    if (!(slab_hit(cur, o, d) > 0.0f)) return;  // mesh.cuh:254
    int node = root;
    int sp = 0;
    if (COUNT) { ctr.pops++; ctr.nodes++; }
    KdNode leaf = sc.nodes[node];
    for (;;) {
        PSTAT_LOCAL
        descend_to_leaf<Stack, COUNT>(sc, o, d, cur, node, leaf, sp, stk, ctr, 1 << 30 PSTAT_PASS);
        float bt;
        int bi;
        leaf_closest<COUNT>(sc, node, leaf, o, d, cur, bt, bi, ctr);
        if (bt > 0.0f) {  // mesh.cuh:300: the first leaf with an accepted hit ends the search (Q8)
            best_t = bt;
            best_tri = bi;
            return;
        }
        if (sp == 0) return;
        sp--;
        if (sp >= kStackDepth) return;  // unreachable for trees built by kd_build.cpp (depth <= 16)
        node = stk.pop(sp);
        leaf = sc.nodes[node];
        load_slab(sc, node, o, d, cur);
        if (COUNT) { ctr.pops++; ctr.nodes++; }
    }
}

// Plane::intersect (plane.cuh:18-30), t only
PT_HD float plane_t(f3 point, f3 n, f3 o, f3 d) {
    const float nv = dot3(n, d);
    if (fabsf(nv) < kEps) return -1.0f;
    const float t = dot3(n, point - o) / nv;
    if (t < kEps) return -1.0f;
    return t;
}
// Sphere::intersect (sphere.cuh:19-37), t only
PT_HD float sphere_t(f3 center, float radius, f3 o, f3 d) {
    const f3 dist = o - center;
    const float a = dot3(d, d);
    const float b = dot3(dist, d) * 2.0f;
    const float c = dot3(dist, dist) - radius * radius;
    const float discr = b * b - 4.0f * a * c;
    if (discr < 0.0f) return -1.0f;
    const float sq = sqrtf(discr);
    const float t1 = (-b + sq) / 2.0f / a;
    const float t2 = (-b - sq) / 2.0f / a;
    if (t1 <= 0.0f) return -1.0f;
    return (t2 > 0.0f) ? t2 : t1;
}

// The sphere / plane part of Scene::firstIntersect (kernel.cu:151-158): closest of them, strict '<' keeps the earlier.
// Evaluated for every ray by the stage that GENERATES it (convergent lanes) and parked in the ray's hit slot; the
// extend kernel starts from that answer and only adds the meshes (a mesh hit wins ties: meshes come first).
PT_HD void quadrics_closest(const DScene &sc, f3 o, f3 d, float &out_t, int &out_prim) {
    float best = -1.0f;
    int prim = -1;
    for (int i = 0; i < sc.n_spheres; i++) {
        const f4 s = ldk4(sc.spheres, 2 * (size_t)i);
        const float t = sphere_t(mk3(s.x, s.y, s.z), s.w, o, d);
        if (t > 0.0f && (best < 0.0f || t < best)) { best = t; prim = sc.total_tris + i; }
    }
    for (int i = 0; i < sc.n_planes; i++) {
        const f4 p = ldk4(sc.planes, 2 * (size_t)i), n = ldk4(sc.planes, 2 * (size_t)i + 1);
        const float t = plane_t(mk3(p.x, p.y, p.z), mk3(n.x, n.y, n.z), o, d);
        if (t > 0.0f && (best < 0.0f || t < best)) { best = t; prim = sc.total_tris + sc.n_spheres + i; }
    }
    out_t = best;
    out_prim = prim;
}
// The same for two rays at once (k_shade: the next camera ray and the direct connection of a vertex): every table entry
// is fetched once, and the two rays' tests are independent instruction streams.
PT_HD void quadrics_closest2(const DScene &sc, f3 o1, f3 d1, f3 o2, f3 d2, float &out_t1, int &out_prim1, float &out_t2,
                             int &out_prim2) {
    float best1 = -1.0f, best2 = -1.0f;
    int prim1 = -1, prim2 = -1;
    for (int i = 0; i < sc.n_spheres; i++) {
        const f4 s = ldk4(sc.spheres, 2 * (size_t)i);
        const float t1 = sphere_t(mk3(s.x, s.y, s.z), s.w, o1, d1), t2 = sphere_t(mk3(s.x, s.y, s.z), s.w, o2, d2);
        if (t1 > 0.0f && (best1 < 0.0f || t1 < best1)) { best1 = t1; prim1 = sc.total_tris + i; }
        if (t2 > 0.0f && (best2 < 0.0f || t2 < best2)) { best2 = t2; prim2 = sc.total_tris + i; }
    }
    for (int i = 0; i < sc.n_planes; i++) {
        const f4 p = ldk4(sc.planes, 2 * (size_t)i), n = ldk4(sc.planes, 2 * (size_t)i + 1);
        const float t1 = plane_t(mk3(p.x, p.y, p.z), mk3(n.x, n.y, n.z), o1, d1), t2 = plane_t(mk3(p.x, p.y, p.z), mk3(n.x, n.y, n.z), o2, d2);
        if (t1 > 0.0f && (best1 < 0.0f || t1 < best1)) { best1 = t1; prim1 = sc.total_tris + sc.n_spheres + i; }
        if (t2 > 0.0f && (best2 < 0.0f || t2 < best2)) { best2 = t2; prim2 = sc.total_tris + sc.n_spheres + i; }
    }
    out_t1 = best1; out_prim1 = prim1;
    out_t2 = best2; out_prim2 = prim2;
}
// A ray that misses the root CULL box of every mesh cannot be given a mesh hit by the reference (an accepted hit lies
// inside it, see box_culled): the parked sphere / plane answer is final and the ray never reaches the (divergent)
// extend kernel -- which applies the reference's own root test (mesh.cuh:254) to the rest.  No IEEE division here.
PT_HD bool misses_every_mesh(const DScene &sc, f3 o, f3 d) {
    for (int m = 0; m < sc.n_meshes; m++) {
        const size_t root = (size_t)ldk_i32(&sc.meshes[m].root);
        if (!box_culled(ldk4(sc.tight, 2 * root), ldk4(sc.tight, 2 * root + 1), o, d)) return false;
    }
    return true;
}

// Visibility by position compare (Q10): did the closest hit `t` of the shadow / connection ray (o, d) land on the point
// it was aimed at?  target.w == 0: ray cast from the light at a surface point (kernel.cu:249); else: connection ray
// from a camera vertex to a light vertex (kernel.cu:268).  One definition for every stage that resolves a shadow ray.
PT_HD bool shadow_visible(f3 o, f3 d, float t, f4 target) {
    if (target.w == 2.0f) return t < 0.0f;  // a ray towards the cube map as a light (use_envmap == 2): it must leave the scene
    const f3 P = ray_at(o, d, t);
    const float dist = len3(P - mk3(target.x, target.y, target.z));
    if (target.w == 0.0f) return (t > 0.0f) && (dist < kEps);
    return !((t < 0.0f) || (dist > kEps));
}

// Scene::firstIntersect (kernel.cu:145-160): meshes, spheres, planes; strict '<' keeps the earlier object.
// prim: -1 miss; [0,total_tris) triangle; then spheres; then planes.
template <class Stack, bool COUNT>
PT_HD void scene_closest(const DScene &sc, f3 o, f3 d, Stack &stk, float &out_t, int &out_prim, TravCounters &ctr) {
    float best = -1.0f;
    int prim = -1;
    for (int m = 0; m < sc.n_meshes; m++) {
        float t;
        int tri;
        mesh_closest<Stack, COUNT>(sc, sc.meshes[m].root, o, d, stk, t, tri, ctr);
        if (t > 0.0f && (best < 0.0f || t < best)) { best = t; prim = tri; }
    }
    for (int i = 0; i < sc.n_spheres; i++) {
        const f4 s = sc.spheres[2 * i];
        const float t = sphere_t(mk3(s.x, s.y, s.z), s.w, o, d);
        if (t > 0.0f && (best < 0.0f || t < best)) { best = t; prim = sc.total_tris + i; }
    }
    for (int i = 0; i < sc.n_planes; i++) {
        const f4 p = sc.planes[2 * i], n = sc.planes[2 * i + 1];
        const float t = plane_t(mk3(p.x, p.y, p.z), mk3(n.x, n.y, n.z), o, d);
        if (t > 0.0f && (best < 0.0f || t < best)) { best = t; prim = sc.total_tris + sc.n_spheres + i; }
    }
    out_t = best;
    out_prim = prim;
}

}  // namespace mipt
