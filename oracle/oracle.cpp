/*
 * oracle.cpp -- CPU ORACLE for the MI355X path-tracer core.  TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch CPU restatement (own structure: indexed scene tables, runtime config,
 * primitive ids, counter-based RNG) of the rendering semantics of
 * fazakasreka/cuda-path-tracer.  Every function cites the reference file:line whose
 * arithmetic it follows (paths relative to /root/reference/cuda-path-tracer/).  All
 * floating point is fp32 evaluated in the reference's operation order; build with
 * -ffp-contract=off so no FMA contraction changes a rounding.
 *
 * PARITY UNPINNED (see oracle.h): the reference has no tests/fixtures and could not be
 * built or run here; pins are analytic KATs, brute-force cross-checks and the published
 * Philox4x32-10 vectors.
 *
 * Deliberate, documented deviations from the reference:
 *   Q9  traversal stack is 64 deep; rays that would overflow int[15] are counted.
 *   Q17 light index is clamped to nLights-1.
 *   RNG is Philox4x32-10 keyed on (seed, pixel, sample) instead of cuRAND XORWOW
 *       (mandated by BASELINE.json north_star); uniforms are in (0,1] like curand_uniform.
 *   sinf/cosf in Light::randomSampleRay use one fixed fp32 polynomial evaluation (see fixed_sinf).
 *   kd build: a node with no in-bounds split candidate or zero width becomes a leaf
 *       (the reference reads out of bounds / divides 0/0 there, mesh.cu:72-78).
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <omp.h>

namespace {

// constants/constants.cuh:7,10,12-15
constexpr float EPS = 1e-5f;
constexpr float PI = 3.14159265358979323846f;
constexpr int MAX_LEAF_TRIS = 50;
constexpr int MAX_KD_HEIGHT = 15;
constexpr int ALLOWED_DIFF = 20;
constexpr int SPLIT_SEARCH_DEPTH = 20;
constexpr int MAX_DEPTH = 16;  // upper bound on cam_depth / light_depth accepted

thread_local std::string g_err;

// ---- mutation switches (tests/test_pin_mutations.py, scripts/pin_mutations.py) -------------------------------------------
// The only reference-held pin of this restatement is image-level (two README renders).  To show what that pin can and cannot
// see, every quirk of SURVEY.md 8.1 that the integrator depends on can be flipped to what a "clean" implementation would do,
// one at a time.  Compiled in only with -DORC_MUTATIONS (oracle/liboracle_mut.so): in the normal build MUT() is a constant
// false and the checker's arithmetic and speed are untouched.
enum {
    MUT_Q5_MINX_EPS = 0,       // isVertexInBox: give min-X the +eps slack of the other five bounds (mesh.cuh:132)
    MUT_Q10_SHADOW_FROM_HIT,   // direct connection: shadow ray from the surface point towards the light (kernel.cu:248)
    MUT_Q13_SAMPLED_LIGHT,     // direct connection to the sampled light instead of lights[0] (kernel.cu:244,250)
    MUT_Q14_NO_OVERWRITE,      // a stored light vertex keeps the throughput that ARRIVES at it (kernel.cu:213-222)
    MUT_Q15_SINGLE_COS,        // connection weight with cosCam once (kernel.cu:276)
    MUT_Q16_ROULETTE_PDF,      // roulette probability not folded into the path pdf (kernel.cu:287-296)
    MUT_Q1_EXACT_NORMALIZE,    // normalize() divides by |v|, no +eps (math.cuh:28)
    MUT_DEV7_UNFUSED,          // ray_at as the source text says: product, then sum, two roundings (deviation 7 off)
    MUT_Q12_UNIFORM_EMISSION,  // light emission direction uniform on the sphere (light.cuh:35-38)
    MUT_Q11_KEEP_CAMERA_PATH,  // a first light ray that hits nothing does not end the sample (kernel.cu:185)
    MUT_Q3_NORMALISED_PARALLEL,// triangle parallel test on the unit normal (mesh.cuh:48-50)
    MUT_COUNT
};
#ifdef ORC_MUTATIONS
uint32_t g_mut = 0u;
#define MUT(bit) (((g_mut >> (bit)) & 1u) != 0u)
#else
#define MUT(bit) false
#endif

// ---------------------------------------------------------------- math/math.cuh:18-47
struct V3 {
    float x = 0.f, y = 0.f, z = 0.f;
};
inline V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // math.cuh:41-43
inline V3 cross(V3 a, V3 b) {                                               // math.cuh:45-47
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  // math.cuh:29
inline V3 normalize(V3 a) { return a * (1 / (length(a) + (MUT(MUT_Q1_EXACT_NORMALIZE) ? 0.0f : EPS))); }  // math.cuh:28 (Q1)
inline float average(V3 a) { return (a.x + a.y + a.z) / 3.0f; }               // math.cuh:30
inline float axis_of(V3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

// ---------------------------------------------------------------- math/math.cuh:50-158, math.cu:3-39
struct V4 {
    float v[4] = {0, 0, 0, 0};
};
inline V4 mk4(float a, float b, float c, float d) { V4 r; r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d; return r; }
inline V4 from_v3_swapped(V3 p, float w) { return mk4(p.x, p.z, p.y, w); }  // math.cuh:55 (Q2: y/z swap)
inline V4 operator*(float a, V4 r) { return mk4(r.v[0] * a, r.v[1] * a, r.v[2] * a, r.v[3] * a); }  // math.cuh:74
inline V4 operator+(V4 a, V4 b) { return mk4(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]); }
struct M4 {
    V4 r[4];
};
inline V4 mul(V4 v, const M4 &m) { return v.v[0] * m.r[0] + v.v[1] * m.r[1] + v.v[2] * m.r[2] + v.v[3] * m.r[3]; }  // :117
inline M4 mul(const M4 &a, const M4 &b) {  // math.cuh:121-125
    M4 o;
    for (int i = 0; i < 4; i++) o.r[i] = mul(a.r[i], b);
    return o;
}
inline M4 translate_m(V3 t) {  // math.cuh:127-132
    M4 m; m.r[0] = mk4(1, 0, 0, 0); m.r[1] = mk4(0, 1, 0, 0); m.r[2] = mk4(0, 0, 1, 0); m.r[3] = mk4(t.x, t.y, t.z, 1);
    return m;
}
inline M4 scale_m(V3 s) {  // math.cuh:134-140
    M4 m; m.r[0] = mk4(s.x, 0, 0, 0); m.r[1] = mk4(0, s.y, 0, 0); m.r[2] = mk4(0, 0, s.z, 0); m.r[3] = mk4(0, 0, 0, 1);
    return m;
}
inline M4 rotation_m(float angle, V3 w) {  // math.cuh:142-150
    float c = cosf(angle), s = sinf(angle);
    w = normalize(w);
    M4 m;
    m.r[0] = mk4(c * (1.0f - w.x * w.x) + w.x * w.x, w.x * w.y * (1.0f - c) + w.z * s, w.x * w.z * (1.0f - c) - w.y * s, 0.0f);
    m.r[1] = mk4(w.x * w.y * (1.0f - c) - w.z * s, c * (1.0f - w.y * w.y) + w.y * w.y, w.y * w.z * (1.0f - c) + w.x * s, 0.0f);
    m.r[2] = mk4(w.x * w.z * (1.0f - c) + w.y * s, w.y * w.z * (1.0f - c) - w.x * s, c * (1.0f - w.z * w.z) + w.z * w.z, 0.0f);
    m.r[3] = mk4(0, 0, 0, 1);
    return m;
}
inline M4 srt_m(V3 scale, V3 rot, V3 tr) {  // math.cuh:152-158
    return mul(mul(mul(mul(scale_m(scale), rotation_m(rot.x, mk(1, 0, 0))), rotation_m(rot.y, mk(0, 1, 0))),
                   rotation_m(rot.z, mk(0, 0, 1))),
               translate_m(tr));
}
inline M4 inverse_m(const M4 &m) {  // math.cu:3-39 (affine "inverse" exactly as written there)
    V3 a = mk(m.r[0].v[0], m.r[0].v[1], m.r[0].v[2]);
    V3 b = mk(m.r[1].v[0], m.r[1].v[1], m.r[1].v[2]);
    V3 c = mk(m.r[2].v[0], m.r[2].v[1], m.r[2].v[2]);
    V3 t = mk(m.r[3].v[0], m.r[3].v[1], m.r[3].v[2]);
    float det = dot(cross(a, b), c);
    M4 o;
    if (fabsf(det) < EPS) {
        o.r[0] = mk4(1, 0, 0, 0); o.r[1] = mk4(0, 1, 0, 0); o.r[2] = mk4(0, 0, 1, 0); o.r[3] = mk4(0, 0, 0, 1);
        return o;
    }
    float inv = 1.0f / det;
    V3 r0 = cross(b, c) * inv, r1 = cross(c, a) * inv, r2 = cross(a, b) * inv;
    float tx = -dot(r0, t), ty = -dot(r1, t), tz = -dot(r2, t);
    o.r[0] = mk4(r0.x, r0.y, r0.z, 0); o.r[1] = mk4(r1.x, r1.y, r1.z, 0); o.r[2] = mk4(r2.x, r2.y, r2.z, 0);
    o.r[3] = mk4(tx, ty, tz, 1);
    return o;
}
inline M4 transpose_m(const M4 &m) {  // math.cuh:104-112
    M4 o;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) o.r[i].v[j] = m.r[j].v[i];
    return o;
}

// ---------------------------------------------------------------- pathTracing/{ray,hit,material}.cuh
struct Ray {
    V3 o, d;
};
inline Ray make_ray(V3 o, V3 dir) { Ray r; r.o = o; r.d = normalize(dir); return r; }  // ray.cuh:10
// `ray.start + ray.dir * t` (mesh.cuh:59,90,153-162; plane.cuh:25; sphere.cuh:32): ONE rounding per component -- the fused
// multiply-add that nvcc's default -fmad=true makes of a product feeding a sum.  The only contraction of the reference's
// build that its published renders show (tests/test_oracle_golden.py, DESIGN.md 2 deviation 7); everything else in this
// file is compiled with -ffp-contract=off.
inline V3 ray_at(const Ray &r, float t) {
    if (MUT(MUT_DEV7_UNFUSED)) return r.o + r.d * t;
    return mk(fmaf(r.d.x, t, r.o.x), fmaf(r.d.y, t, r.o.y), fmaf(r.d.z, t, r.o.z));
}
struct Hit {  // hit.cuh:9-15 (+ prim id)
    float t = -1.0f;
    V3 pos, nrm;
    int mat = -1;
    int prim = -1;
};
struct Material {  // material.cuh:5-13
    V3 kd, ks;
};
struct Light {  // light.cuh:7-18
    V3 loc, power;
};
struct Plane {  // plane.cuh:9-17
    V3 point, normal;
    int mat;
};
struct Sphere {  // sphere.cuh:9-18
    V3 center;
    float radius;
    int mat;
};
struct Tri {  // mesh.cuh:13-34
    V3 pa, na, pb, nb, pc, nc;
};
struct Box {  // mesh.cuh:115-127; lo/hi[axis]
    float lo[3], hi[3];
};

// mesh.cuh:131-139 (Q5: min-X has no epsilon, every other bound has +eps slack)
inline bool box_contains(const Box &b, V3 p) {
    return (b.lo[0] <= p.x + (MUT(MUT_Q5_MINX_EPS) ? EPS : 0.0f) && p.x <= b.hi[0] + EPS && b.lo[1] <= p.y + EPS && p.y <= b.hi[1] + EPS &&
            b.lo[2] <= p.z + EPS && p.z <= b.hi[2] + EPS);
}
// mesh.cuh:141-146 (Q6: at least one vertex inside)
inline bool box_touches_tri(const Box &b, const Tri &t) {
    return box_contains(b, t.pa) || box_contains(b, t.pb) || box_contains(b, t.pc);
}
// mesh.cuh:148-170: smallest t>0 among the six slab planes whose point passes box_contains; -1 if none
inline float box_intersect(const Box &b, const Ray &r) {
    float t[6];
    V3 p[6];
    t[0] = (b.lo[0] - r.o.x) / r.d.x;
    t[1] = (b.hi[0] - r.o.x) / r.d.x;
    t[2] = (b.lo[1] - r.o.y) / r.d.y;
    t[3] = (b.hi[1] - r.o.y) / r.d.y;
    t[4] = (b.lo[2] - r.o.z) / r.d.z;
    t[5] = (b.hi[2] - r.o.z) / r.d.z;
    for (int i = 0; i < 6; i++) p[i] = ray_at(r, t[i]);
    float best = -1.0f;
    for (int i = 0; i < 6; i++)
        if (t[i] > 0.0f && (best < 0.0f || t[i] < best) && box_contains(b, p[i])) best = t[i];
    return best;
}
// mesh.cuh:36-90: returns t only (normal is computed separately for the winner; identical result
// because the candidate normal never influences acceptance).  Q3, Q4.
inline float tri_intersect_t(const Tri &tr, const Ray &r) {
    V3 ab = tr.pb - tr.pa, ac = tr.pc - tr.pa;
    V3 N = cross(ab, ac);
    float nd = dot(N, r.d);
    if (MUT(MUT_Q3_NORMALISED_PARALLEL) ? fabsf(nd) < EPS * length(N) : fabsf(nd) < EPS) return -1.0f;  // :49 (hit.t keeps its default -1)
    float d = -dot(N, tr.pa);
    float t = -(dot(N, r.o) + d) / nd;
    if (t < 0.0f) return t;  // :56 raw negative
    V3 P = ray_at(r, t);
    V3 C = cross(ab, P - tr.pa);
    if (dot(N, C) < 0.0f) return -1.0f;
    V3 bc = tr.pc - tr.pb;
    C = cross(bc, P - tr.pb);
    if (dot(N, C) < 0.0f) return -1.0f;
    V3 ca = tr.pa - tr.pc;
    C = cross(ca, P - tr.pc);
    if (dot(N, C) < 0.0f) return -1.0f;
    return t;
}
// mesh.cuh:90-110: barycentric interpolated vertex normal, flipped toward the ray
inline V3 tri_normal(const Tri &tr, const Ray &r, float t) {
    V3 P = ray_at(r, t);
    V3 va = tr.pa - tr.pb, vc = tr.pc - tr.pb, vh = P - tr.pb;
    float daa = dot(va, va), dac = dot(va, vc), dcc = dot(vc, vc);
    float denom = daa * dcc - dac * dac;
    float dha = dot(vh, va), dhc = dot(vh, vc);
    float b0 = (dcc * dha - dac * dhc) / denom;
    float b1 = (daa * dhc - dac * dha) / denom;
    float b2 = 1.0f - b0 - b1;
    V3 n = normalize(tr.na * b0 + tr.nb * b2 + tr.nc * b1);
    if (dot(n, r.d) > 0.0f) n = n * (-1.0f);
    return n;
}
// plane.cuh:18-30
inline Hit plane_intersect(const Plane &pl, const Ray &r) {
    Hit h;
    float nv = dot(pl.normal, r.d);
    if (fabsf(nv) < EPS) return h;
    float t = dot(pl.normal, pl.point - r.o) / nv;
    if (t < EPS) return h;
    h.t = t;
    h.pos = ray_at(r, h.t);
    h.nrm = pl.normal;
    if (dot(h.nrm, r.d) > 0.0f) h.nrm = h.nrm * (-1.0f);
    h.mat = pl.mat;
    return h;
}
// sphere.cuh:19-37
inline Hit sphere_intersect(const Sphere &s, const Ray &r) {
    Hit h;
    V3 dist = r.o - s.center;
    float a = dot(r.d, r.d);
    float b = dot(dist, r.d) * 2.0f;
    float c = dot(dist, dist) - s.radius * s.radius;
    float discr = b * b - 4.0f * a * c;
    if (discr < 0.0f) return h;
    float sq = sqrtf(discr);
    float t1 = (-b + sq) / 2.0f / a;
    float t2 = (-b - sq) / 2.0f / a;
    if (t1 <= 0.0f) return h;
    h.t = (t2 > 0.0f) ? t2 : t1;
    h.pos = ray_at(r, h.t);
    h.nrm = (h.pos - s.center) / s.radius;
    h.mat = s.mat;
    if (dot(h.nrm, r.d) > 0.0f) h.nrm = h.nrm * (-1.0f);
    return h;
}

// ---------------------------------------------------------------- kd-tree (mesh.cu:5-222)
struct BuildNode {
    Box box;
    int axis = 0;
    std::unique_ptr<BuildNode> left, right;
    std::vector<int> tris;  // leaf payload
    bool degenerate = false;
};
struct FlatNode {  // the fields of mesh.cuh:204-220 that traversal needs
    Box box;
    int axis = 0;
    int left = -1, right = -1;  // -1/-1 = leaf
    int first = 0, count = 0;
};
struct Mesh {
    std::vector<Tri> tris;
    std::vector<FlatNode> nodes;
    std::vector<int> refs;
    int mat = -1;
    int tri_base = 0;
    int64_t degenerate_nodes = 0;
};

// mesh.cu:5-30: child plane = lo + (hi-lo)*pct, recomputed from the percentage
inline void box_half(const Box &b, Box &l, Box &r, int axis, float pct) {
    l = b;
    r = b;
    l.hi[axis] = b.lo[axis] + (b.hi[axis] - b.lo[axis]) * pct;
    r.lo[axis] = l.hi[axis];
}

// mesh.cu:32-136
void build_node(const std::vector<Tri> &T, BuildNode *nd, std::vector<int> tris, int depth) {
    if ((int)tris.size() <= MAX_LEAF_TRIS || depth > MAX_KD_HEIGHT) {  // :33
        nd->tris = std::move(tris);
        return;
    }
    const int axis = nd->axis;
    const Box &box = nd->box;
    // :44-64 candidate split coordinates strictly inside the box on this axis (mesh.cuh:180-192)
    std::vector<float> cand;
    cand.reserve(tris.size() * 3);
    for (int ti : tris) {
        float c3[3] = {axis_of(T[ti].pa, axis), axis_of(T[ti].pb, axis), axis_of(T[ti].pc, axis)};
        for (float c : c3)
            if (c > box.lo[axis] && c < box.hi[axis]) cand.push_back(c);
    }
    const int n_in = (int)cand.size();
    const float width = box.hi[axis] - box.lo[axis];
    if (n_in == 0 || !(width > 0.0f)) {  // documented deviation: reference is undefined here
        nd->tris = std::move(tris);
        nd->degenerate = true;
        return;
    }
    std::sort(cand.begin(), cand.end(), std::less<float>());  // :68
    nd->left.reset(new BuildNode());
    nd->right.reset(new BuildNode());
    Box &lb = nd->left->box, &rb = nd->right->box;
    int idx = n_in / 2;  // :72
    for (int i = 2; i <= SPLIT_SEARCH_DEPTH + 1; i++) {  // :73
        float sp = cand[idx];
        box_half(box, lb, rb, axis, (sp - box.lo[axis]) / width);  // :76-78
        int nl = 0, nr = 0;
        for (int ti : tris) {  // :81-88
            if (box_touches_tri(lb, T[ti])) nl++;
            if (box_touches_tri(rb, T[ti])) nr++;
        }
        if ((nl - nr) < ALLOWED_DIFF && -ALLOWED_DIFF < (nl - nr)) break;  // :90-93
        else if (nl > nr) {
            idx -= n_in / (2 * i);
            if (idx < 0) { idx = 0; break; }
        } else {
            idx += n_in / (2 * i);
            if (idx > n_in - 1) { idx = n_in - 1; break; }
        }
    }
    box_half(box, lb, rb, axis, (cand[idx] - box.lo[axis]) / width);  // :112-114
    std::vector<int> lt, rt;
    for (int ti : tris) {  // :123-130 (straddlers go to both; Q6 membership)
        if (box_touches_tri(lb, T[ti])) lt.push_back(ti);
        if (box_touches_tri(rb, T[ti])) rt.push_back(ti);
    }
    tris.clear();
    tris.shrink_to_fit();
    nd->left->axis = nd->right->axis = (axis + 1) % 3;  // :132-133, math.cuh:9-15
    BuildNode *l = nd->left.get(), *r = nd->right.get();
    if (lt.size() + rt.size() > 20000) {
#pragma omp task shared(T) firstprivate(l, depth)
        build_node(T, l, std::move(lt), depth + 1);
#pragma omp task shared(T) firstprivate(r, depth)
        build_node(T, r, std::move(rt), depth + 1);
#pragma omp taskwait
    } else {
        build_node(T, l, std::move(lt), depth + 1);
        build_node(T, r, std::move(rt), depth + 1);
    }
}

// mesh.cu:138-155 root box + axis X; mesh.cu:173-222 flatten (right chain contiguous, lefts FIFO)
void build_mesh_tree(Mesh &m) {
    std::unique_ptr<BuildNode> root(new BuildNode());
    const std::vector<Tri> &T = m.tris;
    Box &b = root->box;
    b.lo[0] = b.hi[0] = T[0].pa.x; b.lo[1] = b.hi[1] = T[0].pa.y; b.lo[2] = b.hi[2] = T[0].pa.z;
    auto grow = [&](V3 p) {  // mesh.cu:163-170
        if (p.x < b.lo[0]) b.lo[0] = p.x;
        if (p.x > b.hi[0]) b.hi[0] = p.x;
        if (p.y < b.lo[1]) b.lo[1] = p.y;
        if (p.y > b.hi[1]) b.hi[1] = p.y;
        if (p.z < b.lo[2]) b.lo[2] = p.z;
        if (p.z > b.hi[2]) b.hi[2] = p.z;
    };
    for (const Tri &t : T) { grow(t.pa); grow(t.pb); grow(t.pc); }
    root->axis = 0;
    std::vector<int> all(T.size());
    for (size_t i = 0; i < T.size(); i++) all[i] = (int)i;
    BuildNode *rp = root.get();
#pragma omp parallel
#pragma omp single
    build_node(T, rp, std::move(all), 0);

    m.nodes.clear();
    m.refs.clear();
    m.degenerate_nodes = 0;
    struct Pending { BuildNode *n; int parent; bool is_left; };
    std::deque<Pending> q;
    q.push_back({root.get(), -1, false});
    while (!q.empty()) {
        Pending p = q.front();
        q.pop_front();
        BuildNode *cur = p.n;
        int parent = p.parent;
        bool is_left = p.is_left;
        for (;;) {
            int idx = (int)m.nodes.size();
            FlatNode fn;
            fn.box = cur->box;
            fn.axis = cur->axis;
            if (parent >= 0) {
                if (is_left) m.nodes[parent].left = idx;
                else m.nodes[parent].right = idx;
            }
            if (cur->degenerate) m.degenerate_nodes++;
            if (!cur->left) {
                fn.first = (int)m.refs.size();
                fn.count = (int)cur->tris.size();
                m.refs.insert(m.refs.end(), cur->tris.begin(), cur->tris.end());
                m.nodes.push_back(fn);
                break;
            }
            m.nodes.push_back(fn);
            q.push_back({cur->left.get(), idx, true});
            parent = idx;
            is_left = false;
            cur = cur->right.get();
        }
    }
}

struct Ctr {
    int64_t rays = 0, nodes = 0, pops = 0, refs = 0, tris = 0, hits = 0, box_tests = 0, overflow = 0, max_stack = 0;
    void add(const Ctr &o) {
        rays += o.rays; nodes += o.nodes; pops += o.pops; refs += o.refs; tris += o.tris; hits += o.hits;
        box_tests += o.box_tests; overflow += o.overflow; max_stack = std::max(max_stack, o.max_stack);
    }
};

// mesh.cuh:253-305 (Q7 child ordering, Q8 leaf filter + first-leaf early return)
inline void mesh_intersect(const Mesh &m, const Ray &r, float &best_t, int &best_tri, Ctr &c) {
    best_t = -1.0f;
    best_tri = -1;
    c.box_tests++;
    if (!(box_intersect(m.nodes[0].box, r) > 0)) return;  // :254
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    bool over = false;
    while (sp > 0) {
        int cur = stack[--sp];
        c.pops++;
        c.nodes++;
        while (m.nodes[cur].left >= 0) {  // :265
            const FlatNode &n = m.nodes[cur];
            float rt = box_intersect(m.nodes[n.right].box, r);
            float lt = box_intersect(m.nodes[n.left].box, r);
            c.box_tests += 2;
            bool rf = rt < lt;
            float small_t = rf ? rt : lt, big_t = rf ? lt : rt;
            int small_i = rf ? n.right : n.left, big_i = rf ? n.left : n.right;
            cur = (small_t > 0) ? small_i : big_i;  // :281
            c.nodes++;
            if (small_t > 0 && big_t > 0) {  // :285
                if (sp >= MAX_KD_HEIGHT) over = true;  // Q9: reference stack is int[15]
                stack[sp++] = big_i;
                if (sp > c.max_stack) c.max_stack = sp;
            }
        }
        const FlatNode &leaf = m.nodes[cur];
        float bt = -1.0f;
        int bi = -1;
        c.refs += leaf.count;
        c.tris += leaf.count;
        for (int i = 0; i < leaf.count; i++) {  // :291-299
            int ti = m.refs[leaf.first + i];
            float t = tri_intersect_t(m.tris[ti], r);
            if (t > 0 && (bt < 0 || t < bt) && box_contains(leaf.box, ray_at(r, t))) {
                bt = t;
                bi = ti;
            }
        }
        if (bt > 0) {  // :300
            best_t = bt;
            best_tri = bi;
            break;
        }
    }
    if (over) c.overflow++;
}

inline void mesh_intersect_brute(const Mesh &m, const Ray &r, float &best_t, int &best_tri) {
    best_t = -1.0f;
    best_tri = -1;
    for (size_t i = 0; i < m.tris.size(); i++) {
        float t = tri_intersect_t(m.tris[i], r);
        if (t > 0 && (best_t < 0 || t < best_t)) { best_t = t; best_tri = (int)i; }
    }
}

// ---------------------------------------------------------------- HDRLoader/HDRLoader.cu:29-191
struct HdrImage {
    int w = 0, h = 0;
    std::vector<float> cols;
};
inline float rgbe_component(int expo, int val) {  // HDRLoader.cu:99-104
    float v = val / 256.0f;
    float d = (float)pow(2, expo);
    return v * d;
}
bool hdr_old_decrunch(unsigned char (*scan)[4], int len, FILE *f) {  // :161-191
    int rshift = 0;
    int pos = 0;
    while (len > 0) {
        scan[pos][0] = (unsigned char)fgetc(f);
        scan[pos][1] = (unsigned char)fgetc(f);
        scan[pos][2] = (unsigned char)fgetc(f);
        scan[pos][3] = (unsigned char)fgetc(f);
        if (feof(f)) return false;
        if (scan[pos][0] == 1 && scan[pos][1] == 1 && scan[pos][2] == 1) {
            for (int i = scan[pos][3] << rshift; i > 0 && len > 0; i--) {
                if (pos > 0) memcpy(scan[pos], scan[pos - 1], 4);
                pos++;
                len--;
            }
            rshift += 8;
        } else {
            pos++;
            len--;
            rshift = 0;
        }
    }
    return true;
}
bool hdr_decrunch(unsigned char (*scan)[4], int len, FILE *f) {  // :118-159
    if (len < 8 || len > 0x7fff) return hdr_old_decrunch(scan, len, f);
    int i = fgetc(f);
    if (i != 2) {
        fseek(f, -1, SEEK_CUR);
        return hdr_old_decrunch(scan, len, f);
    }
    scan[0][1] = (unsigned char)fgetc(f);
    scan[0][2] = (unsigned char)fgetc(f);
    i = fgetc(f);
    if (scan[0][1] != 2 || (scan[0][2] & 128)) {
        scan[0][0] = 2;
        scan[0][3] = (unsigned char)i;
        return hdr_old_decrunch(scan + 1, len - 1, f);
    }
    for (int comp = 0; comp < 4; comp++) {
        for (int j = 0; j < len;) {
            int ci = fgetc(f);
            if (ci == EOF) return false;
            unsigned char code = (unsigned char)ci;
            if (code > 128) {
                code &= 127;
                unsigned char val = (unsigned char)fgetc(f);
                while (code-- && j < len) scan[j++][comp] = val;
            } else {
                while (code-- && j < len) scan[j++][comp] = (unsigned char)fgetc(f);
            }
        }
    }
    return feof(f) ? false : true;
}
bool hdr_load(const char *path, HdrImage &img) {  // :29-97; rows stay in file order (top row first)
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[10];
    if (fread(magic, 10, 1, f) != 1 || memcmp(magic, "#?RADIANCE", 10)) { fclose(f); return false; }
    fseek(f, 1, SEEK_CUR);
    int c = 0, oldc;
    for (;;) {  // :50-56 header ends at an empty line
        oldc = c;
        c = fgetc(f);
        if (c == EOF) { fclose(f); return false; }
        if (c == 0xa && oldc == 0xa) break;
    }
    char reso[200];
    int i = 0;
    for (;;) {
        c = fgetc(f);
        if (c == EOF || i >= 199) { fclose(f); return false; }
        reso[i++] = (char)c;
        if (c == 0xa) break;
    }
    reso[i] = 0;
    int w = 0, h = 0;
    if (sscanf(reso, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { fclose(f); return false; }
    img.w = w;
    img.h = h;
    img.cols.assign((size_t)w * h * 3, 0.0f);
    std::vector<unsigned char> scanbuf((size_t)w * 4);
    unsigned char(*scan)[4] = reinterpret_cast<unsigned char(*)[4]>(scanbuf.data());
    float *cols = img.cols.data();
    for (int y = h - 1; y >= 0; y--) {
        if (!hdr_decrunch(scan, w, f)) break;
        for (int x = 0; x < w; x++) {  // workOnRGBE :106-116
            int expo = scan[x][3] - 128;
            cols[0] = rgbe_component(expo, scan[x][0]);
            cols[1] = rgbe_component(expo, scan[x][1]);
            cols[2] = rgbe_component(expo, scan[x][2]);
            cols += 3;
        }
    }
    fclose(f);
    return true;
}

// ---------------------------------------------------------------- RNG (replaces random.cuh:7-9 / kernel.cu:331)
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
constexpr uint32_t RNG_KEY1 = 0x6d697074u;  // 'mipt'
struct Rng {  // stream = sequential draws d = 0,1,2,... of one (seed, pixel, sample)
    uint32_t seed, pixel, sample, draw = 0, cached_block = 0xffffffffu, vals[4];
    Rng(uint32_t s, uint32_t p, uint32_t smp) : seed(s), pixel(p), sample(smp) {}
    float u() {  // uniform in (0,1] like curand_uniform (random.cuh:8)
        uint32_t blk = draw >> 2;
        if (blk != cached_block) {
            uint32_t ctr[4] = {blk, sample, pixel, 0u}, key[2] = {seed, RNG_KEY1};
            philox4x32_10(ctr, key, vals);
            cached_block = blk;
        }
        uint32_t x = vals[draw & 3];
        draw++;
        return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
    }
};

// ---------------------------------------------------------------- scene
struct Camera {  // camera.cuh:6-21
    V3 eye, lookat, right, up;
    bool set = false;
};
}  // namespace

struct orc_scene {
    Camera cam;
    std::vector<Material> mats;
    std::vector<Light> lights;
    std::vector<Plane> planes;
    std::vector<Sphere> spheres;
    std::vector<Mesh> meshes;
    HdrImage env[6];  // px,py,pz,nx,ny,nz (envMap.cuh:10-15)
    bool has_env = false;
    // build-defined extension (use_envmap == 2): the cube map as a light, importance-sampled by texel (see env_light_build)
    std::vector<float> env_marg, env_cond;
    bool env_light_ok = false;
    bool committed = false;
    int total_tris = 0;
};

namespace {

// kernel.cu:145-160: meshes, then spheres, then planes; strict '<' keeps the earlier object on ties
inline Hit first_intersect(const orc_scene &sc, const Ray &r, Ctr &c, bool brute = false) {
    Hit best;
    c.rays++;
    int tri_mesh = -1, tri_idx = -1;
    for (size_t i = 0; i < sc.meshes.size(); i++) {
        const Mesh &m = sc.meshes[i];
        float t;
        int ti;
        if (brute) mesh_intersect_brute(m, r, t, ti);
        else mesh_intersect(m, r, t, ti, c);
        if (t > 0 && (best.t < 0 || t < best.t)) {
            best.t = t;
            best.prim = m.tri_base + ti;
            best.mat = m.mat;          // mesh.cuh:332
            best.pos = ray_at(r, t);  // mesh.cuh:90
            tri_mesh = (int)i;         // normal of the winner is resolved below
            tri_idx = ti;
        }
    }
    bool tri_hit = best.prim >= 0;
    for (size_t i = 0; i < sc.spheres.size(); i++) {
        Hit h = sphere_intersect(sc.spheres[i], r);
        if (h.t > 0 && (best.t < 0 || h.t < best.t)) {
            best = h;
            best.prim = sc.total_tris + (int)i;
            tri_hit = false;
        }
    }
    for (size_t i = 0; i < sc.planes.size(); i++) {
        Hit h = plane_intersect(sc.planes[i], r);
        if (h.t > 0 && (best.t < 0 || h.t < best.t)) {
            best = h;
            best.prim = sc.total_tris + (int)sc.spheres.size() + (int)i;
            tri_hit = false;
        }
    }
    if (tri_hit) {
        best.nrm = tri_normal(sc.meshes[tri_mesh].tris[tri_idx], r, best.t);
        c.hits++;
    }
    return best;
}

// camera.cuh:9-16
void camera_set(Camera &c, V3 eye, V3 lookat, V3 vup, float fov) {
    c.eye = eye;
    c.lookat = lookat;
    V3 w = eye - lookat;
    float f = length(w);
    c.right = normalize(cross(vup, w)) * f * tanf(fov / 2.0f);
    c.up = normalize(cross(w, c.right)) * f * tanf(fov / 2.0f);
    c.set = true;
}
// camera.cuh:17-20 with runtime W,H (screenWidth/Height are unsigned ints converted to float)
inline Ray camera_ray(const Camera &c, float X, float Y, int W, int H) {
    V3 dir = c.lookat + c.right * (2.0f * X / (float)W - 1.0f) + c.up * (2.0f * Y / (float)H - 1.0f) - c.eye;
    return make_ray(c.eye, dir);
}
// light.cuh:25-29
inline V3 radiance_at(const Light &l, V3 p) {
    float d2 = dot(l.loc - p, l.loc - p);
    if (d2 < EPS) d2 = EPS;
    return l.power / d2 / 4.0f / PI;
}
// sinf/cosf of light.cuh:35-38, argument in [0, 2*pi].  The last ulp of these functions is
// implementation-defined on every platform (CUDA's included); the build fixes one evaluation --
// Cody-Waite reduction by pi/4 and the classic single-precision minimax polynomials -- so that the
// GPU and this oracle agree bit for bit.  Restated here independently of the product header.
inline void trig_reduce(float x, int &j, float &r) {
    j = (int)(x * 1.27323954473516f);
    if (j & 1) j += 1;
    float y = (float)j;
    r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
inline float trig_sin_kernel(float r) {
    float z = r * r;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}
inline float trig_cos_kernel(float r) {
    float z = r * r;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}
inline float fixed_sinf(float x) {
    int j; float r;
    trig_reduce(x, j, r);
    j &= 7;
    float sign = 1.0f;
    if (j > 3) { sign = -1.0f; j -= 4; }
    float y = (j == 1 || j == 2) ? trig_cos_kernel(r) : trig_sin_kernel(r);
    return sign * y;
}
inline float fixed_cosf(float x) {
    int j; float r;
    trig_reduce(x, j, r);
    j &= 7;
    float sign = 1.0f;
    if (j > 3) { sign = -1.0f; j -= 4; }
    if (j > 1) sign = -sign;
    float y = (j == 1 || j == 2) ? trig_sin_kernel(r) : trig_cos_kernel(r);
    return sign * y;
}
// Diagnostic of orc_explain_sample: which operand of which reference expression made a sample non-finite.
struct Probe {
    int first_cause = 0;     // ORC_NF_* of the FIRST non-finite contribution added to the sample
    int light_vertex = -1;   // j of the connection (ORC_NF_LIGHT_WEIGHT / _GEOMETRY / _CAMERA_WEIGHT)
    int cam_vertex = -1;     // i of the camera vertex
    int sqrt_negative = 0;   // SampleDiffuse calls whose 1 - x*x - y*y was < 0 (samplers.cuh:19: z = NaN)
    int zero_pdf_vertices = 0;  // light vertices stored with pdf == 0 (SampleDiffuse returned z/pi == 0)
    float operand[4] = {0, 0, 0, 0};
};
thread_local Probe *g_probe = nullptr;
inline bool finite3(V3 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }
// light.cuh:30-39 (Q12: not uniform on the sphere)
inline V3 light_sample_dir(Rng &rng) {
    V3 X = mk(0, 0, 1), Y = mk(0, 1, 0), Z = mk(1, 0, 0);
    float alpha = rng.u() * 2.0f * PI;
    float beta = rng.u() * 2.0f * PI;
    if (MUT(MUT_Q12_UNIFORM_EMISSION)) {  // same two draws: azimuth alpha, cos(polar) uniform in [-1, 1)
        const float cz = 1.0f - beta / PI, sz = sqrtf(fmaxf(0.0f, 1.0f - cz * cz));
        return normalize((X * fixed_cosf(alpha) + Z * fixed_sinf(alpha)) * sz + Y * cz);
    }
    return normalize((X * fixed_cosf(alpha) + Z * fixed_sinf(alpha)) * fixed_sinf(beta) + Y * fixed_cosf(beta));
}
// samplers.cuh:8-23 (rejection sampling in the unit disc; frame from N)
inline float sample_diffuse(V3 N, V3 &out, Rng &rng) {
    V3 T = cross(N, mk(1, 0, 0));
    if (length(T) < EPS) T = cross(N, mk(0, 0, 1));
    T = normalize(T);
    V3 B = cross(N, T);
    float x, y, z;
    do {
        x = 2.0f * rng.u() - 1.0f;
        y = 2.0f * rng.u() - 1.0f;
    } while (x * x + y * y > 1.0f);
    z = sqrtf(1.0f - x * x - y * y);
    if (g_probe && (1.0f - x * x - y * y) < 0.0f) g_probe->sqrt_negative++;
    out = N * z + T * x + B * y;
    return z / PI;
}
// samplers.cuh:46-49
inline float sample_mirror(V3 N, V3 in, V3 &out) {
    out = in - N * dot(N, in) * 2.0f;
    return 1.0f;
}
// envMap.cuh:50-89 (Q19 flattened index clamped only above); face order px,py,pz,nx,ny,nz
inline V3 env_lookup(const orc_scene &sc, V3 d, int *face_out = nullptr, int *pos_out = nullptr) {
    int face, pos;
    if ((fabsf(d.x) >= fabsf(d.y)) && (fabsf(d.x) >= fabsf(d.z))) {
        face = d.x > 0.0f ? 0 : 3;
        const HdrImage &im = sc.env[face];
        pos = ((int)((1.0f - ((d.y / fabsf(d.x) + 1.0f) / 2.0f)) * im.h)) * 3 * im.w +
              ((int)((1.0f - ((d.z / d.x + 1.0f) / 2.0f)) * im.w)) * 3;
    } else if ((fabsf(d.y) >= fabsf(d.x)) && (fabsf(d.y) >= fabsf(d.z))) {
        face = d.y > 0.0f ? 1 : 4;
        const HdrImage &im = sc.env[face];
        pos = ((int)((d.z / d.y + 1.0f) / 2.0f * im.h)) * 3 * im.w + ((int)((d.x / fabsf(d.y) + 1.0f) / 2.0f * im.w)) * 3;
    } else {
        face = d.z > 0.0f ? 2 : 5;
        const HdrImage &im = sc.env[face];
        pos = ((int)((1.0f - ((d.y / fabsf(d.z) + 1.0f) / 2.0f)) * im.h)) * 3 * im.w +
              ((int)((d.x / d.z + 1.0f) / 2.0f * im.w)) * 3;
    }
    const HdrImage &im = sc.env[face];
    if (pos > im.h * im.w * 3 - 3) pos = im.h * im.w * 3 - 3;
    if (pos < 0) pos = 0;  // guard (NaN/negative cannot index; reference would read out of bounds)
    if (face_out) *face_out = face;
    if (pos_out) *pos_out = pos;
    return mk(im.cols[pos], im.cols[pos + 1], im.cols[pos + 2]);
}

// ---- The cube map as a LIGHT (use_envmap == 2).  No counterpart in the reference, which only looks radiance up
// (envMap.cuh:50-89); a build-defined extension like the miss term itself, restated here operation for operation so that the
// product's pictures can be held against it bit for bit.  A texel (face f, row r, column c) stands for the direction through
// its centre; its probability is proportional to average(rgb) x solid angle.  Tables: env_marg[f * H + r] = CDF over all 6 H
// rows, env_cond[(f * H + r) * W + c] = CDF inside a row; sums in double, in index order, stored as float, last entry 1.
inline void env_texel_geometry(int face, int row, int col, int w, int h, V3 &dir, float &solid_angle) {
    const float fu = ((float)col + 0.5f) / (float)w, fv = ((float)row + 0.5f) / (float)h;
    const float a = 2.0f * fu - 1.0f, b = 2.0f * fv - 1.0f;
    const float len2 = 1.0f + a * a + b * b;
    solid_angle = (4.0f / ((float)w * (float)h)) / (len2 * sqrtf(len2));
    // the inverse of env_lookup's index formulas, face by face: env_lookup(dir) lands in this texel again
    switch (face) {
        case 0: dir = mk(1.0f, -b, -a); break;
        case 3: dir = mk(-1.0f, -b, a); break;
        case 1: dir = mk(a, 1.0f, b); break;
        case 4: dir = mk(a, -1.0f, -b); break;
        case 2: dir = mk(a, -b, 1.0f); break;
        default: dir = mk(-a, -b, -1.0f); break;
    }
}
void env_light_build(orc_scene &sc) {
    sc.env_light_ok = false;
    sc.env_marg.clear();
    sc.env_cond.clear();
    if (!sc.has_env) return;
    const int w = sc.env[0].w, h = sc.env[0].h;
    const size_t rows = (size_t)6 * h;
    std::vector<double> row_sum(rows, 0.0);
    sc.env_cond.assign(rows * w, 1.0f);
    sc.env_marg.assign(rows, 1.0f);
    std::vector<double> acc((size_t)w);
    for (size_t R = 0; R < rows; R++) {
        const int face = (int)(R / h), row = (int)(R % h);
        const HdrImage &im = sc.env[face];
        double run = 0.0;
        for (int c = 0; c < w; c++) {
            V3 dir;
            float sa;
            env_texel_geometry(face, row, c, w, h, dir, sa);
            const float *px = &im.cols[((size_t)row * w + c) * 3];
            const double lum = ((double)px[0] + (double)px[1] + (double)px[2]) / 3.0;
            const double wgt = (lum > 0.0 && lum < (double)INFINITY) ? lum * (double)sa : 0.0;  // NaN / negative / inf texels get no probability
            run += wgt;
            acc[(size_t)c] = run;
        }
        row_sum[R] = run;
        if (run > 0.0)
            for (int c = 0; c + 1 < w; c++) sc.env_cond[R * w + c] = (float)(acc[(size_t)c] / run);
    }
    double total = 0.0;
    for (size_t R = 0; R < rows; R++) total += row_sum[R];
    if (!(total > 0.0)) return;  // a black map is no light
    double run = 0.0;
    for (size_t R = 0; R + 1 < rows; R++) {
        run += row_sum[R];
        sc.env_marg[R] = (float)(run / total);
    }
    sc.env_light_ok = true;
}
// first index whose CDF entry is >= u (u in (0, 1], the last entry is 1)
inline int cdf_search(const float *cdf, int n, float u) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] < u) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// One importance sample: the texel, its probability, its direction (not normalised), solid angle and radiance.
inline bool env_light_sample(const orc_scene &sc, float u1, float u2, V3 &dir, float &solid_angle, float &pmf, V3 &radiance) {
    const int w = sc.env[0].w, h = sc.env[0].h;
    const int R = cdf_search(sc.env_marg.data(), 6 * h, u1);
    const float *cond = &sc.env_cond[(size_t)R * w];
    const int c = cdf_search(cond, w, u2);
    const float p_row = sc.env_marg[(size_t)R] - (R > 0 ? sc.env_marg[(size_t)R - 1] : 0.0f);
    const float p_col = cond[c] - (c > 0 ? cond[c - 1] : 0.0f);
    pmf = p_row * p_col;
    const int face = R / h, row = R % h;
    env_texel_geometry(face, row, c, w, h, dir, solid_angle);
    const float *px = &sc.env[face].cols[((size_t)row * w + c) * 3];
    radiance = mk(px[0], px[1], px[2]);
    return pmf > 0.0f;
}

struct LightVertex {
    Hit hit;
    V3 T;
    float pdf = 0.f;
};


// kernel.cu:173-306.  light_depth == 0 selects the build-defined "simple" integrator:
// the camera loop :239-303 without the light sub-path and without the connection block :259-280.
V3 trace_sample(const orc_scene &sc, const orc_render_config &cfg, Ray cam_ray, Rng &rng, Ctr &c) {
    V3 out = mk(0, 0, 0);
    const int Dl = cfg.light_depth, Dc = cfg.cam_depth;
    LightVertex L[MAX_DEPTH];
    int direct_light = 0;  // Q13: always lights[0]
    if (Dl > 0) {
        const int nl = (int)sc.lights.size();
        int li = (int)(rng.u() * ((float)nl));  // :178
        if (li >= nl) li = nl - 1;              // Q17 clamp (documented deviation)
        V3 ldir = light_sample_dir(rng);        // :180
        Hit h = first_intersect(sc, make_ray(sc.lights[li].loc, ldir), c);  // :183-184
        if (h.t < 0 && !MUT(MUT_Q11_KEEP_CAMERA_PATH)) return out;  // :185 (Q11)
        direct_light = MUT(MUT_Q13_SAMPLED_LIGHT) ? li : 0;
        L[0].hit = h;
        if (h.t >= 0) L[0].T = radiance_at(sc.lights[li], h.pos);
        L[0].pdf = (1.0f / (float)nl) * (1.0f / (4.0f * PI));
        for (int i = 0; i < Dl - 1 && h.t >= 0; i++) {  // :197-231
            const Material &m = sc.mats[L[i].hit.mat];
            float pd = average(m.kd), pm = average(m.ks);
            float rnd = rng.u();
            if (rnd >= pd + pm) break;
            V3 in = (i == 0) ? ldir : normalize(L[i].hit.pos - L[i - 1].hit.pos);
            V3 od;
            float pdf_b;
            if (rnd < pd) {
                pdf_b = sample_diffuse(L[i].hit.nrm, od, rng) * pd;
                float ct = dot(L[i].hit.nrm, neg(in));
                if (ct < EPS) break;
                L[i].T = (i == 0) ? L[i].T * m.kd / PI * ct : L[i - 1].T * m.kd / PI * ct;
            } else {
                pdf_b = sample_mirror(L[i].hit.nrm, in, od) * pm;
                float ct = dot(L[i].hit.nrm, neg(in));
                if (ct < EPS) break;
                L[i].T = (i == 0) ? L[i].T * m.ks : L[i - 1].T * m.ks;
            }
            Hit nh = first_intersect(sc, make_ray(L[i].hit.pos + L[i].hit.nrm * EPS, od), c);  // :225-226
            if (nh.t < 0) break;
            L[i + 1].hit = nh;
            L[i + 1].pdf = L[i].pdf * pdf_b;
            if (MUT(MUT_Q14_NO_OVERWRITE)) L[i + 1].T = L[i].T;  // (then :213-222 at i + 1 reads the same value from slot i)
            if (g_probe && L[i + 1].pdf == 0.0f) g_probe->zero_pdf_vertices++;
        }
    }
    V3 Tc = mk(1, 1, 1);
    float pc = 1.0f;
    // use_envmap == 2 (build-defined): the map is ALSO sampled as a light at every camera vertex; a ray that leaves the scene
    // then only collects the map where the sampling cannot have: at the first vertex and behind a mirror bounce
    const bool env_light = cfg.use_envmap == 2 && sc.has_env && sc.env_light_ok;
    bool after_mirror = false;
    for (int i = 0; i < Dc; i++) {  // :239-303
        Hit hit = first_intersect(sc, cam_ray, c);
        if (hit.t < 0) {
            if (cfg.use_envmap && sc.has_env && (!env_light || i == 0 || after_mirror))
                out = out + (Tc / pc) * env_lookup(sc, cam_ray.d);  // build-defined
            break;
        }
        const Material &m = sc.mats[hit.mat];
        {  // direct connection to lights[0] (Q13), shadow ray cast FROM the light (Q10)  :244-257
            V3 lp = sc.lights[direct_light].loc;
            V3 ld = normalize(hit.pos - lp);
            bool visible;
            if (MUT(MUT_Q10_SHADOW_FROM_HIT)) {  // from the surface towards the light: nothing in front of the light
                Hit sh = first_intersect(sc, make_ray(hit.pos + hit.nrm * EPS, lp - hit.pos), c);
                visible = sh.t < 0 || sh.t > length(lp - hit.pos) - EPS;
            } else {
                Hit sh = first_intersect(sc, make_ray(lp, ld), c);
                visible = sh.t > 0 && length(sh.pos - hit.pos) < EPS;
            }
            if (visible) {
                V3 lr = radiance_at(sc.lights[direct_light], hit.pos);
                float ct = dot(hit.nrm, neg(ld));
                if (ct > EPS) {
                    V3 brdf = m.kd / PI;
                    if (g_probe && !g_probe->first_cause && !finite3((Tc / pc) * (lr * brdf * ct))) {
                        g_probe->first_cause = ORC_NF_DIRECT;  // kernel.cu:254: cam_throughput / cam_pdf
                        g_probe->cam_vertex = i;
                        g_probe->operand[0] = Tc.x; g_probe->operand[1] = pc; g_probe->operand[2] = lr.x; g_probe->operand[3] = ct;
                    }
                    out = out + (Tc / pc) * (lr * brdf * ct);
                }
            }
        }
        for (int j = 0; j < Dl; j++) {  // :260-280 (Q15 weights, no MIS)
            if (L[j].hit.t < 0) break;
            V3 dir = normalize(L[j].hit.pos - hit.pos);
            float dist = length(L[j].hit.pos - hit.pos);
            Hit sh = first_intersect(sc, make_ray(hit.pos + hit.nrm * EPS, dir), c);
            if (sh.t < 0 || length(sh.pos - L[j].hit.pos) > EPS) continue;
            float cc = dot(hit.nrm, dir), cl = dot(L[j].hit.nrm, neg(dir));
            if (cc < EPS || cl < EPS) continue;
            float G = cc * cl / (dist * dist);
            V3 brdf = m.kd / PI;
            V3 cw = MUT(MUT_Q15_SINGLE_COS) ? Tc * brdf / pc : Tc * brdf * cc / pc;
            V3 lw = L[j].T / L[j].pdf;
            if (g_probe && !g_probe->first_cause && !finite3(lw * G * cw)) {
                // kernel.cu:277 light_throughput[j] / light_pdf[j]; :274 G = cosCam cosLight / dist^2; :276 camera_weight
                g_probe->first_cause = !finite3(lw) ? ORC_NF_LIGHT_WEIGHT : (!std::isfinite(G) ? ORC_NF_GEOMETRY : (!finite3(cw) ? ORC_NF_CAMERA_WEIGHT : ORC_NF_PRODUCT));
                g_probe->light_vertex = j;
                g_probe->cam_vertex = i;
                g_probe->operand[0] = L[j].T.x; g_probe->operand[1] = L[j].pdf; g_probe->operand[2] = G; g_probe->operand[3] = cw.x;
            }
            out = out + lw * G * cw;
        }
        if (env_light) {  // two draws, always; the shadow ray must leave the scene
            const float u1 = rng.u(), u2 = rng.u();
            V3 edir, erad;
            float sa, pmf;
            if (env_light_sample(sc, u1, u2, edir, sa, pmf, erad)) {
                const V3 wdir = normalize(edir);
                const float ct = dot(hit.nrm, wdir);
                if (ct > EPS) {
                    Hit sh = first_intersect(sc, make_ray(hit.pos + hit.nrm * EPS, wdir), c);
                    if (sh.t < 0) {
                        V3 brdf = m.kd / PI;
                        out = out + ((Tc / pc) * (erad * brdf * ct)) * (sa / pmf);
                    }
                }
            }
        }
        float pd = average(m.kd), pm = average(m.ks);  // :283-300 (Q16)
        float rnd = rng.u();
        V3 od;
        if (rnd < pd) {
            float pdf = sample_diffuse(hit.nrm, od, rng);
            float ct = dot(hit.nrm, od);
            if (ct < EPS) break;
            Tc = Tc * m.kd / PI * ct;
            pc = MUT(MUT_Q16_ROULETTE_PDF) ? pc * pdf : pc * pdf * pd;
            after_mirror = false;
        } else if (rnd < pd + pm) {
            float pdf = sample_mirror(hit.nrm, cam_ray.d, od);
            Tc = Tc * m.ks;
            pc = MUT(MUT_Q16_ROULETTE_PDF) ? pc * pdf : pc * pdf * pm;
            after_mirror = true;
        } else {
            break;
        }
        cam_ray = make_ray(hit.pos + hit.nrm * EPS, od);  // :302
    }
    return out;
}

int fail(const char *msg) {
    g_err = msg;
    return -1;
}
V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
void export_ctr(const Ctr &c, orc_counters *o) {
    if (!o) return;
    o->rays = c.rays; o->nodes = c.nodes; o->pops = c.pops; o->refs = c.refs; o->tris = c.tris; o->hits = c.hits;
    o->box_tests = c.box_tests; o->stack_overflow = c.overflow; o->max_stack = c.max_stack;
}

// mesh.cu:232-317: OBJ v / vn / f (p//n or p/t/n), SRT baked in, y/z swap through vec4(vec3,w) (Q2)
int read_obj(const char *path, const M4 &srt, std::vector<Tri> &out) {
    std::ifstream in(path);
    if (!in.is_open()) return fail("oracle: cannot open OBJ file");
    std::vector<V3> P, Nn;
    std::vector<int> pi, ni;
    std::string line, prefix;
    std::stringstream ss;
    while (std::getline(in, line)) {
        ss.clear();
        ss.str(line);
        prefix.clear();
        ss >> prefix;
        if (prefix == "v") {
            V3 t; ss >> t.x >> t.y >> t.z; P.push_back(t);
        } else if (prefix == "vn") {
            V3 t; ss >> t.x >> t.y >> t.z; Nn.push_back(t);
        } else if (prefix == "f") {
            int counter = 0, i;
            while (ss >> i) {  // :272-291
                if (counter == 0) pi.push_back(i - 1);
                else if (counter == 2) ni.push_back(i - 1);
                for (int k = 0; k < 3; k++)
                    if (ss.peek() == '/') { counter++; ss.ignore(1, '/'); }
                if (ss.peek() == ' ') { counter = 0; ss.ignore(1, ' '); }
            }
        }
    }
    if (pi.size() != ni.size() || pi.size() % 3 != 0) return fail("oracle: OBJ faces must be triangles with normals");
    M4 nm = transpose_m(inverse_m(srt));
    auto xp = [&](int idx) { V4 r = mul(from_v3_swapped(P[idx], 1), srt); return mk(r.v[0], r.v[1], r.v[2]); };
    auto xn = [&](int idx) { V4 r = mul(from_v3_swapped(Nn[idx], 0), nm); return normalize(mk(r.v[0], r.v[1], r.v[2])); };
    for (size_t k = 0; k < pi.size(); k++)
        if (pi[k] < 0 || pi[k] >= (int)P.size() || ni[k] < 0 || ni[k] >= (int)Nn.size()) return fail("oracle: OBJ index out of range");
    for (size_t f = 0; f < pi.size() / 3; f++) {  // :298-315
        Tri t;
        t.pa = xp(pi[3 * f]); t.na = xn(ni[3 * f]);
        t.pb = xp(pi[3 * f + 1]); t.nb = xn(ni[3 * f + 1]);
        t.pc = xp(pi[3 * f + 2]); t.nc = xn(ni[3 * f + 2]);
        out.push_back(t);
    }
    return 0;
}
}  // namespace

// ================================================================= C interface
extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }
// Mutation mask (bit i = MUT_* above).  Only the -DORC_MUTATIONS build honours it; the checker itself refuses a non-zero mask.
int orc_set_mutations(uint32_t mask) {
#ifdef ORC_MUTATIONS
    if (mask >> MUT_COUNT) return fail("unknown mutation bit");
    g_mut = mask;
    return 0;
#else
    return mask == 0u ? 0 : fail("this build of the oracle has no mutation switches (build liboracle_mut.so)");
#endif
}
int orc_mutation_count(void) { return MUT_COUNT; }
orc_scene *orc_scene_create(void) { return new orc_scene(); }
void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_scene_set_camera(orc_scene *s, const float eye[3], const float lookat[3], const float vup[3], float fov) {
    camera_set(s->cam, ld3(eye), ld3(lookat), ld3(vup), fov);
    return 0;
}
int orc_scene_add_material(orc_scene *s, const float kd[3], const float ks[3]) {
    Material m; m.kd = ld3(kd); m.ks = ld3(ks);
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
int orc_scene_add_light(orc_scene *s, const float loc[3], const float power[3]) {
    Light l; l.loc = ld3(loc); l.power = ld3(power);
    s->lights.push_back(l);
    return (int)s->lights.size() - 1;
}
int orc_scene_add_plane(orc_scene *s, const float point[3], const float normal[3], int mat) {
    if (mat < 0 || mat >= (int)s->mats.size()) return fail("oracle: bad material id");
    Plane p; p.point = ld3(point); p.normal = normalize(ld3(normal));  // plane.cuh:16
    p.mat = mat;
    s->planes.push_back(p);
    return (int)s->planes.size() - 1;
}
int orc_scene_add_sphere(orc_scene *s, const float center[3], float radius, int mat) {
    if (mat < 0 || mat >= (int)s->mats.size()) return fail("oracle: bad material id");
    Sphere sp; sp.center = ld3(center); sp.radius = radius; sp.mat = mat;
    s->spheres.push_back(sp);
    return (int)s->spheres.size() - 1;
}
int orc_scene_add_mesh(orc_scene *s, const float *pos, const float *nrm, int64_t n, int mat) {
    if (mat < 0 || mat >= (int)s->mats.size()) return fail("oracle: bad material id");
    if (n <= 0) return fail("oracle: empty mesh");
    Mesh m;
    m.mat = mat;
    m.tris.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        Tri &t = m.tris[(size_t)i];
        t.pa = ld3(pos + 9 * i); t.pb = ld3(pos + 9 * i + 3); t.pc = ld3(pos + 9 * i + 6);
        t.na = ld3(nrm + 9 * i); t.nb = ld3(nrm + 9 * i + 3); t.nc = ld3(nrm + 9 * i + 6);
    }
    s->meshes.push_back(std::move(m));
    s->committed = false;
    return (int)s->meshes.size() - 1;
}
int orc_scene_add_mesh_obj(orc_scene *s, const char *path, const float position[3], const float rotate[3],
                           const float scale[3], int mat) {
    if (mat < 0 || mat >= (int)s->mats.size()) return fail("oracle: bad material id");
    Mesh m;
    m.mat = mat;
    M4 srt = srt_m(ld3(scale), ld3(rotate), ld3(position));  // mesh.cuh:323
    if (read_obj(path, srt, m.tris) != 0) return -1;
    if (m.tris.empty()) return fail("oracle: OBJ has no faces");
    s->meshes.push_back(std::move(m));
    s->committed = false;
    return (int)s->meshes.size() - 1;
}
int orc_scene_set_envmap_rgbe_dir(orc_scene *s, const char *folder) {  // envMap.cuh:34-40
    const char *names[6] = {"px.hdr", "py.hdr", "pz.hdr", "nx.hdr", "ny.hdr", "nz.hdr"};
    for (int i = 0; i < 6; i++) {
        std::string p = std::string(folder) + names[i];
        if (!hdr_load(p.c_str(), s->env[i])) return fail("oracle: cannot load HDR face");
    }
    s->has_env = true;
    return 0;
}
int orc_scene_set_envmap_f32(orc_scene *s, const float *faces6, int w, int h) {
    for (int i = 0; i < 6; i++) {
        s->env[i].w = w; s->env[i].h = h;
        s->env[i].cols.assign(faces6 + (size_t)i * w * h * 3, faces6 + (size_t)(i + 1) * w * h * 3);
    }
    s->has_env = true;
    return 0;
}
int orc_scene_commit(orc_scene *s, int n_threads) {
    if (n_threads > 0) omp_set_num_threads(n_threads);
    int base = 0;
    for (Mesh &m : s->meshes) {
        m.tri_base = base;
        base += (int)m.tris.size();
        if (m.nodes.empty()) build_mesh_tree(m);
    }
    s->total_tris = base;
    env_light_build(*s);
    s->committed = true;
    return 0;
}

int orc_trace_closest(orc_scene *s, const float *ro, const float *rd, int64_t n, int32_t *prim, float *t,
                      orc_counters *ctr, int brute, int n_threads) {
    if (!s->committed) return fail("oracle: scene not committed");
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    Ctr total;
#pragma omp parallel num_threads(n_threads)
    {
        Ctr c;
#pragma omp for schedule(dynamic, 1024)
        for (int64_t i = 0; i < n; i++) {
            Ray r = make_ray(ld3(ro + 3 * i), ld3(rd + 3 * i));
            Hit h = first_intersect(*s, r, c, brute != 0);
            prim[i] = h.prim;
            t[i] = h.t;
        }
#pragma omp critical
        total.add(c);
    }
    export_ctr(total, ctr);
    return 0;
}

int orc_render(orc_scene *s, const orc_render_config *cfg, float *out, orc_counters *ctr, int n_threads) {
    if (!s->committed) return fail("oracle: scene not committed");
    if (!s->cam.set) return fail("oracle: camera not set");
    if (s->lights.empty()) return fail("oracle: at least one light required (kernel.cu:244)");
    if (cfg->cam_depth < 0 || cfg->cam_depth > MAX_DEPTH || cfg->light_depth < 0 || cfg->light_depth > MAX_DEPTH)
        return fail("oracle: depth out of range");
    if (cfg->use_envmap && !s->has_env) return fail("oracle: use_envmap without an env map");
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    const int W = cfg->width, H = cfg->height;
    const int stride = cfg->sample_stride > 0 ? cfg->sample_stride : 1;
    const int pstride = cfg->pixel_stride > 0 ? cfg->pixel_stride : 1;
    std::memset(out, 0, sizeof(float) * 3 * (size_t)W * H);
    Ctr total;
#pragma omp parallel num_threads(n_threads)
    {
        Ctr c;
#pragma omp for schedule(dynamic, 64)
        for (int64_t id = 0; id < (int64_t)W * H; id += pstride) {
            int row = (int)(id / W), col = (int)(id % W);
            V3 acc = mk(0, 0, 0);
            for (int smp = cfg->sample_begin; smp < cfg->sample_end; smp += stride) {  // kernel.cu:338-344
                Rng rng(cfg->seed, (uint32_t)id, (uint32_t)smp);
                float jx = rng.u();  // Q18: first draw -> X, second -> Y
                float jy = rng.u();
                Ray r = camera_ray(s->cam, (float)col + jx, (float)row + jy, W, H);
                V3 L = trace_sample(*s, *cfg, r, rng, c);
                acc = acc + L / (float)cfg->spp;  // :343
            }
            out[3 * id] = acc.x; out[3 * id + 1] = acc.y; out[3 * id + 2] = acc.z;
        }
#pragma omp critical
        total.add(c);
    }
    export_ctr(total, ctr);
    return 0;
}

// One (pixel, sample) of orc_render with the probe armed: rgb = the sample's radiance (NOT divided by spp).
int orc_explain_sample(orc_scene *s, const orc_render_config *cfg, int64_t pixel, int32_t sample, float rgb[3], orc_sample_probe *out) {
    if (!s->committed || !s->cam.set || s->lights.empty()) return fail("oracle: scene not ready");
    const int W = cfg->width, H = cfg->height;
    if (pixel < 0 || pixel >= (int64_t)W * H) return fail("oracle: pixel out of range");
    Probe pr;
    Ctr c;
    g_probe = &pr;
    Rng rng(cfg->seed, (uint32_t)pixel, (uint32_t)sample);
    float jx = rng.u();
    float jy = rng.u();
    Ray r = camera_ray(s->cam, (float)(pixel % W) + jx, (float)(pixel / W) + jy, W, H);
    V3 L = trace_sample(*s, *cfg, r, rng, c);
    g_probe = nullptr;
    rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    if (out) {
        out->first_cause = pr.first_cause; out->light_vertex = pr.light_vertex; out->cam_vertex = pr.cam_vertex;
        out->sqrt_negative = pr.sqrt_negative; out->zero_pdf_vertices = pr.zero_pdf_vertices; out->rng_draws = (int32_t)rng.draw;
        for (int k = 0; k < 4; k++) out->operand[k] = pr.operand[k];
    }
    return 0;
}

int orc_mesh_count(orc_scene *s) { return (int)s->meshes.size(); }
int orc_kd_sizes(orc_scene *s, int mesh, int64_t *nn, int64_t *nr, int64_t *nt) {
    if (mesh < 0 || mesh >= (int)s->meshes.size() || !s->committed) return fail("oracle: bad mesh / not committed");
    const Mesh &m = s->meshes[mesh];
    if (nn) *nn = (int64_t)m.nodes.size();
    if (nr) *nr = (int64_t)m.refs.size();
    if (nt) *nt = (int64_t)m.tris.size();
    return 0;
}
int orc_kd_export(orc_scene *s, int mesh, float *boxes, int32_t *axis, int32_t *left, int32_t *right, int32_t *first,
                  int32_t *count, int32_t *refs) {
    if (mesh < 0 || mesh >= (int)s->meshes.size() || !s->committed) return fail("oracle: bad mesh / not committed");
    const Mesh &m = s->meshes[mesh];
    for (size_t i = 0; i < m.nodes.size(); i++) {
        const FlatNode &n = m.nodes[i];
        if (boxes) {
            boxes[6 * i + 0] = n.box.lo[0]; boxes[6 * i + 1] = n.box.hi[0]; boxes[6 * i + 2] = n.box.lo[1];
            boxes[6 * i + 3] = n.box.hi[1]; boxes[6 * i + 4] = n.box.lo[2]; boxes[6 * i + 5] = n.box.hi[2];
        }
        if (axis) axis[i] = n.axis;
        if (left) left[i] = n.left;
        if (right) right[i] = n.right;
        if (first) first[i] = n.first;
        if (count) count[i] = n.left < 0 ? n.count : 0;
    }
    if (refs) std::copy(m.refs.begin(), m.refs.end(), refs);
    return 0;
}
int orc_mesh_triangles(orc_scene *s, int mesh, float *pos9, float *nrm9) {
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return fail("oracle: bad mesh");
    const Mesh &m = s->meshes[mesh];
    for (size_t i = 0; i < m.tris.size(); i++) {
        const Tri &t = m.tris[i];
        const V3 p[3] = {t.pa, t.pb, t.pc}, n[3] = {t.na, t.nb, t.nc};
        for (int k = 0; k < 3; k++) {
            if (pos9) { pos9[9 * i + 3 * k] = p[k].x; pos9[9 * i + 3 * k + 1] = p[k].y; pos9[9 * i + 3 * k + 2] = p[k].z; }
            if (nrm9) { nrm9[9 * i + 3 * k] = n[k].x; nrm9[9 * i + 3 * k + 1] = n[k].y; nrm9[9 * i + 3 * k + 2] = n[k].z; }
        }
    }
    return 0;
}

void orc_normalize(const float v[3], float out[3]) {
    V3 r = normalize(ld3(v));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
static Ray unit_ray(const float o[3], const float d[3]) { Ray r; r.o = ld3(o); r.d = ld3(d); return r; }
float orc_triangle_intersect(const float p[9], const float o[3], const float d[3]) {
    Tri t; t.pa = ld3(p); t.pb = ld3(p + 3); t.pc = ld3(p + 6);
    return tri_intersect_t(t, unit_ray(o, d));
}
static Box ld_box(const float b[6]) {
    Box x; x.lo[0] = b[0]; x.hi[0] = b[1]; x.lo[1] = b[2]; x.hi[1] = b[3]; x.lo[2] = b[4]; x.hi[2] = b[5];
    return x;
}
float orc_box_intersect(const float b[6], const float o[3], const float d[3]) { return box_intersect(ld_box(b), unit_ray(o, d)); }
int orc_box_contains(const float b[6], const float p[3]) { return box_contains(ld_box(b), ld3(p)) ? 1 : 0; }
float orc_plane_intersect(const float point[3], const float n[3], const float o[3], const float d[3]) {
    Plane p; p.point = ld3(point); p.normal = ld3(n); p.mat = 0;
    return plane_intersect(p, unit_ray(o, d)).t;
}
float orc_sphere_intersect(const float c[3], float radius, const float o[3], const float d[3]) {
    Sphere s; s.center = ld3(c); s.radius = radius; s.mat = 0;
    return sphere_intersect(s, unit_ray(o, d)).t;
}
void orc_camera_ray(orc_scene *s, float X, float Y, int W, int H, float o[3], float d[3]) {
    Ray r = camera_ray(s->cam, X, Y, W, H);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
void orc_sincos(float x, float *s, float *c) { *s = fixed_sinf(x); *c = fixed_cosf(x); }
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
float orc_rng_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t draw) {
    Rng r(seed, pixel, sample);
    r.draw = draw;
    return r.u();
}
// The sampling tables of the cube map as a light (after commit): marg[6 H], cond[6 H W]; returns 1 if the map can be sampled
// (it exists and is not black), 0 if not, < 0 on error.  Null pointers: only the answer.
int orc_env_light_tables(orc_scene *s, float *marg, float *cond) {
    if (!s->committed) return fail("oracle: scene not committed");
    if (!s->has_env) return 0;
    if (marg) memcpy(marg, s->env_marg.data(), s->env_marg.size() * sizeof(float));
    if (cond) memcpy(cond, s->env_cond.data(), s->env_cond.size() * sizeof(float));
    return s->env_light_ok ? 1 : 0;
}
// The texel a pair of uniform numbers selects: face, row, column, probability, solid angle, direction (not normalised).
int orc_env_light_sample(orc_scene *s, float u1, float u2, int32_t *frc, float *pmf_sa, float *dir) {
    if (!s->committed || !s->env_light_ok) return fail("oracle: no sampled env map");
    V3 d, rad;
    float sa, pmf;
    env_light_sample(*s, u1, u2, d, sa, pmf, rad);
    const int w = s->env[0].w, h = s->env[0].h;
    const int R = cdf_search(s->env_marg.data(), 6 * h, u1);
    const int c = cdf_search(&s->env_cond[(size_t)R * w], w, u2);
    frc[0] = R / h; frc[1] = R % h; frc[2] = c;
    pmf_sa[0] = pmf; pmf_sa[1] = sa;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
    return 0;
}
int orc_env_lookup(orc_scene *s, const float *dirs, int64_t n, float *rgb, int32_t *face, int32_t *pos) {
    if (!s->has_env) return fail("oracle: no env map");
    for (int64_t i = 0; i < n; i++) {
        int f, p;
        V3 c = env_lookup(*s, ld3(dirs + 3 * i), &f, &p);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        if (face) face[i] = f;
        if (pos) pos[i] = p;
    }
    return 0;
}
int orc_hdr_load(const char *path, int *w, int *h, float *cols) {
    HdrImage im;
    if (!hdr_load(path, im)) return fail("oracle: cannot load HDR");
    *w = im.w; *h = im.h;
    if (cols) std::copy(im.cols.begin(), im.cols.end(), cols);
    return 0;
}
void orc_srt_matrix(const float scale[3], const float rotate[3], const float translate[3], float m16[16], float it16[16]) {
    M4 m = srt_m(ld3(scale), ld3(rotate), ld3(translate));
    M4 it = transpose_m(inverse_m(m));
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { m16[4 * i + j] = m.r[i].v[j]; it16[4 * i + j] = it.r[i].v[j]; }
}
}  // extern "C"
