// kernels_extend.h -- the `extend` stage: kd-tree closest hit over a ray queue (gfx950, wave64).
//
// One lane = one ray (T3/T4 of SURVEY.md 8a).  Rays differ wildly in cost (a ray of the 1.3M-triangle test mesh
// visits 1..40 leaves), so the kernel is PERSISTENT: the grid is exactly the number of resident waves, every
// lane is a small state machine (start mesh -> descend -> leaf -> pop / next mesh -> finish), and a wave
// refills its idle lanes from a global work counter (ballot + mbcnt rank + one atomic per refill) as soon as
// enough of them have finished.  Lanes re-converge after at most `step_cap` node steps and at every leaf, so a
// lane never waits for a neighbour's whole ray, only for its current few steps / leaf scan.
// The traversal stack is an LDS array laid out [slot][lane]: a wave's 64 lanes hit 64 consecutive banks.
// The ray count is read from device memory (no host round trip between bounces of the wavefront tracer).
//
// Work cursors: ONE word pulled by 5,120 waves saturates (a same-address atomic completes at ~88 M/s on MI355X and the
// 1080p frame asked for 72 M/s), so the queue is cut into 8 contiguous ranges with a cursor each, 256 B apart; the waves
// of workgroup b start on range b mod 8 (workgroups go round the 8 XCDs, so one XCD's L2 sees one eighth of a coherent
// queue instead of all of it) and move on to the next range when theirs runs dry, until all eight are.  Which range a
// wave pulls from changes no result.
#pragma once
#include <hip/hip_runtime.h>
#include "device_ctx.h"
#include "traverse.h"

namespace mipt {

constexpr int kExtendBlock = 256;

struct LdsStack {
    int *col;  // &stack[threadIdx.x]
    __device__ __forceinline__ void push(int sp, int v) { col[sp * kExtendBlock] = v; }
    __device__ __forceinline__ int pop(int sp) { return col[sp * kExtendBlock]; }
#if 1
    // Rows above the stack: per-ray values that are written when the ray is fetched and read when it is finished (the sphere /
    // plane answer it arrived with, the best mesh triangle so far).  In registers they were what the compiler spilled to
    // scratch MEMORY -- and every `s_waitcnt vmcnt(0)` behind such a spill store waits for its acknowledgement.
    __device__ __forceinline__ void cold_set(int k, int v) { col[(kStackDepth + k) * kExtendBlock] = v; }
    __device__ __forceinline__ int cold_get(int k) const { return col[(kStackDepth + k) * kExtendBlock]; }
#endif
};
constexpr int kColdRows = 3;  // q_t, q_prim, prim

__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// kernel variants: the production walk; the reference's walk node for node with its counters (SURVEY 8d's units); the
// production walk that also counts what it reads (product-layout algorithmic bytes)
constexpr int kProd = 0, kCountRef = 1, kCountWork = 2;
constexpr int kCursorStride = 64;  // ints between two cursors (256 B)

__device__ __forceinline__ void flush_work_counters(const WorkCounters &wc, unsigned long long *ctr) {
    const unsigned int v[6] = {wave_sum(wc.steps), wave_sum(wc.pops), wave_sum(wc.sphere_pairs), wave_sum(wc.planes),
                               wave_sum(wc.verts), wave_sum(wc.leaves)};
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&ctr[i], (unsigned long long)v[i]);
}

__device__ __forceinline__ void flush_counters(const TravCounters &tc, unsigned int hit_tris, unsigned long long *ctr) {
    const unsigned int a = wave_sum(tc.nodes), b = wave_sum(tc.pops), c = wave_sum(tc.refs), d = wave_sum(tc.tris),
                       e = wave_sum(hit_tris), f = wave_sum(tc.culled);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr[0], (unsigned long long)a);
        atomicAdd(&ctr[1], (unsigned long long)b);
        atomicAdd(&ctr[2], (unsigned long long)c);
        atomicAdd(&ctr[3], (unsigned long long)d);
        atomicAdd(&ctr[4], (unsigned long long)e);
        atomicAdd(&ctr[5], (unsigned long long)f);
    }
}

// ---- leaf scan with a wave-cooperative second pass ------------------------------------------------------------------
// leaf_scan() (traverse.h) lets every lane evaluate its own marked candidates one after another: the wave runs that ~100-
// instruction body as often as its busiest lane has marks, with a third (C2) to an eighth (incoherent batches) of the
// lanes in it.  Here the marks of all lanes waiting at a leaf are pooled instead: each owner writes (owner lane, reference)
// items into a per-wave LDS list at its prefix-sum offset, the 64 lanes take the items 64 at a time -- the owner's ray and
// leaf box come over by ds_bpermute -- and a passed candidate goes back as an LDS atomicMin on the key (bits(t) << 32 |
// reference): the smallest t, and among equal t the earliest reference, which is what the reference's sequential
// "t < best" rule keeps (mesh.cuh:293-297; t > 0, so its bits order like the value).  Must be called by all 64 lanes.
constexpr int kCoopItems = 128;  // items per round and wave (more marks wait for the next round)

__device__ __forceinline__ float plane_candidate_nobest(f4 pl, f3 o, f3 d, const float lo[3], const float hi[3]) {
    const f3 N = mk3(pl.x, pl.y, pl.z);
    const float nd = dot3(N, d);
    if (fabsf(nd) < kEps) return -1.0f;  // Q3
    const float t = -(dot3(N, o) + pl.w) / nd;
    if (!(t > 0.0f)) return -1.0f;
    const f3 P = ray_at(o, d, t);
    if (!point_in_box(P, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])) return -1.0f;
    return t;
}

// wave-wide inclusive prefix sum of a small per-lane count
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    return v;
}

template <bool WORK>
__device__ __forceinline__ void leaf_scan_coop(const DScene &sc, bool pending, int first, int count, f3 o, f3 d, const SlabBox &cur,
                                               float &bt, int &bi, WorkCounters &wc, int *block_items, unsigned long long *block_res,
                                               bool pool_pass1 PSTAT_ARG) {
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: the per-wave LDS bases cost no VGPR
    int *const items = block_items + wave * 2 * kCoopItems;
    unsigned long long *const res = block_res + wave * 64;
    const int end = first + count;
    unsigned long long best_key = ~0ull;
    if (WORK && pending) wc.leaves++;
    for (int base = first & ~1;; base += 64) {
        const bool active = pending && base < end;
        if (__ballot(active) == 0ull) break;
        // ---- pass 1: the candidate spheres.  With most lanes of the wave waiting at a leaf every lane streams its own leaf
        // (as leaf_scan() does); with few of them (pool_pass1, wave-uniform) the work is pooled: a unit = 4 candidates = two
        // sphere pairs (64 B), the owners list their units, the 64 lanes take them 64 at a time (owner's ray by ds_bpermute)
        // and OR the marks into the owner's 64-bit mask in LDS.
        const int stop = active ? ((end - base < 64) ? end - base : 64) : 0;
        unsigned long long marked = 0ull;
        if (!pool_pass1) {
            if (active) {
                const SpherePair *sp = sc.leaf_spheres + (base >> 1);
                for (int p = 0; 2 * p < stop; p += 2) {
                    PSTAT(ps, 5);
                    const SpherePair s0 = sp[p], s1 = sp[p + 1];
                    if (WORK) wc.sphere_pairs += 2;
                    marked |= (unsigned long long)(sphere_pair_hit(s0, o, d) | (sphere_pair_hit(s1, o, d) << 2)) << (2 * p);
                }
            }
        } else {
            const int n_units = (stop + 3) >> 2;
            int next_unit = 0;
            res[lane] = 0ull;
            while (__ballot(next_unit < n_units) != 0ull) {
                const int c = n_units - next_unit;
                const int incl = wave_incl_scan(c, lane);
                const int total = __shfl(incl, 63, 64);
                for (int pos = incl - c; next_unit < n_units && pos < 2 * kCoopItems; pos++, next_unit++) items[pos] = lane | (next_unit << 8);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int T = total < 2 * kCoopItems ? total : 2 * kCoopItems;
                for (int i0 = 0; i0 < T; i0 += 64) {
                    const int i = i0 + lane;
                    const bool valid = i < T;
                    const int it = valid ? items[i] : lane;
                    const int p = it & 63, u = it >> 8;
                    const f3 fo = mk3(__shfl(o.x, p, 64), __shfl(o.y, p, 64), __shfl(o.z, p, 64));
                    const f3 fd = mk3(__shfl(d.x, p, 64), __shfl(d.y, p, 64), __shfl(d.z, p, 64));
                    const int ffirst = __shfl(first, p, 64);
                    if (valid) {
                        PSTAT(ps, 5);
                        const SpherePair *sp = sc.leaf_spheres + (((ffirst & ~1) + (base - (first & ~1))) >> 1) + 2 * u;
                        const SpherePair s0 = sp[0], s1 = sp[1];
                        if (WORK) wc.sphere_pairs += 2;
                        const unsigned long long m4 = (unsigned long long)(sphere_pair_hit(s0, fo, fd) | (sphere_pair_hit(s1, fo, fd) << 2));
                        if (m4 != 0ull) atomicOr(&res[p], m4 << (4 * u));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            marked = res[lane];
            __builtin_amdgcn_wave_barrier();
        }
        marked &= (stop < 64 ? (1ull << stop) - 1ull : ~0ull) & ~((1ull << (first > base ? first - base : 0)) - 1ull);
        // ---- pass 2, pooled: items = (owner, reference)
        while (__ballot(marked != 0ull) != 0ull) {
            const int c = __popcll(marked);
            const int incl = wave_incl_scan(c, lane);
            const int total = __shfl(incl, 63, 64);
            res[lane] = ~0ull;
            for (int pos = incl - c; marked != 0ull && pos < kCoopItems; pos++) {
                const int bit = __builtin_ffsll((long long)marked) - 1;
                marked &= marked - 1ull;
                items[pos] = lane;
                items[kCoopItems + pos] = base + bit;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int T = total < kCoopItems ? total : kCoopItems;
            for (int i0 = 0; i0 < T; i0 += 64) {  // wave-uniform trip count: every lane takes part in the shuffles
                const int i = i0 + lane;
                const bool valid = i < T;
                const int p = valid ? items[i] : lane;
                const int r = valid ? items[kCoopItems + i] : 0;
                const f3 fo = mk3(__shfl(o.x, p, 64), __shfl(o.y, p, 64), __shfl(o.z, p, 64));
                const f3 fd = mk3(__shfl(d.x, p, 64), __shfl(d.y, p, 64), __shfl(d.z, p, 64));
                const float flo[3] = {__shfl(cur.lo[0], p, 64), __shfl(cur.lo[1], p, 64), __shfl(cur.lo[2], p, 64)};
                const float fhi[3] = {__shfl(cur.hi[0], p, 64), __shfl(cur.hi[1], p, 64), __shfl(cur.hi[2], p, 64)};
                if (valid) {
                    PSTAT(ps, 6);
                    const f4 *rec = sc.leaf_recs + 4 * (size_t)r;
                    const f4 pl = rec[0];
                    const float t = plane_candidate_nobest(pl, fo, fd, flo, fhi);
                    if (WORK) wc.planes++;
                    if (t > 0.0f) {
                        if (WORK) wc.verts++;
                        // (loading the whole 64-byte record before the plane test -- one round trip instead of two -- measured:
                        //  coherent closest hit -2 %, the rest unchanged; 42 % of the marked candidates fail the plane test)
                        const f4 pa = rec[1], pb = rec[2], pc = rec[3];
                        if (inside_triangle(pl, mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z), mk3(pc.x, pc.y, pc.z), fo, fd, t))
                            atomicMin(&res[p], ((unsigned long long)as_u32(t) << 32) | (unsigned long long)(unsigned int)r);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long k = res[lane];
            if (pending && k < best_key) best_key = k;
        }
    }
    bt = -1.0f;
    bi = -1;
    if (pending && best_key != ~0ull) {
        bt = as_f32((uint32_t)(best_key >> 32));
#ifdef MIPT_TRI_FROM_REFS
        bi = sc.refs[(int)(uint32_t)best_key];  // reference position -> triangle id, once per leaf
#else
        // reference position -> triangle id, once per leaf: it rides in a.w of the leaf record, a line the winning test has
        // just touched (`refs` is a cold array: this load used to go to HBM)
        bi = (int)as_u32(sc.leaf_recs[4 * (size_t)(uint32_t)best_key + 1].w);
#endif
    }
}

// rays: o/d as float4 (w not read here), d already normalised by the Ray constructor rule (ray.cuh:10).
// hits: on entry the sphere / plane answer of every ray (quadrics_closest, written by the stage that made the ray),
//       on exit the closest hit of the scene.
// queue: optional indirection (slot ids); count_ptr: optional device-side ray count; work: zeroed counter.
// tune.common_origin != 0: every ray of the launch starts at the eye (1: primary rays) or at lights[0] (2: direct-
// connection shadow rays); ray_o is not read -- the stage that made the rays does not write 16 bytes per ray either.
template <int MODE>
#ifndef MIPT_EXTEND_WAVES
#define MIPT_EXTEND_WAVES 5  // waves per SIMD the register budget is cut for (5: 96 VGPRs; 4: 128 -- measured slower)
#endif
__global__ __launch_bounds__(kExtendBlock, MIPT_EXTEND_WAVES) void k_extend(DScene sc, const f4 *__restrict__ ray_o,
                                                         const f4 *__restrict__ ray_d, const int *__restrict__ queue,
                                                         const int *__restrict__ count_ptr, int n_fixed,
                                                         HitRec *__restrict__ hits, int *__restrict__ work,
                                                         unsigned long long *ctr, ExtendTuning tune, ExtendTuning tune_if_coherent,
                                                         const int *__restrict__ probe) {
    constexpr bool COUNT = MODE == kCountRef;
    if (probe) {  // closest-hit batch: k_make_rays sampled the order the rays arrive in (wave-uniform choice)
        const int sampled = probe[0], coherent = probe[1];
        if (sampled > 0 && 10 * (long long)coherent >= 9 * (long long)sampled) {
            tune.refill_lanes = tune_if_coherent.refill_lanes;
            tune.leaf_batch_lanes = tune_if_coherent.leaf_batch_lanes;
            tune.step_cap = tune_if_coherent.step_cap;
            tune.pool1_below = tune_if_coherent.pool1_below;
        }
    }
    constexpr bool WORK = MODE == kCountWork;
    __shared__ int stack_mem[(kStackDepth + kColdRows) * kExtendBlock];
    __shared__ int coop_items[(kExtendBlock / 64) * 2 * kCoopItems];
    __shared__ unsigned long long coop_res[kExtendBlock];
    LdsStack stk{stack_mem + threadIdx.x};
    const int n = count_ptr ? *count_ptr : n_fixed;
    const int lane = (int)(threadIdx.x & 63u);
    // the 8 ranges of the queue (multiples of 64 rays) and the one this wave pulls from
    const int n_heads = tune.n_heads;  // 1 or 8
    const int range = n_heads == 1 ? n : (int)((((long long)n + 7) / 8 + 63) & ~63ll);
    int head = n_heads == 1 ? 0 : (int)(blockIdx.x & 7u);
    int heads_dry = 0;
    WorkCounters wc = {0u, 0u, 0u, 0u, 0u, 0u};
    const int root0 = sc.n_meshes > 0 ? ldk_i32(&sc.meshes[0].root) : 0;  // wave-uniform: one link less in a fresh ray's load chain
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    unsigned int hit_tris = 0;
#ifdef MIPT_PHASE_STATS
    PhaseStats ps = {};
#endif
    // developer build (-DMIPT_PHASE_CLOCKS, scripts/build_phase_stats.sh): shader clocks of every wave per section of the loop
#ifdef MIPT_PHASE_CLOCKS
    // (32-bit, forced into scalar registers: as vector values the compiler spilled them to scratch memory, and every reload
    //  waited for ALL outstanding vector memory operations -- the result store of the finish section among them, 8-13 k clocks)
    uint32_t pclk0 = 0u, pclk1 = 0u, pclk2 = 0u, pclk3 = 0u, pclk4 = 0u, pclk5 = 0u;
    uint32_t pclk_t = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter());
#define PCLK(i)                                                                                            \
    do {                                                                                                   \
        const uint32_t now_ = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter());      \
        pclk##i = __builtin_amdgcn_readfirstlane(pclk##i + (now_ - pclk_t));                               \
        pclk_t = now_;                                                                                     \
    } while (0)
#else
#define PCLK(i)
#endif

    bool has_ray = false, exhausted = false;
    bool start_mesh = false;  // lane must open mesh `m`, or store its result when m == n_meshes
    bool pending = false;     // lane sits at an un-culled leaf (`nd`) waiting for a batched scan
    bool need_pop = false;    // lane must continue with the node on top of its stack
    // two flags ride in the top bits of `slot` (< 2^30, touched only at fetch and store) to stay inside 96 VGPRs:
    //   kTaintBit   this mesh's search dropped a subtree by distance (see descend_fast)
    //   kNoPruneBit ... and then found a hit that counts: the mesh is being repeated without distance pruning
    constexpr uint32_t kTaintBit = 1u << 31, kNoPruneBit = 1u << 30, kSlotMask = (1u << 30) - 1u;
    const bool prune_meshes = !COUNT && sc.n_meshes > 1;  // wave-uniform
    uint32_t slot = 0u;
    int m = 0, node = 0, sp = 0;
#ifndef MIPT_COLD_MASK
#define MIPT_COLD_MASK 7
#endif
#if (MIPT_COLD_MASK & 1)
#define COLD_QT() as_f32((uint32_t)stk.cold_get(0))
#define COLD_SET_QT(t_) stk.cold_set(0, (int)as_u32(t_))
#else
    float q_t = -1.0f;
#define COLD_QT() q_t
#define COLD_SET_QT(t_) q_t = (t_)
#endif
#if (MIPT_COLD_MASK & 2)
#define COLD_QPRIM() stk.cold_get(1)
#define COLD_SET_QPRIM(p_) stk.cold_set(1, (p_))
#else
    int q_prim = -1;
#define COLD_QPRIM() q_prim
#define COLD_SET_QPRIM(p_) q_prim = (p_)
#endif
#if (MIPT_COLD_MASK & 4)
#define COLD_PRIM() stk.cold_get(2)
#define COLD_SET_PRIM(p_) stk.cold_set(2, (p_))
#else
    int prim = -1;
#define COLD_PRIM() prim
#define COLD_SET_PRIM(p_) prim = (p_)
#endif
    float best = -1.0f;
    KdNode nd = {0u, kLeafTag};      // the 8 bytes of `node`
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    SlabBox cur;  // fast path: a masked slab (see descend_fast)
#pragma unroll
    for (int k = 0; k < 3; k++) cur.lo[k] = cur.hi[k] = cur.tlo[k] = cur.thi[k] = 0.f;

    // what follows a leaf visit: bt > 0 = accepted hit at triangle bi (mesh.cuh:300 ends this mesh; kernel.cu:149
    // keeps the earlier mesh on ties), else continue with the stacked far child, else the mesh is exhausted
    const float inf = __builtin_inff();
    auto closest_so_far = [&]() {
        const float qt = COLD_QT();
        return fmin_raw(best > 0.0f ? best : inf, qt > 0.0f ? qt : inf);
    };
    auto after_leaf = [&](float bt, int bi) {
        if (prune_meshes && bt > 0.0f && (slot & kTaintBit) != 0u && bt <= closest_so_far()) {
            slot |= kNoPruneBit;  // the hit would be used, but the reference may have stopped in a subtree that was dropped
            start_mesh = true;
        } else if (bt > 0.0f) {
            // (equal hits: the lower triangle id is the earlier mesh -- the production kernel opens meshes out of order)
            bool take = best < 0.0f || bt < best;
            if (!take && bt == best) take = bi < COLD_PRIM();
            if (take) { best = bt; COLD_SET_PRIM(bi); }
            m++;
            start_mesh = true;
            slot &= ~kNoPruneBit;
        } else if (sp == 0 || sp > kStackDepth) {  // (sp > kStackDepth: unreachable for trees of kd_build.cpp)
            m++;
            start_mesh = true;
            slot &= ~kNoPruneBit;
        } else {
            need_pop = true;
        }
    };

    for (;;) {
        PSTAT(ps, 0);
        // ---- rays that ran out of meshes: Scene::firstIntersect's result (kernel.cu:145-160) ------------------
        if (has_ray && start_mesh && m >= sc.n_meshes) {
            // spheres/planes were evaluated when the ray was made; a mesh hit wins ties because meshes come first
            PSTAT(ps, 1);
            const float qt = COLD_QT();
            const bool mesh_wins = (best > 0.0f) && (qt < 0.0f || best <= qt);
            HitRec h;
            h.t = mesh_wins ? best : qt;
            h.prim = mesh_wins ? COLD_PRIM() : COLD_QPRIM();
            hits[slot & kSlotMask] = h;
            if (COUNT) hit_tris += mesh_wins ? 1u : 0u;
            has_ray = false;
            start_mesh = false;
        }
        PCLK(0);
        // ---- refill idle lanes from the global work counter --------------------------------------------
        const unsigned long long idle = __ballot(!has_ray);
        if (idle != 0ull && !exhausted) {
            const int n_idle = __popcll(idle);
            if (n_idle >= tune.refill_lanes || n_idle == 64) {
                const int leader = __ffsll((long long)idle) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(work + head * kCursorStride, n_idle);
                base = __shfl(base, leader, 64);
                const long long lo = (long long)range * head;               // wave-uniform
                const long long hi = lo + range < (long long)n ? lo + range : (long long)n;
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
                if (!has_ray && lo + base + rank < hi) {
                    PSTAT(ps, 2);
                    const int qi = (int)(lo + base + rank);
                    slot = (uint32_t)(queue ? queue[qi] : qi);
                    const f4 d4 = ray_d[slot];
                    if (tune.common_origin == 0) {  // wave-uniform
                        const f4 o4 = ray_o[slot];
                        o = mk3(o4.x, o4.y, o4.z);
                    } else if (tune.common_origin == 1) {
                        o = mk3(sc.cam_eye[0], sc.cam_eye[1], sc.cam_eye[2]);
                    } else {
                        const f4 l0 = ldk4(sc.lights, 0);
                        o = mk3(l0.x, l0.y, l0.z);
                    }
                    d = mk3(d4.x, d4.y, d4.z);
                    best = -1.0f;
                    COLD_SET_PRIM(-1);
                    {  // the sphere / plane answer parked in the hit slot by the stage that made the ray (kernel.cu:151-158)
                        const HitRec q = hits[slot];
                        COLD_SET_QT(q.t);
                        COLD_SET_QPRIM(q.prim);
                    }
                    m = 0;
                    start_mesh = true;
                    pending = false;
                    need_pop = false;
                    has_ray = true;
                }
                if (lo + base + n_idle >= hi) {  // wave-uniform: this range is used up -- on to the next one
                    heads_dry++;
                    head = (head + 1) & 7;
                    if (heads_dry >= n_heads) exhausted = true;
                }
            }
        }
        PCLK(1);
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- open the next mesh: root box test (mesh.cuh:254) -------------------------------------------------
        // a mesh whose root cull box the ray misses cannot be hit: skipped before any exact slab arithmetic
        bool skip_mesh = false;
        // position m -> root of the m-th mesh along the ray's dominant axis, near to far (sc.mesh_order)
        auto root_at = [&](int pos) -> int {
            if (COUNT) return sc.meshes[pos].root;  // the reference's order
            if (sc.n_meshes == 1) return root0;
            const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
            const int a = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
            const float da = a == 0 ? d.x : (a == 1 ? d.y : d.z);
            return sc.mesh_order[a * sc.n_meshes + (da < 0.0f ? sc.n_meshes - 1 - pos : pos)];
        };
        if (!COUNT && has_ray && start_mesh) {
            // (bounded: lanes in other states wait meanwhile; with few meshes one try per round measured faster --
            //  default scene, 3 meshes: 523 vs 433 Msamples/s bidirectional; 8 meshes: 49 vs 54 the other way)
            const int max_tries = sc.n_meshes > 4 ? 4 : 1;
            const float t_max = closest_so_far();
#pragma unroll 1
            for (int tries = 0; tries < max_tries && m < sc.n_meshes; tries++) {
                skip_mesh = mesh_skippable(sc, root_at(m), o, d, t_max);
                if (!skip_mesh) break;
                m++;
            }
        }
        if (has_ray && start_mesh && !skip_mesh && m < sc.n_meshes) {
            PSTAT(ps, 3);
            node = root_at(m);
            nd = sc.nodes[node];
            load_slab(sc, node, o, d, cur);
            bool enter;
            if (COUNT) {
                enter = slab_hit(cur, o, d) > 0.0f;
            } else {
                mask_slab(cur, o, d);
                enter = masked_min(cur.tlo, cur.thi) > 0.0f;
                if (WORK) wc.pops++;
            }
            if (enter) {
                sp = 0;
                start_mesh = false;
                slot &= ~kTaintBit;
                if (COUNT) { tc.pops++; tc.nodes++; }
                else if ((nd.b >> 30) == 3u) pending = true;  // single-leaf mesh
            } else {
                m++;  // the ray misses this mesh's root box
            }
        }
        PCLK(2);
        // ---- postponed leaf scans: run the candidate loop only when enough lanes wait for one -------------------
        // (about 9 of 10 leaf visits are culled; scanning the rest one by one would run the loop at ~10% lane use)
        {
            const int n_pend = __popcll(__ballot(pending));
            const int n_trav = __popcll(__ballot(has_ray && !pending && !start_mesh));
            if (n_pend >= tune.leaf_batch_lanes || (n_pend > 0 && n_trav < tune.leaf_batch_lanes)) {  // wave-uniform
                float bt;
                int bi;
                if (pending) { PSTAT(ps, 4); }
                leaf_scan_coop<WORK>(sc, pending, (int)nd.a, (int)(nd.b & kIdxMask), o, d, cur, bt, bi, wc, coop_items, coop_res, n_pend < tune.pool1_below PSTAT_PASS);
                if (pending) {
                    pending = false;
                    after_leaf(bt, bi);
                }
            }
        }
        PCLK(3);
        // ---- continue with the far child that was stacked (mesh.cuh:263): its box is re-read --------------------
        if (COUNT) {
            if (need_pop) {
                sp--;
                node = stk.pop(sp);
                nd = sc.nodes[node];
                load_slab(sc, node, o, d, cur);
                need_pop = false;
                tc.pops++;
                tc.nodes++;
            }
        } else if (need_pop) {  // only subtrees that passed their cull test were stacked (descend_fast)
            PSTAT(ps, 7);
            sp--;
            node = stk.pop(sp);
            nd = sc.nodes[node];  // (the 8 bytes also ride in the parent's step record; reading them there instead -- one line of
                                  //  `nodes` fewer per pop -- measured no gain on the incoherent batch and -0.6 % on the 1080p frame)
            load_slab(sc, node, o, d, cur);
            mask_slab(cur, o, d);
#ifdef MIPT_POP_TWICE
            {   // sensitivity probe (developer build): the pop's slab arithmetic executed a second time, same results.  What the
                // kernel slows down by is the share of its time this arithmetic takes -- the ceiling of what pooling it could save.
                f3 o2 = o;
                asm volatile("" : "+v"(o2.x), "+v"(o2.y), "+v"(o2.z));
                SlabBox c2;
                load_slab(sc, node, o2, d, c2);
                mask_slab(c2, o2, d);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    cur.tlo[k] = fmin_raw(cur.tlo[k], c2.tlo[k]);
                    cur.thi[k] = fmin_raw(cur.thi[k], c2.thi[k]);
                }
            }
#endif
            need_pop = false;
            if (WORK) wc.pops++;
            if ((nd.b >> 30) == 3u) pending = true;
        }
        PCLK(4);
        if (!has_ray || pending || start_mesh) continue;

        // ---- walk down (at most step_cap steps per round), then the leaf visit ---------------------------------
        PSTAT(ps, 8);
        if (COUNT) {  // the reference's walk, node for node (its counters are the unit of SURVEY 8d)
            descend_to_leaf<LdsStack, true>(sc, o, d, cur, node, nd, sp, stk, tc, tune.step_cap PSTAT_PASS);
            if ((nd.b >> 30) != 3u) continue;  // out of steps for this round: keep walking next round
            tc.refs += nd.b & kIdxMask;
            tc.tris += nd.b & kIdxMask;
            if (!leaf_culled(sc, node, o, d)) {
                pending = true;  // scanned later, together with other lanes' leaves
            } else {
                tc.culled += nd.b & kIdxMask;
                after_leaf(-1.0f, -1);
            }
        } else {
            const float t_prune = (prune_meshes && (slot & kNoPruneBit) == 0u) ? closest_so_far() : inf;
            const bool dead_end = descend_fast<LdsStack, WORK>(sc, o, d, cur, node, nd, sp, stk, tune.step_cap, prune_meshes, t_prune, slot, kTaintBit, wc PSTAT_PASS);
            if (dead_end) after_leaf(-1.0f, -1);                 // same continuation as a fruitless leaf visit
            else if ((nd.b >> 30) == 3u) pending = true;        // a leaf that passed its cull test: scanned in a batch
        }
        PCLK(5);
    }
    if (COUNT) flush_counters(tc, hit_tris, ctr);
    if (WORK) flush_work_counters(wc, ctr + kCtrWork / 8);
#ifdef MIPT_PHASE_CLOCKS
    if (!COUNT && lane == 0) {
        const uint32_t pc[6] = {pclk0, pclk1, pclk2, pclk3, pclk4, pclk5};
        for (int i = 0; i < 6; i++) atomicAdd(&ctr[16 + i], (unsigned long long)pc[i]);
    }
#endif
#ifdef MIPT_PHASE_STATS
    if (!COUNT) {  // ctr + 16: kCtrPhase (device_ctx.h)
        for (int i = 0; i < kPhases; i++) {
            const unsigned int e = wave_sum(ps.exec[i]), l = wave_sum(ps.lanes[i]);
            if (lane == 0) {
                atomicAdd(&ctr[16 + 2 * i], (unsigned long long)e);
                atomicAdd(&ctr[16 + 2 * i + 1], (unsigned long long)l);
            }
        }
    }
#endif
}

// Reference formulation (one ray start to finish per lane, grid-stride): kept for A/B timing of the
// persistent kernel in one process (MIPT_EXTEND=simple) and as a second implementation in the parity tests.
template <bool COUNT>
__global__ __launch_bounds__(kExtendBlock) void k_extend_simple(DScene sc, const f4 *__restrict__ ray_o,
                                                                const f4 *__restrict__ ray_d, const int *__restrict__ queue,
                                                                const int *__restrict__ count_ptr, int n_fixed,
                                                                HitRec *__restrict__ hits, int *__restrict__ work,
                                                                unsigned long long *ctr) {
    __shared__ int stack_mem[kStackDepth * kExtendBlock];
    LdsStack stk{stack_mem + threadIdx.x};
    const int n = count_ptr ? *count_ptr : n_fixed;
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    unsigned int hit_tris = 0;
    for (int i = blockIdx.x * kExtendBlock + threadIdx.x; i < n; i += gridDim.x * kExtendBlock) {
        const int slot = queue ? queue[i] : i;
        const f4 o4 = ray_o[slot], d4 = ray_d[slot];
        float t;
        int prim;
        scene_closest<LdsStack, COUNT>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), stk, t, prim, tc);
        HitRec h;
        h.t = t;
        h.prim = prim;
        hits[slot] = h;
        if (COUNT) hit_tris += (prim >= 0 && prim < sc.total_tris) ? 1u : 0u;
    }
    if (COUNT) flush_counters(tc, hit_tris, ctr);
}

// Ray(o, d) constructor (ray.cuh:10): dir = d * (1/(|d|+eps)).  Packed xyz triples -> float4 pairs; the sphere / plane
// answer of each ray is parked in its hit slot (see quadrics_closest).
// Coherence probe (probe != null): the first wave of every 64th block, when all its 64 rays exist, looks at the spread of
// its rays -- origins within 5 % of their magnitude (+1), directions within 0.25 per component: a bundle like 64
// neighbouring camera rays -- and counts itself in probe[0], and in probe[1] if coherent.  The extend kernel picks its
// scheduling thresholds from the two counts (a choice that changes no result).
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
static __global__ __launch_bounds__(256) void k_make_rays(DScene sc, const float *__restrict__ o3, const float *__restrict__ d3,
                                                          int64_t n, f4 *__restrict__ ray_o, f4 *__restrict__ ray_d,
                                                          HitRec *__restrict__ hits, int *__restrict__ probe) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    if (live) {
        o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
        d = norm3(mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]));
        ray_o[i] = f4{o.x, o.y, o.z, 0.f};
        ray_d[i] = f4{d.x, d.y, d.z, 0.f};
        HitRec h;
        quadrics_closest(sc, o, d, h.t, h.prim);
        hits[i] = h;
    }
    if (probe && (blockIdx.x & 63u) == 0u && threadIdx.x < 64u && __ballot(live) == ~0ull) {
        // spread of each component = max + max of the negated values; NaN rays compare false below: not coherent
        const float so = fmaxf(fmaxf(wave_max_f(o.x) + wave_max_f(-o.x), wave_max_f(o.y) + wave_max_f(-o.y)), wave_max_f(o.z) + wave_max_f(-o.z));
        const float sd = fmaxf(fmaxf(wave_max_f(d.x) + wave_max_f(-d.x), wave_max_f(d.y) + wave_max_f(-d.y)), wave_max_f(d.z) + wave_max_f(-d.z));
        const float mo = wave_max_f(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)));
        if (threadIdx.x == 0) {
            atomicAdd(&probe[0], 1);
            if (so <= 0.05f * (mo + 1.0f) && sd <= 0.25f) atomicAdd(&probe[1], 1);
        }
    }
}

static __global__ __launch_bounds__(256) void k_split_hits(const HitRec *__restrict__ hits, int64_t n,
                                                           int32_t *__restrict__ prim, float *__restrict__ t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HitRec h = hits[i];
    prim[i] = h.prim;
    t[i] = h.t;
}

// Launch the extend stage on dc.stream.  `work` must point at 8 zeroed ints, kCursorStride apart (the persistent kernel's
// ray cursors).  mode: kProd / kCountRef / kCountWork.
inline void launch_extend(const DeviceCtx &dc, int mode, const f4 *ro, const f4 *rd, const int *queue, const int *count_ptr,
                          int n_fixed, int64_t n_max, HitRec *hits, int *work, unsigned long long *ctr, bool coherent_queue = false,
                          int common_origin = 0, const int *probe = nullptr) {  // (common_origin != 0, probe: persistent kernel only)
    ExtendTuning tune = coherent_queue ? dc.tune_coherent : dc.tune;
    tune.common_origin = common_origin;
    tune.n_heads = dc.n_heads;
    const bool count = mode == kCountRef;
    if (dc.extend_simple) {
        const int64_t want = (n_max + kExtendBlock - 1) / kExtendBlock;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)dc.num_cus * 8));
        if (count)
            hipLaunchKernelGGL(k_extend_simple<true>, dim3(grid), dim3(kExtendBlock), 0, dc.stream, dc.view, ro, rd, queue,
                               count_ptr, n_fixed, hits, work, ctr);
        else
            hipLaunchKernelGGL(k_extend_simple<false>, dim3(grid), dim3(kExtendBlock), 0, dc.stream, dc.view, ro, rd, queue,
                               count_ptr, n_fixed, hits, work, ctr);
        return;
    }
    const int64_t want = (n_max + kExtendBlock - 1) / kExtendBlock;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)dc.num_cus * dc.extend_blocks_per_cu));
    if (mode == kCountRef)
        hipLaunchKernelGGL(k_extend<kCountRef>, dim3(grid), dim3(kExtendBlock), 0, dc.stream, dc.view, ro, rd, queue, count_ptr,
                           n_fixed, hits, work, ctr, tune, dc.tune_batch_coherent, probe);
    else if (mode == kCountWork)
        hipLaunchKernelGGL(k_extend<kCountWork>, dim3(grid), dim3(kExtendBlock), 0, dc.stream, dc.view, ro, rd, queue, count_ptr,
                           n_fixed, hits, work, ctr, tune, dc.tune_batch_coherent, probe);
    else
        hipLaunchKernelGGL(k_extend<kProd>, dim3(grid), dim3(kExtendBlock), 0, dc.stream, dc.view, ro, rd, queue, count_ptr,
                           n_fixed, hits, work, ctr, tune, dc.tune_batch_coherent, probe);
}

// Resident 256-thread blocks per CU of the persistent kernel (occupancy API, capped like the hardware admits).
inline int query_extend_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<kProd>, kExtendBlock, 0) != hipSuccess || nb <= 0) nb = 4;
    return std::min(nb, 8);
}

}  // namespace mipt
