// kernels_extend.h -- the `extend` stage: kd-tree closest hit over a ray queue (gfx950, wave64).
//
// One lane = one ray (T3/T4 of SURVEY.md 8a).  The block is 256 lanes = 4 waves; the traversal
// stack is an LDS array laid out [slot][lane] so that a wave's 64 lanes hit 64 consecutive banks.
// The grid is sized to the chip (256 CUs x resident blocks) and strides over the queue, so the
// same launch shape serves 4K-pixel frames and the shrinking queues of later bounces, with the
// ray count read from device memory (no host round trip between bounces).
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
};

__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// rays: o/d as float4 (w unused), d already normalised by the Ray constructor rule (ray.cuh:10).
// queue: optional indirection (slot ids); count_ptr: optional device-side ray count.
template <bool COUNT>
__global__ __launch_bounds__(kExtendBlock) void k_extend(DScene sc, const f4 *__restrict__ ray_o,
                                                         const f4 *__restrict__ ray_d, const int *__restrict__ queue,
                                                         const int *__restrict__ count_ptr, int n_fixed,
                                                         HitRec *__restrict__ hits, unsigned long long *ctr) {
    __shared__ int stack_mem[kStackDepth * kExtendBlock];
    LdsStack stk{stack_mem + threadIdx.x};
    const int n = count_ptr ? *count_ptr : n_fixed;
    TravCounters tc = {0u, 0u, 0u, 0u};
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
    if (COUNT) {
        const unsigned int a = wave_sum(tc.nodes), b = wave_sum(tc.pops), c = wave_sum(tc.refs), d = wave_sum(tc.tris),
                           e = wave_sum(hit_tris);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ctr[0], (unsigned long long)a);
            atomicAdd(&ctr[1], (unsigned long long)b);
            atomicAdd(&ctr[2], (unsigned long long)c);
            atomicAdd(&ctr[3], (unsigned long long)d);
            atomicAdd(&ctr[4], (unsigned long long)e);
        }
    }
}

// Ray(o, d) constructor (ray.cuh:10): dir = d * (1/(|d|+eps)).  Packed xyz triples -> float4 pairs.
static __global__ __launch_bounds__(256) void k_make_rays(const float *__restrict__ o3, const float *__restrict__ d3, int64_t n,
                                                   f4 *__restrict__ ray_o, f4 *__restrict__ ray_d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f3 o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
    const f3 d = norm3(mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]));
    ray_o[i] = f4{o.x, o.y, o.z, 0.f};
    ray_d[i] = f4{d.x, d.y, d.z, 0.f};
}

static __global__ __launch_bounds__(256) void k_split_hits(const HitRec *__restrict__ hits, int64_t n, int32_t *__restrict__ prim,
                                                    float *__restrict__ t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HitRec h = hits[i];
    prim[i] = h.prim;
    t[i] = h.t;
}

inline int extend_grid(const DeviceCtx &dc, int64_t n_max) {
    const int64_t want = (n_max + kExtendBlock - 1) / kExtendBlock;
    const int64_t cap = (int64_t)dc.num_cus * 8;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

}  // namespace mipt
