// wavefront.hip -- wave64 wavefront path tracer: replaces the per-pixel megakernel `render`
// (kernel.cu:310-346) and the recursive-by-loop `Scene::trace` (kernel.cu:173-306).
//
// One path = one (pixel, sample).  A pass keeps P = pixels x samples_per_pass paths in flight in SoA
// buffers and advances them one vertex at a time through short kernels:
//
//   raygen                    jittered Camera::getRay, RNG draws 0,1                     kernel.cu:338-341
//   [bidirectional only]
//   lightgen                  light choice + emission direction                          kernel.cu:178-183
//   extend / light_shade(i)   light sub-path vertex i, Russian roulette, bounce ray      kernel.cu:184-231
//   [camera sub-path, per vertex i]
//   extend                    closest hit of the camera rays                             kernel.cu:240
//   shade(i)                  direct-light + light-vertex connections emitted as batched shadow rays,
//                             BSDF sampling / roulette, next ray, queue compaction        kernel.cu:244-302
//   [bin]                     shadow queue of vertex >= 1 (and the first light rays) ordered by direction: they
//                             share an origin, so binned they are as coherent as primary rays (queue_sort.h)
//   extend                    closest hit of the shadow-ray batch (all rays are closest-hit, Q10)
//   resolve                   visibility compare + in-order accumulation of contributions kernel.cu:249-256,268-279
//                             (connections answered without the extend kernel are resolved by shade itself)
//   accumulate                image[id] += trace/nSamples, in sample order               kernel.cu:339-343
//
// Queues between stages are built with ballot + mbcnt prefix ranks, an LDS scan over the block's waves and ONE atomic
// per block and queue; ray counts stay on the device (every extend and every binning kernel reads its count from memory)
// and a pass makes no host round trip.
// Every path's contributions are summed in the reference's order, and the order of a queue changes no result, so the
// images are bit-identical to the CPU oracle's.
#include "wavefront.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "kernels_extend.h"
#include "queue_sort.h"
#include "shade.h"

namespace mipt {
namespace {

constexpr int kBlock = 512;
// streaming stores: written once, read once by a later kernel -- keep them from displacing the gathers' lines
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(f4 *p, f4 v) { __builtin_nontemporal_store((v4f_nt){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_nt *>(p)); }
__device__ __forceinline__ void st_nt(HitRec *p, HitRec v) {
    typedef float v2f_nt __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((v2f_nt){v.t, as_f32((uint32_t)v.prim)}, reinterpret_cast<v2f_nt *>(p));
}


// ---- queue append: ballot + mbcnt rank inside a wave, LDS scan over the block's waves, ONE atomic per block ----
// Same-address atomics complete at ~88 M/s on MI355X (scripts/micro/atomic_bench.hip): one per wave capped the
// shading kernels at ~5.6 G paths/s per queue, one per 512-thread block does not.
// Every thread of the block must call it (convergent); `lds` holds 2 * (kWaves + 1) ints and `phase` alternates
// between its halves so that back-to-back appends need no third barrier.
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ int block_append(bool pred, int *counter, int *lds, int &phase) {
    int *slot = lds + phase * (kWaves + 1);
    phase ^= 1;
    const unsigned long long mask = __ballot(pred);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) slot[wave] = __popcll(mask);
    __syncthreads();
    if (wave == 0) {
        const int c = lane < kWaves ? slot[lane] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < kWaves; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        const int total = __shfl(incl, kWaves - 1, 64);
        int base = 0;
        if (lane == 0 && total > 0) base = atomicAdd(counter, total);
        base = __shfl(base, 0, 64);
        if (lane < kWaves) slot[lane] = base + incl - c;
    }
    __syncthreads();
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
    return pred ? slot[wave] + rank : -1;
}

// Three appends at once (k_shade: next-vertex state, camera-ray queue, shadow-ray queue): one barrier pair and three
// atomics IN FLIGHT TOGETHER instead of three barrier pairs around three atomic round trips, during each of which the
// whole block waits.  Wave 0 scans the three per-wave count vectors in 8-lane segments.  `lds`: 2 * 3 * kWaves ints.
struct Idx3 {
    int a, b, c;
};
__device__ __forceinline__ Idx3 block_append3(bool pa, bool pb, bool pc, int *ca, int *cb, int *cc, int *lds, int &phase) {
    static_assert(kWaves == 8, "8-lane segments");
    int *slot = lds + phase * (3 * kWaves);
    phase ^= 1;
    const unsigned long long ma = __ballot(pa), mb = __ballot(pb), mc = __ballot(pc);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) {
        slot[wave] = __popcll(ma);
        slot[kWaves + wave] = __popcll(mb);
        slot[2 * kWaves + wave] = __popcll(mc);
    }
    __syncthreads();
    if (wave == 0) {
        const int seg = lane >> 3, l = lane & 7;
        const int c = lane < 3 * kWaves ? slot[lane] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < kWaves; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (l >= off) incl += t;
        }
        const int total = __shfl(incl, lane | 7, 64);
        int base = 0;
        if (l == 0 && lane < 3 * kWaves && total > 0) base = atomicAdd(seg == 0 ? ca : (seg == 1 ? cb : cc), total);
        base = __shfl(base, lane & ~7, 64);
        if (lane < 3 * kWaves) slot[lane] = base + incl - c;
    }
    __syncthreads();
    auto rank = [](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    };
    Idx3 r;
    r.a = pa ? slot[wave] + rank(ma) : -1;
    r.b = pb ? slot[kWaves + wave] + rank(mb) : -1;
    r.c = pc ? slot[2 * kWaves + wave] + rank(mc) : -1;
    return r;
}

// Append `cnt` >= 0 entries per thread with one atomic per block; returns the index of the thread's first entry.
__device__ __forceinline__ int block_append_n(int cnt, int *counter, int *lds, int &phase) {
    int *slot = lds + phase * (kWaves + 1);
    phase ^= 1;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) slot[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int c = lane < kWaves ? slot[lane] : 0;
        int wincl = c;
#pragma unroll
        for (int off = 1; off < kWaves; off <<= 1) {
            const int t = __shfl_up(wincl, off, 64);
            if (lane >= off) wincl += t;
        }
        const int total = __shfl(wincl, kWaves - 1, 64);
        int base = 0;
        if (lane == 0 && total > 0) base = atomicAdd(counter, total);
        base = __shfl(base, 0, 64);
        if (lane < kWaves) slot[lane] = base + wincl - c;
    }
    __syncthreads();
    return slot[wave] + incl - cnt;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Park the sphere / plane answer of a freshly generated ray in its hit slot; if the ray misses every mesh's root cull
// box that answer is final, else queue the ray for extend.  Call it from block-convergent code.  Returns true when the
// ray was decided here (callers tally those).
__device__ __forceinline__ bool emit_ray(const DScene &sc, bool valid, f3 o, f3 d, HitRec *hit_slot, int queue_value, int *queue,
                                         int *queue_count, int *lds, int &phase, uint32_t *keys = nullptr, uint32_t key = 0u) {
    bool need_extend = false;
    if (valid) {
        HitRec h;
        quadrics_closest(sc, o, d, h.t, h.prim);
        *hit_slot = h;
        need_extend = !misses_every_mesh(sc, o, d);
    }
    const int qi = block_append(need_extend, queue_count, lds, phase);
    if (need_extend) {
        queue[qi] = queue_value;
        if (keys) keys[qi] = key;  // sort key of the queue entry (see wavefront_render)
    }
    return valid && !need_extend;
}

// Shadow rays of the simple integrator all start at lights[0]: binned by direction they become bundles as coherent as
// primary rays, whatever the order of the paths that asked for them (second-vertex shadow rays of the 1080p box-room
// frame: 1.43x faster through the extend kernel).  256 x 256 cells of the octahedral map of the direction.
__device__ __forceinline__ uint32_t direction_cell(f3 d) {
    const float s = rcp_approx(fabsf(d.x) + fabsf(d.y) + fabsf(d.z));
    float u = d.x * s, w = d.z * s;
    if (d.y < 0.0f) {
        const float uu = (1.0f - fabsf(w)) * (u >= 0.0f ? 1.0f : -1.0f);
        w = (1.0f - fabsf(u)) * (w >= 0.0f ? 1.0f : -1.0f);
        u = uu;
    }
    const int iu = min(4095, max(0, (int)((u * 0.5f + 0.5f) * 4096.0f))), iw = min(4095, max(0, (int)((w * 0.5f + 0.5f) * 4096.0f)));
    auto spread = [](uint32_t x) {  // 12 bits -> every other bit of 24
        x = (x | (x << 8)) & 0x00ff00ffu;
        x = (x | (x << 4)) & 0x0f0f0f0fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        return x;
    };
    return spread((uint32_t)iu) | (spread((uint32_t)iw) << 1);  // Morton order: consecutive keys = a compact patch of directions
}

constexpr unsigned kShadowKeyBits = 24;
constexpr uint32_t kKeyLast = (1u << kShadowKeyBits) - 1u;  // rays that are not binned keep their order behind the binned ones

struct PassParams {
    int W, H, npix;
    int spp;             // normalisation constant
    int sample0;         // sample index of slot 0 of this pass
    int sample_stride;   // sample index step between slots
    int n_slots;         // valid slots in this pass (paths = n_slots * npix)
    int Dc, Dl, K;       // camera depth, light depth, shadow slots per path (1 + Dl, + 1 with env_light)
    int use_env;
    int env_light;       // use_envmap == 2: the cube map is also sampled as a light at every camera vertex (shade.h); its
                         // shadow ray sits in slot 1 + Dl of the path
    uint32_t seed;
    int tiles_x;         // > 0: paths of a sample are laid out in 8x8 pixel tiles (W and H multiples of 8), else row-major
    int common_origins;  // simple integrator, persistent extend kernel: primary rays and direct-connection rays are launched
                         // with their common origin (eye / lights[0]); vertex 0 has no stored origin, throughput or draw count
};

// Position of a path inside its sample slot <-> pixel.  64 consecutive paths (one wave of primary rays, and what the
// queues keep together afterwards) cover an 8x8 tile instead of a 64x1 strip: +6..10 % on primary-ray traversal.
__device__ __forceinline__ int pixel_of(const PassParams &pp, int p) {
    if (pp.tiles_x <= 0) return p;
    const int tile = p >> 6, in = p & 63;
    const int ty = tile / pp.tiles_x, tx = tile - ty * pp.tiles_x;
    return (ty * 8 + (in >> 3)) * pp.W + tx * 8 + (in & 7);
}
__device__ __forceinline__ int position_of(const PassParams &pp, int pixel) {
    if (pp.tiles_x <= 0) return pixel;
    const int row = pixel / pp.W, col = pixel - row * pp.W;
    return ((row >> 3) * pp.tiles_x + (col >> 3)) * 64 + (row & 7) * 8 + (col & 7);
}

struct WfBuffers {
    // Camera-path state of the paths ALIVE at a vertex, dense (index = rank in the vertex's queue) and ping-ponged by
    // vertex parity: a path that continues writes its next state at its rank in the next vertex's arrays.  Indexing by
    // path id instead made every vertex stream the whole pass's state through the cache no matter how few paths were
    // left (a 128-byte line holds eight paths).  st_o.w = bits(RNG draw counter), st_d.w = bits(path id).
    f4 *st_o[2], *st_d[2], *st_T[2];
    HitRec *st_hit[2];
    f4 *path_rad;        // by path id: radiance accumulated so far
    uint32_t *path_rng;  // by path id: RNG draw counter handed from the light sub-path to the camera sub-path
    uint32_t *sh_mask;   // by dense index of the current vertex
    f4 *sh_o, *sh_d, *sh_target, *sh_contrib;  // [dense index * K + k]
    HitRec *sh_hit;
    int *q_cam0;                 // bidirectional only: path ids whose camera path is traced (Q11); vertex 0 is indexed by path id
    int *q_light[2], *q_shadow;  // light-path bookkeeping / shadow rays to trace
    uint32_t *q_shadow_key;      // direction cell of each queued direct-connection ray (see direction_cell); connection rays to
                                 // light vertices get kKeyLast; or null
    uint32_t *q_light_key;       // light index << 24 | direction cell of each queued first light ray, or null
    int *q_ext_cam, *q_ext_light;           // the subset of camera / light rays that needs the extend kernel
    int *counts;                 // per-stage queue counts of the current pass
    unsigned long long *totals;  // [0] camera rays [1] shadow rays [2] light rays [3] rays traced by k_extend [4] non-finite samples
    f4 *lv_pos, *lv_nrm, *lv_T;  // light vertices [path*Dl + j]: (pos, valid) (normal, pdf) (throughput, -)
    f4 *l_o, *l_d, *l_state;     // light rays; l_state = (emission dir, pdf carried to the next vertex)
    HitRec *l_hit;
};

// count slots inside `counts`
constexpr int kCntCam = 0;      // + i  (camera queue feeding vertex i), i <= 16
constexpr int kCntShadow = 20;  // + i
constexpr int kCntLight = 40;   // + i
constexpr int kCntExtCam = 128;     // + i: camera rays of vertex i that need the extend kernel (not trivially decided)
constexpr int kCntExtLight = 148;   // + i
constexpr int kCntShadowTriv = 168; // + i: shadow rays answered on the spot
constexpr int kCntNonFinite = 188;  // samples of the pass whose radiance is not finite
constexpr int kCntTotal = 192;
constexpr int kCntPad = 64;  // counters sit 256 B apart: atomics on neighbouring words of one line serialise like same-address ones
// behind the table: the 8 ray cursors of every persistent extend launch of the pass (launch index * 8 + range)
constexpr int kCntWork = kCntTotal;
constexpr int kMaxLaunches = 3 * kMaxPathDepth;
constexpr int kCntSlots = kCntWork + 8 * kMaxLaunches;
static_assert(kCntPad == kCursorStride, "launch_extend expects its 8 cursors one counter slot apart");

__global__ __launch_bounds__(kBlock) void k_raygen(DScene sc, PassParams pp, WfBuffers b) {
    __shared__ int ap_lds[2 * (kWaves + 1)];
    int phase = 0;
    const int pid = blockIdx.x * kBlock + threadIdx.x;
    const int n = pp.n_slots * pp.npix;
    if (pid == 0) {
        if (pp.Dl == 0) b.counts[kCntPad * (kCntCam + 0)] = n;
        else b.counts[kCntPad * (kCntLight + 0)] = n;
    }
    const bool live = pid < n;
    f3 dir = mk3(0.f, 0.f, 1.f);
    if (live) {
        const int slot = pid / pp.npix, pixel = pixel_of(pp, pid - slot * pp.npix);
        const int sample = pp.sample0 + slot * pp.sample_stride;
        const int row = pixel / pp.W, col = pixel - row * pp.W;
        Rng rng;
        rng.init(pp.seed, (uint32_t)pixel, (uint32_t)sample, 0u);
        const float jx = rng.u();  // Q18: first draw -> X, second -> Y
        const float jy = rng.u();
        dir = norm3(camera_dir(sc, (float)col + jx, (float)row + jy, pp.W, pp.H));
        if (!pp.common_origins) {
            b.st_o[0][pid] = mk4(ld3(sc.cam_eye), as_f32(rng.draw));  // vertex 0: dense index == path id
            b.st_T[0][pid] = f4{1.f, 1.f, 1.f, 1.f};
        }
        b.st_d[0][pid] = mk4(dir, as_f32((uint32_t)pid));
        b.path_rad[pid] = f4{0.f, 0.f, 0.f, 0.f};
        if (pp.Dl != 0) {
            b.path_rng[pid] = rng.draw;
            b.q_light[0][pid] = pid;
        }
    }
    (void)emit_ray(sc, live, ld3(sc.cam_eye), dir, live ? &b.st_hit[0][pid] : nullptr, pid, b.q_ext_cam,
                   &b.counts[kCntPad * (kCntExtCam + 0)], ap_lds, phase);
}

__global__ __launch_bounds__(kBlock) void k_lightgen(DScene sc, PassParams pp, WfBuffers b) {
    __shared__ int ap_lds[2 * (kWaves + 1)];
    int phase = 0;
    const int pid = blockIdx.x * kBlock + threadIdx.x;
    const int n = pp.n_slots * pp.npix;
    const bool live = pid < n;
    f3 lo = mk3(0.f, 0.f, 0.f), ld = mk3(0.f, 0.f, 1.f);
    int li_key = 0;
    if (live) {
        const int slot = pid / pp.npix, pixel = pixel_of(pp, pid - slot * pp.npix);
        Rng rng;
        rng.init(pp.seed, (uint32_t)pixel, (uint32_t)(pp.sample0 + slot * pp.sample_stride), b.path_rng[pid]);
        int li = (int)(rng.u() * ((float)sc.n_lights));  // kernel.cu:178
        if (li >= sc.n_lights) li = sc.n_lights - 1;      // Q17 clamp
        li_key = li;
        const f3 ldir = light_sample_dir(rng);
        lo = xyz(sc.lights[2 * li]);
        ld = norm3(ldir);                                 // Ray ctor normalises again (ray.cuh:10)
        b.l_o[pid] = mk4(lo, 0.f);
        b.l_d[pid] = mk4(ld, 0.f);
        b.l_state[pid] = mk4(ldir, as_f32((uint32_t)li));
        b.path_rng[pid] = rng.draw;
        for (int j = 0; j < pp.Dl; j++) {
            b.lv_pos[(size_t)pid * pp.Dl + j] = f4{0.f, 0.f, 0.f, -1.f};
            b.lv_T[(size_t)pid * pp.Dl + j] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    (void)emit_ray(sc, live, lo, ld, live ? &b.l_hit[pid] : nullptr, pid, b.q_ext_light, &b.counts[kCntPad * (kCntExtLight + 0)],
                   ap_lds, phase, b.q_light_key, b.q_light_key ? ((uint32_t)min(li_key, 255) << 24) | direction_cell(ld) : 0u);
}

// light sub-path vertex i (kernel.cu:184-231)
__global__ __launch_bounds__(kBlock) void k_light_shade(DScene sc, PassParams pp, WfBuffers b, int i) {
    __shared__ int ap_lds[2 * (kWaves + 1)];
    int phase = 0;
    const int n = b.counts[kCntPad * (kCntLight + i)];
    const int *qin = b.q_light[i & 1];
    int *qout = b.q_light[(i + 1) & 1];
    for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const int idx = base + threadIdx.x;
        bool cont = false, to_cam = false;
        int pid = -1;
        f3 nro = mk3(0.f, 0.f, 0.f), nrd = mk3(0.f, 0.f, 1.f);  // the bounce ray, when the walk continues
        if (idx < n) {
            pid = qin[idx];
            const HitRec h = b.l_hit[pid];
            if (h.prim >= 0) {
                const f3 o = xyz(b.l_o[pid]), d = xyz(b.l_d[pid]);
                const Surface s = surface_at(sc, o, d, h.t, h.prim);
                const f4 st = b.l_state[pid];
                const size_t vi = (size_t)pid * pp.Dl + i;
                f3 T = mk3(0.f, 0.f, 0.f);
                float pdf;
                if (i == 0) {
                    const int li = (int)as_u32(st.w);
                    T = radiance_at(sc, li, s.pos);
                    pdf = (1.0f / (float)sc.n_lights) * (1.0f / (4.0f * kPi));
                    to_cam = true;  // Q11: the camera path is traced only if the first light ray hit something
                } else {
                    pdf = st.w;  // light_pdf[i] = light_pdf[i-1] * pdf_brdf, carried from vertex i-1
                }
                if (i < pp.Dl - 1) {
                    const int slot = pid / pp.npix, pixel = pixel_of(pp, pid - slot * pp.npix);
                    Rng rng;
                    rng.init(pp.seed, (uint32_t)pixel, (uint32_t)(pp.sample0 + slot * pp.sample_stride), b.path_rng[pid]);
                    const float pd = avg3(s.kd), pm = avg3(s.ks);
                    const float rnd = rng.u();
                    if (!(rnd >= pd + pm)) {
                        const f3 in = (i == 0) ? xyz(st) : norm3(s.pos - xyz(b.lv_pos[vi - 1]));
                        const f3 Tprev = (i == 0) ? T : xyz(b.lv_T[vi - 1]);
                        f3 od;
                        float pdf_b;
                        bool ok;
                        if (rnd < pd) {
                            pdf_b = sample_diffuse(s.nrm, od, rng) * pd;
                            const float ct = dot3(s.nrm, neg3(in));
                            ok = !(ct < kEps);
                            if (ok) T = Tprev * s.kd / kPi * ct;
                        } else {
                            pdf_b = sample_mirror(s.nrm, in, od) * pm;
                            const float ct = dot3(s.nrm, neg3(in));
                            ok = !(ct < kEps);
                            if (ok) T = Tprev * s.ks;
                        }
                        if (ok) {
                            nro = s.pos + s.nrm * kEps;
                            nrd = norm3(od);
                            b.l_o[pid] = mk4(nro, 0.f);
                            b.l_d[pid] = mk4(nrd, 0.f);
                            b.l_state[pid] = mk4(xyz(st), pdf * pdf_b);
                            cont = true;
                        }
                    }
                    b.path_rng[pid] = rng.draw;
                }
                b.lv_pos[vi] = mk4(s.pos, 1.0f);
                b.lv_nrm[vi] = mk4(s.nrm, pdf);
                b.lv_T[vi] = mk4(T, 0.f);
            }
        }
        const int qi = block_append(cont, &b.counts[kCntPad * (kCntLight + i + 1)], ap_lds, phase);
        if (cont) qout[qi] = pid;
        (void)emit_ray(sc, cont, nro, nrd, cont ? &b.l_hit[pid] : nullptr, pid, b.q_ext_light,
                       &b.counts[kCntPad * (kCntExtLight + i + 1)], ap_lds, phase);
        if (i == 0) {
            const int ci = block_append(to_cam, &b.counts[kCntPad * (kCntCam + 0)], ap_lds, phase);
            if (to_cam) b.q_cam0[ci] = pid;
        }
    }
}

// camera sub-path vertex i (kernel.cu:239-303)
// ENVL: the instantiation that also samples the cube map as a light (use_envmap == 2).  A template parameter, not a run-time
// flag: as a flag the extra live state cost the default kernel 57 % (5.3 -> 8.3 ms per launch on the 1080p frame).
template <bool ENVL>
__global__ __launch_bounds__(kBlock, 4) void k_shade(DScene sc, PassParams pp, WfBuffers b, int i) {
    __shared__ int ap_lds[2 * (kWaves + 1)];
    __shared__ int ap3_lds[2 * 3 * kWaves];
    int phase = 0, phase3 = 0;
    int n_trivial = 0;  // shadow rays answered here; one atomic per wave at the end
    const int n = b.counts[kCntPad * (kCntCam + i)];
    const int par = i & 1;
    const bool by_id = (i == 0) && (pp.Dl != 0);  // bidirectional vertex 0: the alive set is a list of path ids
    // the cube map as a light (build-defined, oracle.cpp trace_sample): two draws and one shadow ray per vertex; a ray that
    // leaves the scene then collects the map only at vertex 0 and behind a mirror bounce (bit 31 of the path's draw word)
    const bool env_light = ENVL && env_light_usable(sc);
    for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const int idx = base + threadIdx.x;
        bool cont = false;
        int pid = -1, j = 0;
        uint32_t mask = 0u, draw = 0u;
        f3 nro = mk3(0.f, 0.f, 0.f), nrd = mk3(0.f, 0.f, 1.f);  // the next camera ray, when the path continues
        f3 Tc = mk3(0.f, 0.f, 0.f);
        float pc = 1.0f;
        // the direct connection (shadow slot 0) stays in registers until it is known whether it needs the extend kernel
        bool direct = false;
        f3 dr_d = mk3(0.f, 0.f, 1.f), dr_target = mk3(0.f, 0.f, 0.f), dr_contrib = mk3(0.f, 0.f, 0.f);
        if (idx < n) {
            j = by_id ? b.q_cam0[idx] : idx;
            const HitRec h = b.st_hit[par][j];
            const bool v0 = (i == 0) && pp.common_origins;  // vertex 0 of the simple integrator: eye, throughput 1, two draws made
            const f4 o4 = v0 ? mk4(ld3(sc.cam_eye), as_f32(2u)) : b.st_o[par][j], d4 = b.st_d[par][j];
            const f3 o = xyz(o4), d = xyz(d4);
            pid = (int)as_u32(d4.w);
            const f4 Tp = v0 ? f4{1.f, 1.f, 1.f, 1.f} : b.st_T[par][j];
            Tc = xyz(Tp);
            pc = Tp.w;
            if (h.prim < 0) {
                if (pp.use_env && (!env_light || i == 0 || (as_u32(o4.w) >> 31) != 0u)) {  // build-defined extension: radiance of the cube map on a miss
                    f4 r = b.path_rad[pid];
                    const f3 add = (Tc / pc) * env_lookup(sc, d);
                    r.x = r.x + add.x; r.y = r.y + add.y; r.z = r.z + add.z;
                    b.path_rad[pid] = r;
                }
            } else {
                const Surface s = surface_at(sc, o, d, h.t, h.prim);
                const size_t sbase = (size_t)j * pp.K;
                {  // direct connection to lights[0] (Q13); the shadow ray starts AT the light (Q10)
                    const f3 lp = xyz(ldk4(sc.lights, 0));
                    const f3 ld = norm3(s.pos - lp);
                    const float ct = dot3(s.nrm, neg3(ld));
                    if (ct > kEps) {
                        const f3 lr = radiance_at(sc, 0, s.pos);
                        const f3 brdf = s.kd / kPi;
                        dr_contrib = (Tc / pc) * (lr * brdf * ct);
                        dr_d = norm3(ld);
                        dr_target = s.pos;
                        direct = true;
                    }
                }
                for (int jj = 0; jj < pp.Dl; jj++) {  // connections to the stored light vertices (Q15, no MIS)
                    const size_t vi = (size_t)pid * pp.Dl + jj;
                    const f4 lp4 = b.lv_pos[vi];
                    if (lp4.w < 0.0f) break;
                    const f3 LT = xyz(b.lv_T[vi]);
                    const f4 ln4 = b.lv_nrm[vi];
                    const f3 dir = norm3(xyz(lp4) - s.pos);
                    const float dist = len3(xyz(lp4) - s.pos);
                    const float cc = dot3(s.nrm, dir), cl = dot3(xyz(ln4), neg3(dir));
                    if (cc < kEps || cl < kEps) continue;
                    const float G = cc * cl / (dist * dist);
                    const f3 brdf = s.kd / kPi;
                    const f3 cw = Tc * brdf * cc / pc;
                    const f3 lw = LT / ln4.w;
                    const f3 contrib = lw * G * cw;
                    // Q14: a stored vertex whose throughput is zero adds exactly 0 and needs no ray -- unless its pdf is 0
                    // too (SampleDiffuse returned z / pi == 0): then kernel.cu:277 is 0/0 and the NaN must be kept
                    if (contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f) continue;
                    const size_t si = sbase + 1 + jj;
                    b.sh_o[si] = mk4(s.pos + s.nrm * kEps, 0.f);
                    b.sh_d[si] = mk4(norm3(dir), 0.f);
                    b.sh_target[si] = mk4(xyz(lp4), 1.f);
                    b.sh_contrib[si] = mk4(contrib, as_f32((uint32_t)pid));
                    mask |= 1u << (1 + jj);
                }
                if (i + 1 < pp.Dc || env_light) {  // next vertex (Q16); nothing downstream reads it after the last vertex
                    const int slot = pid / pp.npix, pixel = pixel_of(pp, pid - slot * pp.npix);
                    Rng rng;
                    rng.init(pp.seed, (uint32_t)pixel, (uint32_t)(pp.sample0 + slot * pp.sample_stride),
                             (by_id ? b.path_rng[pid] : as_u32(o4.w)) & 0x7fffffffu);
                    if (env_light) {  // two draws, always; the contribution waits for its shadow ray in slot 1 + Dl
                        const float u1 = rng.u(), u2 = rng.u();
                        f3 edir, erad;
                        float sa, pmf;
                        if (env_light_sample(sc, u1, u2, edir, sa, pmf, erad)) {
                            const f3 wdir = norm3(edir);
                            const float ct = dot3(s.nrm, wdir);
                            if (ct > kEps) {
                                const f3 brdf = s.kd / kPi;
                                const f3 contrib = ((Tc / pc) * (erad * brdf * ct)) * (sa / pmf);
                                const size_t si = sbase + 1 + pp.Dl;
                                b.sh_o[si] = mk4(s.pos + s.nrm * kEps, 0.f);
                                b.sh_d[si] = mk4(norm3(wdir), 0.f);
                                b.sh_target[si] = mk4(mk3(0.f, 0.f, 0.f), 2.f);
                                b.sh_contrib[si] = mk4(contrib, as_f32((uint32_t)pid));
                                mask |= 1u << (1 + pp.Dl);
                            }
                        }
                    }
                  if (i + 1 < pp.Dc) {
                    const float pd = avg3(s.kd), pm = avg3(s.ks);
                    const float rnd = rng.u();
                    f3 od;
                    if (rnd < pd) {
                        const float pdf = sample_diffuse(s.nrm, od, rng);
                        const float ct = dot3(s.nrm, od);
                        if (!(ct < kEps)) {
                            Tc = Tc * s.kd / kPi * ct;
                            pc = pc * pdf * pd;
                            cont = true;
                        }
                    } else if (rnd < pd + pm) {
                        const float pdf = sample_mirror(s.nrm, d, od);
                        Tc = Tc * s.ks;
                        pc = pc * pdf * pm;
                        cont = true;
                        if (env_light) rng.draw |= 0x80000000u;  // behind a mirror bounce a miss still collects the map
                    }
                    draw = rng.draw;
                    if (cont) {
                        nro = s.pos + s.nrm * kEps;
                        nrd = norm3(od);
                    }
                  }
                }
            }
        }
        {
            // the two rays this vertex makes: the next camera ray and the direct connection.  Their sphere / plane
            // answers are taken here (convergent lanes); a ray that misses every mesh's root cull box is finished
            // (52 % of them in the box-room scenes), the others are queued for the extend kernel.
            const f3 lp = xyz(ldk4(sc.lights, 0));
            HitRec ch, sh;
            ch.t = sh.t = -1.0f;
            ch.prim = sh.prim = -1;
            bool cam_extend = false, need_extend = false;
            {   // (evaluated on every lane: a lane without one of the two rays works on a placeholder and drops the answer)
                HitRec c2, s2;
                quadrics_closest2(sc, nro, nrd, lp, dr_d, c2.t, c2.prim, s2.t, s2.prim);
                if (cont) {
                    ch = c2;
                    cam_extend = !misses_every_mesh(sc, nro, nrd);
                }
                if (direct) {
                    sh = s2;
                    need_extend = !misses_every_mesh(sc, lp, dr_d);
                }
            }
            // a continuing path takes its rank among the block's continuing paths as its index at the next vertex
            const Idx3 at = block_append3(cont, cam_extend, need_extend, &b.counts[kCntPad * (kCntCam + i + 1)],
                                          &b.counts[kCntPad * (kCntExtCam + i + 1)], &b.counts[kCntPad * (kCntShadow + i)],
                                          ap3_lds, phase3);
            if (cont) {
                st_nt(&b.st_o[par ^ 1][at.a], mk4(nro, as_f32(draw)));
                st_nt(&b.st_d[par ^ 1][at.a], mk4(nrd, as_f32((uint32_t)pid)));
                st_nt(&b.st_T[par ^ 1][at.a], mk4(Tc, pc));
                st_nt(&b.st_hit[par ^ 1][at.a], ch);
            }
            if (cam_extend) b.q_ext_cam[at.b] = at.a;
            const int qi = at.c;
            // where the direct connection's record goes: with one shadow ray per path (simple integrator) at its RANK in the
            // shadow queue -- the records of a block are one dense run of full lines, the queue is the identity and
            // k_resolve_queue streams them; with connection rays besides (K > 1) in the path's slot group j * K
            const size_t si = pp.K == 1 ? (size_t)qi : (size_t)j * pp.K;
            if (need_extend) {
                b.q_shadow[qi] = (int)si;
                if (b.q_shadow_key) b.q_shadow_key[qi] = direction_cell(dr_d);
                if (!pp.common_origins) b.sh_o[si] = mk4(lp, 0.f);
                st_nt(&b.sh_d[si], mk4(dr_d, 0.f));
                st_nt(&b.sh_target[si], mk4(dr_target, 0.f));
                st_nt(&b.sh_contrib[si], mk4(dr_contrib, as_f32((uint32_t)pid)));
                st_nt(&b.sh_hit[si], sh);
                mask |= 1u;
            } else if (direct) {
                n_trivial++;
                if (shadow_visible(lp, dr_d, sh.t, mk4(dr_target, 0.f))) {
                    f4 r = b.path_rad[pid];
                    r.x = r.x + dr_contrib.x; r.y = r.y + dr_contrib.y; r.z = r.z + dr_contrib.z;
                    b.path_rad[pid] = r;
                }
            }
            if (idx < n && pp.K > 1) b.sh_mask[j] = mask;  // (K == 1: k_resolve_queue walks the queue instead)
        }
        if (pp.K > 1) {  // connection rays to the light vertices: sphere / plane answers, then ONE append for all of them
            uint32_t queued = 0u;
            for (int k = 1; k < pp.K; k++) {
                if (!((mask >> k) & 1u)) continue;
                const size_t si = (size_t)j * pp.K + k;
                const f3 so = xyz(b.sh_o[si]), sd = xyz(b.sh_d[si]);
                HitRec h;
                quadrics_closest(sc, so, sd, h.t, h.prim);
                b.sh_hit[si] = h;
                if (misses_every_mesh(sc, so, sd)) n_trivial++;
                else queued |= 1u << k;
            }
            int at = block_append_n(__popc(queued), &b.counts[kCntPad * (kCntShadow + i)], ap_lds, phase);
            for (int k = 1; k < pp.K; k++) {
                if (!((queued >> k) & 1u)) continue;
                b.q_shadow[at] = (int)((size_t)j * pp.K + k);
                if (b.q_shadow_key) b.q_shadow_key[at] = kKeyLast;
                at++;
            }
        }
    }
    n_trivial = wave_sum_i(n_trivial);
    if ((threadIdx.x & 63u) == 0 && n_trivial > 0) atomicAdd(&b.counts[kCntPad * (kCntShadowTriv + i)], n_trivial);
}

// visibility + accumulation for vertex i (kernel.cu:249-256, 268-279).  HBM-bound: it reads as little as it can -- the
// ray of a direct connection is recomputed from its target (same operations as k_shade: origin = lights[0], direction
// = normalised twice, kernel.cu:245-248 + ray.cuh:10), the path id rides in contrib.w, and the radiance is only
// touched by paths with a visible contribution: 28 B per path + 48 B per visible one instead of 124 B per path.
__global__ __launch_bounds__(kBlock) void k_resolve(DScene sc, PassParams pp, WfBuffers b, int i) {
    const int n = b.counts[kCntPad * (kCntCam + i)];
    const bool by_id = (i == 0) && (pp.Dl != 0);
    const f3 lp = xyz(ldk4(sc.lights, 0));
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < n; idx += gridDim.x * kBlock) {
        const int j = by_id ? b.q_cam0[idx] : idx;
        const uint32_t mask = b.sh_mask[j];
        if (mask == 0u) continue;
        f4 r = f4{0.f, 0.f, 0.f, 0.f};
        int pid = -1;
        for (int k = 0; k < pp.K; k++) {
            if (!((mask >> k) & 1u)) continue;
            const size_t si = (size_t)j * pp.K + k;
            const HitRec h = b.sh_hit[si];
            const f4 tg = b.sh_target[si];
            f3 o, d;
            if (k == 0) {
                o = lp;
                d = norm3(norm3(xyz(tg) - lp));
            } else {
                o = xyz(b.sh_o[si]);
                d = xyz(b.sh_d[si]);
            }
            if (shadow_visible(o, d, h.t, tg)) {
                const f4 c = b.sh_contrib[si];
                if (pid < 0) {
                    pid = (int)as_u32(c.w);
                    r = b.path_rad[pid];
                }
                r.x = r.x + c.x; r.y = r.y + c.y; r.z = r.z + c.z;
            }
        }
        if (pid >= 0) b.path_rad[pid] = r;
    }
}

// The same for the simple integrator (one shadow ray per path and vertex, so no order of additions to keep): walk the
// shadow QUEUE instead of every alive path -- no per-path mask to write and read, no thread for the 45 % of the paths
// whose connection was answered on the spot.
__global__ __launch_bounds__(kBlock) void k_resolve_queue(DScene sc, WfBuffers b, int i) {
    const int n = b.counts[kCntPad * (kCntShadow + i)];
    const f3 lp = xyz(ldk4(sc.lights, 0));
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < n; idx += gridDim.x * kBlock) {
        const int si = idx;  // K == 1: record index == queue rank (k_shade)
        const HitRec h = b.sh_hit[si];
        const f4 tg = b.sh_target[si];
        if (shadow_visible(lp, norm3(norm3(xyz(tg) - lp)), h.t, tg)) {
            const f4 c = b.sh_contrib[si];
            const int pid = (int)as_u32(c.w);
            f4 r = b.path_rad[pid];
            r.x = r.x + c.x; r.y = r.y + c.y; r.z = r.z + c.z;
            b.path_rad[pid] = r;
        }
    }
}

// image[id] += trace / nSamples, samples in increasing order (kernel.cu:339-343)
// A sample whose radiance is not finite (kept quirk, see DESIGN.md 2: the reference's arithmetic can produce one) is
// counted; with drop_nonfinite (MIPT_FLAG_DROP_NONFINITE, an opt-in deviation) it adds nothing to the pixel.
__global__ __launch_bounds__(kBlock) void k_accumulate(PassParams pp, WfBuffers b, float *accum, int drop_nonfinite) {
    const int pixel = blockIdx.x * kBlock + threadIdx.x;
    int bad = 0;
    if (pixel < pp.npix) {
        const int pos = position_of(pp, pixel);
        float ax = accum[3 * (size_t)pixel], ay = accum[3 * (size_t)pixel + 1], az = accum[3 * (size_t)pixel + 2];
        const float ns = (float)pp.spp;
        for (int s = 0; s < pp.n_slots; s++) {
            const f4 r = b.path_rad[(size_t)s * pp.npix + pos];
            const bool finite = (fabsf(r.x) < __builtin_inff()) & (fabsf(r.y) < __builtin_inff()) & (fabsf(r.z) < __builtin_inff());
            bad += finite ? 0 : 1;
            if (!finite && drop_nonfinite) continue;
            ax = ax + r.x / ns;
            ay = ay + r.y / ns;
            az = az + r.z / ns;
        }
        accum[3 * (size_t)pixel] = ax;
        accum[3 * (size_t)pixel + 1] = ay;
        accum[3 * (size_t)pixel + 2] = az;
    }
    if (__ballot(bad != 0) != 0ull) {  // rare: one atomic per wave that saw one
        bad = wave_sum_i(bad);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&b.counts[kCntPad * kCntNonFinite], bad);
    }
}

__global__ void k_tally(WfBuffers b, int Dc, int Dl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long cam = 0, sh = 0, li = 0, ext = 0;
    for (int i = 0; i < Dc; i++) { cam += (unsigned long long)b.counts[kCntPad * (kCntCam + i)]; sh += (unsigned long long)b.counts[kCntPad * (kCntShadow + i)] + (unsigned long long)b.counts[kCntPad * (kCntShadowTriv + i)]; }
    for (int i = 0; i < Dl; i++) li += (unsigned long long)b.counts[kCntPad * (kCntLight + i)];
    for (int i = 0; i < Dc; i++) ext += (unsigned long long)b.counts[kCntPad * (kCntExtCam + i)] + (unsigned long long)b.counts[kCntPad * (kCntShadow + i)];
    for (int i = 0; i < Dl; i++) ext += (unsigned long long)b.counts[kCntPad * (kCntExtLight + i)];
    b.totals[0] += cam;
    b.totals[1] += sh;
    b.totals[2] += li;
    b.totals[3] += ext;
    b.totals[4] += (unsigned long long)b.counts[kCntPad * kCntNonFinite];
}

__global__ __launch_bounds__(kBlock) void k_env_lookup(DScene sc, const float *dirs, int64_t n, float *rgb) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const f3 c = env_lookup(sc, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

#define WF_TRY(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return -2;                                                            \
        }                                                                         \
    } while (0)

}  // namespace

int wavefront_env_lookup(DeviceCtx &dc, const float *d_dirs, int64_t n, float *d_rgb, std::string &err) {
    hipLaunchKernelGGL(k_env_lookup, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, dc.stream, dc.view, d_dirs,
                       n, d_rgb);
    WF_TRY(hipGetLastError());
    WF_TRY(hipStreamSynchronize(dc.stream));
    return 0;
}

int wavefront_render(const HostScene &hs, DeviceCtx &dc, const mipt_render_config &cfg, float *d_accum, mipt_stats *st,
                     std::string &err) {
    if (st) memset(st, 0, sizeof(*st));
    if (!hs.cam_set) { err = "camera not set"; return -1; }
    if (hs.lights.empty()) { err = "at least one light is required (kernel.cu:244)"; return -1; }
    if (cfg.width <= 0 || cfg.height <= 0 || (int64_t)cfg.width * cfg.height > (1 << 26)) { err = "bad image size"; return -1; }
    if (cfg.spp <= 0) { err = "spp must be positive"; return -1; }
    if (cfg.cam_depth < 0 || cfg.cam_depth > kMaxPathDepth || cfg.light_depth < 0 || cfg.light_depth > kMaxPathDepth) {
        err = "path depth out of range (0..16)";
        return -1;
    }
    if (cfg.use_envmap < 0 || cfg.use_envmap > 2) { err = "use_envmap must be 0 (off), 1 (radiance on a miss) or 2 (... and the map sampled as a light)"; return -1; }
    if (cfg.use_envmap && !hs.env.present()) { err = "use_envmap set but the scene has no environment map"; return -1; }
    const int stride = cfg.sample_stride > 0 ? cfg.sample_stride : 1;
    int64_t n_samples = 0;
    if (cfg.sample_end > cfg.sample_begin) n_samples = ((int64_t)cfg.sample_end - cfg.sample_begin + stride - 1) / stride;
    const int npix = cfg.width * cfg.height;
    const int Dc = cfg.cam_depth, Dl = cfg.light_depth, K = 1 + Dl + (cfg.use_envmap == 2 ? 1 : 0);
    WF_TRY(hipSetDevice(dc.device));
    if (n_samples == 0 || Dc == 0) return 0;
    int S = cfg.samples_per_pass;
    // auto: ~128 M paths per pass (28 GB of path state for simple PT, 10 % of the HBM).  The persistent extend kernel
    // holds ~330 k rays at a time and every launch pays a ramp-up and a tail in which its waves run dry one by one; at
    // 1080p 16 / 32 / 64 / 128 samples of every pixel per pass give 476 / 490 / 497 / 500 Msamples/s (1 per pass: 122)
    if (S <= 0) S = (int)std::max<int64_t>(1, ((int64_t)1 << 27) / npix);
    S = (int)std::min<int64_t>(std::min<int64_t>(S, n_samples), std::max<int64_t>(1, ((int64_t)1 << 28) / ((int64_t)npix * K)));
    const size_t P = (size_t)S * npix;

    size_t sizes[36] = {0};
    for (int k = 0; k < 6; k++) sizes[k] = P * sizeof(f4);         // st_o[2], st_d[2], st_T[2]
    sizes[6] = sizes[7] = P * sizeof(HitRec);                       // st_hit[2]
    sizes[8] = P * sizeof(f4);                                      // path_rad
    sizes[9] = P * 4;                                               // sh_mask
    sizes[10] = sizes[11] = sizes[12] = sizes[13] = P * K * sizeof(f4);  // sh_o, sh_d, sh_target, sh_contrib
    sizes[14] = P * K * sizeof(HitRec);                             // sh_hit
    sizes[15] = P * K * 4;                                          // q_shadow
    sizes[16] = P * 4;                                              // q_ext_cam
    sizes[17] = (size_t)kCntSlots * kCntPad * 4 + 64;               // counts + ray cursors + totals
    if (Dl > 0) {
        sizes[18] = sizes[19] = sizes[20] = P * Dl * sizeof(f4);    // lv_pos, lv_nrm, lv_T
        sizes[21] = sizes[22] = sizes[24] = P * sizeof(f4);         // l_o, l_d, l_state
        sizes[23] = P * sizeof(HitRec);                             // l_hit
        sizes[25] = sizes[26] = sizes[27] = sizes[28] = sizes[29] = P * 4;  // path_rng, q_cam0, q_light[2], q_ext_light
    }
    // direction binning of the shadow rays (simple integrator, persistent kernel): keys in / out, sorted queue, scratch
    // (not worth its ~1 ms per queue when the trees are a few nodes deep: default scene, 36 triangles)
    const bool sort_shadow = !dc.extend_simple && dc.shadow_sort_from >= 0 && dc.shadow_sort_from < Dc && hs.total_tris >= 4096;
    const bool sort_light = sort_shadow && Dl > 0;  // first light rays: binned by (light, direction)
    if (sort_shadow) {
        // keys in / intermediate keys / binned queue / intermediate values (queue_sort.h), + the small scan scratch
        sizes[30] = sizes[31] = sizes[32] = sizes[33] = P * K * 4;
        sizes[35] = (size_t)kSortScratchInts * sizeof(int);
        if (sort_light) sizes[34] = P * 4;
    }
    static_assert(sizeof(dc.wf) / sizeof(dc.wf[0]) >= 36, "wf buffer table too small");
    for (int i = 0; i < 36; i++)
        if (sizes[i]) WF_TRY(dc.wf[i].reserve(sizes[i]));
    WfBuffers b;
    memset(&b, 0, sizeof(b));
    for (int k = 0; k < 2; k++) {
        b.st_o[k] = dc.wf[k].as<f4>(); b.st_d[k] = dc.wf[2 + k].as<f4>(); b.st_T[k] = dc.wf[4 + k].as<f4>();
        b.st_hit[k] = dc.wf[6 + k].as<HitRec>();
    }
    b.path_rad = dc.wf[8].as<f4>(); b.sh_mask = dc.wf[9].as<uint32_t>();
    b.sh_o = dc.wf[10].as<f4>(); b.sh_d = dc.wf[11].as<f4>(); b.sh_target = dc.wf[12].as<f4>(); b.sh_contrib = dc.wf[13].as<f4>();
    b.sh_hit = dc.wf[14].as<HitRec>(); b.q_shadow = dc.wf[15].as<int>(); b.q_ext_cam = dc.wf[16].as<int>();
    b.counts = dc.wf[17].as<int>();
    b.totals = reinterpret_cast<unsigned long long *>(dc.wf[17].as<char>() + (size_t)kCntSlots * kCntPad * 4 + 8);
    b.lv_pos = dc.wf[18].as<f4>(); b.lv_nrm = dc.wf[19].as<f4>(); b.lv_T = dc.wf[20].as<f4>();
    b.l_o = dc.wf[21].as<f4>(); b.l_d = dc.wf[22].as<f4>(); b.l_hit = dc.wf[23].as<HitRec>(); b.l_state = dc.wf[24].as<f4>();
    b.path_rng = dc.wf[25].as<uint32_t>(); b.q_cam0 = dc.wf[26].as<int>();
    b.q_light[0] = dc.wf[27].as<int>(); b.q_light[1] = dc.wf[28].as<int>(); b.q_ext_light = dc.wf[29].as<int>();
    b.q_shadow_key = sort_shadow ? dc.wf[30].as<uint32_t>() : nullptr;
    b.q_light_key = sort_light ? dc.wf[34].as<uint32_t>() : nullptr;

    dc.events.reset();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ext_ev, shade_ev;
    hipStream_t sm = dc.stream;
    WF_TRY(hipMemsetAsync(dc.wf[17].p, 0, sizes[17], sm));
    const int mode = extend_mode(cfg.flags);
    const bool count_mode = mode == kCountRef;
    if (mode != kProd) WF_TRY(hipMemsetAsync(dc.counters.p, 0, kCtrCursor, sm));
    hipEvent_t ev_begin = dc.events.next(), ev_end = dc.events.next();
    WF_TRY(hipEventRecord(ev_begin, sm));

    PassParams pp;
    pp.n_slots = S;
    // Bin a ray queue by the top bits of its keys into wf[32] (queue_sort.h); `queue` then points at the binned copy.  The
    // queue's length never leaves the device.
    auto sort_queue = [&](const uint32_t *keys, const int *queue_in, int count_slot, unsigned key_bits, const int *&queue) {
        launch_queue_sort(sm, dc.num_cus, keys, queue_in, &b.counts[kCntPad * count_slot], key_bits, dc.wf[31].as<uint32_t>(),
                          dc.wf[33].as<int>(), dc.wf[35].as<int>(), dc.wf[32].as<int>());
        queue = dc.wf[32].as<int>();
    };
    int n_launch = 0;
    auto extend = [&](const f4 *ro, const f4 *rd, const int *queue, const int *count, HitRec *hits, size_t n_max, bool coherent_queue,
                      int common_origin = 0) {
        hipEvent_t a = dc.events.next(), e = dc.events.next();
        (void)hipEventRecord(a, sm);
        launch_extend(dc, mode, ro, rd, queue, count, 0, (int64_t)n_max, hits, &b.counts[kCntPad * (kCntWork + 8 * (n_launch++))],
                      dc.counters.as<unsigned long long>(), coherent_queue, common_origin);
        (void)hipEventRecord(e, sm);
        ext_ev.emplace_back(a, e);
    };
    const int path_grid = (int)std::min<size_t>((P + kBlock - 1) / kBlock, (size_t)dc.num_cus * 16);

    pp.W = cfg.width; pp.H = cfg.height; pp.npix = npix; pp.spp = cfg.spp; pp.sample_stride = stride;
    pp.Dc = Dc; pp.Dl = Dl; pp.K = K; pp.use_env = cfg.use_envmap ? 1 : 0; pp.env_light = cfg.use_envmap == 2 ? 1 : 0; pp.seed = cfg.seed;
    pp.tiles_x = (cfg.width % 8 == 0 && cfg.height % 8 == 0) ? cfg.width / 8 : 0;
    pp.common_origins = (K == 1 && !dc.extend_simple) ? 1 : 0;

    for (int64_t done = 0; done < n_samples; done += S) {
        pp.n_slots = (int)std::min<int64_t>(S, n_samples - done);
        pp.sample0 = cfg.sample_begin + (int)done * stride;
        const size_t np = (size_t)pp.n_slots * npix;
        const unsigned full_grid = (unsigned)((np + kBlock - 1) / kBlock);
        WF_TRY(hipMemsetAsync(b.counts, 0, (size_t)kCntSlots * kCntPad * 4, sm));
        n_launch = 0;
        hipLaunchKernelGGL(k_raygen, dim3(full_grid), dim3(kBlock), 0, sm, dc.view, pp, b);
        if (Dl > 0) {
            hipLaunchKernelGGL(k_lightgen, dim3(full_grid), dim3(kBlock), 0, sm, dc.view, pp, b);
            for (int i = 0; i < Dl; i++) {
                const int *light_queue = b.q_ext_light;
                if (sort_light && i == 0) {
                    unsigned light_bits = 0;
                    while ((1 << light_bits) < std::min<int>((int)hs.lights.size(), 256)) light_bits++;
                    sort_queue(b.q_light_key, b.q_ext_light, kCntExtLight + 0, kShadowKeyBits + light_bits, light_queue);
                }
                extend(b.l_o, b.l_d, light_queue, &b.counts[kCntPad * (kCntExtLight + i)], b.l_hit, np, light_queue != b.q_ext_light);
                hipLaunchKernelGGL(k_light_shade, dim3(path_grid), dim3(kBlock), 0, sm, dc.view, pp, b, i);
            }
        }
        for (int i = 0; i < Dc; i++) {
            extend((i == 0 && pp.common_origins) ? nullptr : b.st_o[i & 1], b.st_d[i & 1], b.q_ext_cam,
                   &b.counts[kCntPad * (kCntExtCam + i)], b.st_hit[i & 1], np, i == 0, (i == 0 && pp.common_origins) ? 1 : 0);
            {
                hipEvent_t a = dc.events.next(), e = dc.events.next();
                (void)hipEventRecord(a, sm);
                if (pp.env_light) hipLaunchKernelGGL(k_shade<true>, dim3(path_grid), dim3(kBlock), 0, sm, dc.view, pp, b, i);
                else hipLaunchKernelGGL(k_shade<false>, dim3(path_grid), dim3(kBlock), 0, sm, dc.view, pp, b, i);
                (void)hipEventRecord(e, sm);
                shade_ev.emplace_back(a, e);
            }
            const int *shadow_queue = b.q_shadow;
            if (sort_shadow && i >= dc.shadow_sort_from)
                sort_queue(b.q_shadow_key, b.q_shadow, kCntShadow + i, kShadowKeyBits, shadow_queue);
            const bool binned = shadow_queue != b.q_shadow;
            if (K == 1 && !binned) shadow_queue = nullptr;  // K == 1: the records sit at their queue rank, the unsorted queue is the identity
            extend(pp.common_origins ? nullptr : b.sh_o, b.sh_d, shadow_queue, &b.counts[kCntPad * (kCntShadow + i)], b.sh_hit, np * K,
                   K == 1 && (i == 0 || binned),  // (K > 1: mostly connection rays, in path order)
                   pp.common_origins ? 2 : 0);
            if (K == 1) hipLaunchKernelGGL(k_resolve_queue, dim3(path_grid), dim3(kBlock), 0, sm, dc.view, b, i);
            else hipLaunchKernelGGL(k_resolve, dim3(path_grid), dim3(kBlock), 0, sm, dc.view, pp, b, i);
            // the shadow queue is refilled from slot 0 by the next vertex: its count lives in its own slot
        }
        hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, sm, pp, b, d_accum,
                           (cfg.flags & MIPT_FLAG_DROP_NONFINITE) ? 1 : 0);
        hipLaunchKernelGGL(k_tally, dim3(1), dim3(64), 0, sm, b, Dc, Dl);
        WF_TRY(hipGetLastError());
    }
    WF_TRY(hipEventRecord(ev_end, sm));
    WF_TRY(hipStreamSynchronize(sm));
#if defined(MIPT_PHASE_STATS) || defined(MIPT_PHASE_CLOCKS)
    if (!count_mode) dump_phase_stats(dc, "render");
#endif
    if (getenv("MIPT_TIMING")) {  // queue lengths of the last pass and the duration of every extend launch, in launch order
        std::vector<int> cnt((size_t)kCntTotal * kCntPad);
        WF_TRY(hipMemcpy(cnt.data(), b.counts, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "[mipt] last pass, per camera vertex: alive / camera rays through extend / shadow rays through extend / answered on the spot\n");
        for (int i = 0; i < Dc; i++)
            fprintf(stderr, "[mipt]   vertex %d: %d / %d / %d / %d\n", i, cnt[(size_t)kCntPad * (kCntCam + i)], cnt[(size_t)kCntPad * (kCntExtCam + i)],
                    cnt[(size_t)kCntPad * (kCntShadow + i)], cnt[(size_t)kCntPad * (kCntShadowTriv + i)]);
        fprintf(stderr, "[mipt] extend launches (ms):");
        for (size_t k = ext_ev.size() >= (size_t)n_launch ? ext_ev.size() - (size_t)n_launch : 0; k < ext_ev.size(); k++) {
            float m = 0.f;
            (void)hipEventElapsedTime(&m, ext_ev[k].first, ext_ev[k].second);
            fprintf(stderr, " %.2f", m);
        }
        fprintf(stderr, "\n");
    }
    if (st) {
        unsigned long long tot[5] = {0, 0, 0, 0, 0};
        WF_TRY(hipMemcpy(tot, b.totals, sizeof(tot), hipMemcpyDeviceToHost));
        float ms = 0.f;
        WF_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        double ms_ext = 0.0;
        for (auto &pr : ext_ev) {
            float m = 0.f;
            WF_TRY(hipEventElapsedTime(&m, pr.first, pr.second));
            ms_ext += m;
        }
        st->rays_camera = (int64_t)tot[0];
        st->rays_shadow = (int64_t)tot[1];
        st->rays_light = (int64_t)tot[2];
        st->rays_extended = (int64_t)tot[3];
        st->nonfinite_samples = (int64_t)tot[4];
        st->rays_total = st->rays_camera + st->rays_shadow + st->rays_light;
        st->samples = n_samples * npix;
        st->extend_launches = (int64_t)ext_ev.size();
        st->ms_total = ms;
        st->ms_extend = ms_ext;
        st->ms_other = ms - ms_ext;
        double ms_shade = 0.0;
        for (auto &pr : shade_ev) {
            float m = 0.f;
            WF_TRY(hipEventElapsedTime(&m, pr.first, pr.second));
            ms_shade += m;
        }
        st->shade_launches = (int64_t)shade_ev.size();
        st->ms_shade = ms_shade;
        if (count_mode) {
            unsigned long long c[8] = {0};
            WF_TRY(hipMemcpy(c, dc.counters.p, sizeof(c), hipMemcpyDeviceToHost));
            st->ctr_nodes = (int64_t)c[0]; st->ctr_pops = (int64_t)c[1]; st->ctr_refs = (int64_t)c[2];
            st->ctr_tris = (int64_t)c[3]; st->ctr_hits = (int64_t)c[4]; st->ctr_culled = (int64_t)c[5];
        }
        if (mode == kCountWork) fill_work_counters(dc, st);
    }
    return 0;
}

}  // namespace mipt
