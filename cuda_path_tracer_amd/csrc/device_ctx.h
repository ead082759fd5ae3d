// device_ctx.h -- per-device state of a committed scene (internal to libmipt).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#include "scene_host.h"

namespace mipt {

struct HitRec {
    float t;
    int32_t prim;
};

// A growable device allocation reused across calls (no per-call hipMalloc once warm).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t next() {
        if (used == ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() { used = 0; }
    void destroy() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
        used = 0;
    }
};

// Scheduling knobs of the persistent extend kernel (wave-uniform kernel arguments; MIPT_TUNE="refill,batch,cap").
struct ExtendTuning {
    int refill_lanes = 16;      // refill a wave once this many lanes are idle (or all of them)
    int leaf_batch_lanes = 16;  // scan leaves once this many lanes wait at one (or fewer lanes than this still traverse)
    int step_cap = 4;           // node steps per lane per round before lanes re-converge
    int pool1_below = 40;       // fewer lanes than this waiting at a leaf: the sphere pass is pooled over the wave too (kernels_extend.h)
    int common_origin = 0;      // per launch: 0 = rays carry their origins, 1 = all start at the eye, 2 = all start at lights[0]
    int n_heads = 8;            // ray cursors of the launch (8: one per XCD; 1: a single word, for A/B timing -- MIPT_HEADS)
    // (defaults from the sweeps of scripts/tune_closest.py on MI355X; the optimum is flat within ~1 % around them.  With the
    //  leaf scan pooled over the wave a scan is cheap at any count, so lanes wait less: round 1 shipped 32 / 28)
};

// Layout of DeviceCtx::counters (bytes): what mipt_trace_closest* uses outside a wavefront pass.
constexpr size_t kCtrRef = 0;        // 8 x u64: counters of the reference walk (k_extend<kCountRef>)
constexpr size_t kCtrPhase = 128;    // 24 x u64: developer build, per-phase lane use
constexpr size_t kCtrWork = 320;     // 8 x u64: what the production kernel read (k_extend<kCountWork>)
constexpr size_t kCtrProbe = 384;    // 2 x int: coherence probe of a closest-hit batch (k_make_rays): waves sampled, waves found coherent
constexpr size_t kCtrCursor = 1024;  // 8 ray cursors of the persistent kernel, 256 B apart (one per XCD)
constexpr size_t kCounterBytes = kCtrCursor + 8 * 256;

struct DeviceCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t caller_stream = nullptr;  // mipt_scene_set_caller_stream: what the *_device entry points order themselves after
    hipEvent_t ev_entry = nullptr;        // recorded on caller_stream at entry, waited for by `stream`
    void *arena = nullptr;
    DScene view;
    int num_cus = 256;
    int extend_blocks_per_cu = 4;  // resident blocks of the persistent extend kernel
    ExtendTuning tune;             // MIPT_TUNE: bounce-ray queues and mipt_trace_closest (order of the rays unknown)
    // MIPT_TUNE_COHERENT: queues in a coherent order (primary rays, binned shadow / light rays).  Their lanes finish
    // together, so refilling only when nearly the whole wave is idle keeps neighbouring rays in one wave: +3.5 % on the
    // 1080p box-room frame over refilling at 32 (40 / 48 / 56 / 60 / 64 idle lanes: 640 / 647 / 652 / 653 / 651 Msamples/s).
    ExtendTuning tune_coherent = {60, 16, 4};
    // MIPT_TUNE_BATCH_COHERENT: mipt_trace_closest* batches whose rays turn out to arrive in a coherent order (k_make_rays
    // samples one wave in 256: common origin, directions within a narrow cone) -- chosen by the kernel itself from the
    // probe's two counters, no host round trip.  1.31 M triangles, 2^24 camera rays: 16 / 32 / 48 / 60 idle lanes ->
    // 3358 / 3471 / 3640 / 3547 Mrays/s; the same settings cost an incoherent batch 3 ... 19 %.
    ExtendTuning tune_batch_coherent = {48, 16, 4};
    int n_heads = 8;               // MIPT_HEADS=1|8: ray cursors per launch of the persistent kernel
    bool extend_simple = false;    // MIPT_EXTEND=simple: grid-stride reference kernel (A/B timing)
    int shadow_sort_from = 1;      // MIPT_SHADOW_SORT=<vertex>: first camera vertex whose shadow rays are binned by direction (-1: never)
    // closest-hit work buffers
    DevBuf ray_o, ray_d, hits, io_a, io_b, io_c, io_d, counters;
    unsigned char *noalloc_scratch = nullptr;  // device address of api.hip's fixed scratch array on this device
    DevBuf accum;  // mipt_render: this slot's partial framebuffer (W*H*3 fp32), reduced to slot 0 on the devices
    // wavefront work buffers (see wavefront.hip)
    DevBuf wf[36];
    EventPool events;
};

#ifdef MIPT_PHASE_CLOCKS
// developer build: shader clocks the waves of the extend kernel spent per section of the persistent loop (kernels_extend.h)
inline void dump_phase_stats(DeviceCtx &dc, const char *what) {
    static const char *names[6] = {"finish (store result)", "refill (fetch rays)", "open mesh (root slab)", "leaf scan (pooled)",
                                   "pop (rebuild slab)", "descend (<= step_cap steps)"};
    unsigned long long v[6];
    (void)hipMemcpy(v, dc.counters.as<char>() + 128, sizeof(v), hipMemcpyDeviceToHost);
    (void)hipMemset(dc.counters.as<char>() + 128, 0, sizeof(v));
    double total = 0;
    for (int i = 0; i < 6; i++) total += (double)v[i];
    fprintf(stderr, "[phase clocks] %s\n", what);
    for (int i = 0; i < 6; i++)
        fprintf(stderr, "  %-28s wave-clocks %16llu  share %.3f\n", names[i], v[i], total > 0 ? (double)v[i] / total : 0.0);
}
#elif defined(MIPT_PHASE_STATS)
// developer build: print and clear the phase table the instrumented extend kernel accumulated (see traverse.h)
inline void dump_phase_stats(DeviceCtx &dc, const char *what) {
    static const char *names[12] = {"loop", "finish", "refill", "open_mesh", "scan_entry", "scan_group4", "edge_tests", "pop_iter",
                                    "descend_entry", "descend_step", "leaf_cull", "-"};
    unsigned long long v[24];
    (void)hipMemcpy(v, dc.counters.as<char>() + 128, sizeof(v), hipMemcpyDeviceToHost);
    (void)hipMemset(dc.counters.as<char>() + 128, 0, sizeof(v));
    fprintf(stderr, "[phase stats] %s\n", what);
    for (int i = 0; i < 11; i++)
        fprintf(stderr, "  %-14s exec %12llu lanes %14llu  util %.3f\n", names[i], v[2 * i], v[2 * i + 1],
                v[2 * i] ? (double)v[2 * i + 1] / (64.0 * (double)v[2 * i]) : 0.0);
}
#endif

}  // namespace mipt
