// api.hip -- the C-ABI of libmipt (include/mipt.h) over the HIP runtime.
#include "../../include/mipt.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is opened at run time, when several devices render one frame
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "device_ctx.h"
#include "kernels_extend.h"
#include "pack_gpu.h"
#include "queue_sort.h"
#include "scene_host.h"
#include "wavefront.h"

using namespace mipt;

struct mipt_scene {
    HostScene host;
    std::vector<DeviceCtx> devs;
    bool committed = false;
    std::vector<ncclComm_t> comms;  // one per device slot, made by the first mipt_render over several devices
};

namespace {
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    } while (0)

// RCCL, bound at run time (a process that already holds an RCCL -- PyTorch's -- gets that one; a single-device user never
// loads it).  Replaces nothing in the reference (single GPU); it is what north_star's "RCCL all-reduce over xGMI" asks for
// when ONE process drives several devices: each device reduces its partial framebuffer to slot 0 over xGMI.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(lib, "ncclCommInitAll"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(lib, "ncclReduce"));
        group_start = reinterpret_cast<decltype(group_start)>(dlsym(lib, "ncclGroupStart"));
        group_end = reinterpret_cast<decltype(group_end)>(dlsym(lib, "ncclGroupEnd"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(lib, "ncclGetErrorString"));
        if (!comm_init_all || !comm_destroy || !reduce || !group_start || !group_end || !error_string) {
            err = "librccl lacks an expected symbol";
            lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

void release_comms(mipt_scene *s) {
    for (ncclComm_t c : s->comms)
        if (c && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c);
    s->comms.clear();
}

// dst += src over n floats (partial framebuffers of two slots that live on the SAME device: no transport needed)
__global__ __launch_bounds__(256) void k_add_into(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

// ---- scratch of mipt_trace_closest_device_noalloc ------------------------------------------------------------------
// A fixed device array instead of hipMalloc'ed work buffers: [0, 4 KiB) ray cursors + counters, then per ray the padded
// origin (16 B), direction (16 B) and hit record (8 B).  Every GPU that loads the library holds its own copy (~40 MiB)
// until the process ends; the array's NAME is a host-side stand-in -- host code uses the per-device address that
// hipGetSymbolAddress returned at commit (DeviceCtx::noalloc_scratch), never the name.
constexpr int64_t kNoAllocMaxRays = (int64_t)1 << 20;
constexpr size_t kNoAllocHead = 4096;
constexpr size_t kNoAllocBytes = kNoAllocHead + (size_t)kNoAllocMaxRays * (sizeof(f4) + sizeof(f4) + sizeof(HitRec));
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

void release_device(DeviceCtx &dc) {
    if (dc.device < 0) return;
    (void)hipSetDevice(dc.device);
    if (dc.stream) (void)hipStreamSynchronize(dc.stream);
    dc.ray_o.release(); dc.ray_d.release(); dc.hits.release();
    dc.io_a.release(); dc.io_b.release(); dc.io_c.release(); dc.io_d.release(); dc.counters.release(); dc.accum.release();
    for (DevBuf &b : dc.wf) b.release();
    dc.events.destroy();
    if (dc.ev_entry) (void)hipEventDestroy(dc.ev_entry);
    if (dc.arena) (void)hipFree(dc.arena);
    if (dc.stream) (void)hipStreamDestroy(dc.stream);
    dc = DeviceCtx();
}

// The scheduling knobs come from the environment (developer sweeps): a step cap of 0 would leave lanes on interior nodes
// that never advance, and the persistent kernel would never drain -- clamp everything into its working range.
void clamp_tuning(ExtendTuning &t) {
    t.refill_lanes = std::min(64, std::max(1, t.refill_lanes));
    t.leaf_batch_lanes = std::min(64, std::max(1, t.leaf_batch_lanes));
    t.step_cap = std::min(64, std::max(1, t.step_cap));
    t.pool1_below = std::min(65, std::max(0, t.pool1_below));
}

// A scene packed on the GPU keeps no host copy of its arena (pack_gpu.hip).  Whatever needs one on the host -- export,
// the introspection calls, the upload to a further device slot, a re-commit -- fetches it here, once.
int ensure_host_arena(mipt_scene *s) {
    HostScene &h = s->host;
    if (!h.packed || !h.arena.empty()) return 0;
    const void *src = h.dev_arena;
    int dev = h.dev_arena_device;
    if (!src && !s->devs.empty()) { src = s->devs[0].arena; dev = s->devs[0].device; }
    if (!src) return fail(-2, "the packed arena exists neither on the host nor on a device");
    h.arena.resize(h.layout.total);
    std::string err;
    if (!gpu_arena_read(dev, src, h.layout.total, h.arena.data(), err)) { h.arena.clear(); return fail(-2, err); }
    return 0;
}
void drop_dev_arena(HostScene &h) {
    if (h.dev_arena) gpu_free(h.dev_arena_device, h.dev_arena);
    h.dev_arena = nullptr;
    h.dev_arena_device = -1;
}

int setup_device(mipt_scene *s, DeviceCtx &dc) {
    HIP_TRY(hipSetDevice(dc.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dc.device));
    dc.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&dc.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&dc.ev_entry, hipEventDisableTiming));
    if (s->host.dev_arena && s->host.dev_arena_device == dc.device) {  // packed on this device: the arena is there already
        dc.arena = s->host.dev_arena;
        s->host.dev_arena = nullptr;
        s->host.dev_arena_device = -1;
    } else {
        if (ensure_host_arena(s) != 0) return -2;
        HIP_TRY(hipSetDevice(dc.device));
        HIP_TRY(hipMalloc(&dc.arena, s->host.arena.size()));
        HIP_TRY(hipMemcpyAsync(dc.arena, s->host.arena.data(), s->host.arena.size(), hipMemcpyHostToDevice, dc.stream));
        HIP_TRY(hipStreamSynchronize(dc.stream));
    }
    HIP_TRY(dc.counters.reserve(kCounterBytes));
    HIP_TRY(hipMemset(dc.counters.p, 0, kCounterBytes));
    HIP_TRY(hipGetSymbolAddress((void **)&dc.noalloc_scratch, HIP_SYMBOL(scratch)));  // this device's copy of the array
    dc.extend_blocks_per_cu = query_extend_blocks_per_cu();
    const char *ev = getenv("MIPT_EXTEND");
    dc.extend_simple = ev && std::string(ev) == "simple";
    if (const char *sv = getenv("MIPT_SHADOW_SORT")) dc.shadow_sort_from = atoi(sv);
    if (const char *hv = getenv("MIPT_HEADS")) dc.n_heads = atoi(hv) == 1 ? 1 : 8;
    if (const char *tv = getenv("MIPT_TUNE")) sscanf(tv, "%d,%d,%d,%d", &dc.tune.refill_lanes, &dc.tune.leaf_batch_lanes, &dc.tune.step_cap, &dc.tune.pool1_below);
    if (const char *tv = getenv("MIPT_TUNE_COHERENT"))
        sscanf(tv, "%d,%d,%d,%d", &dc.tune_coherent.refill_lanes, &dc.tune_coherent.leaf_batch_lanes, &dc.tune_coherent.step_cap,
               &dc.tune_coherent.pool1_below);
    clamp_tuning(dc.tune);
    if (const char *tv = getenv("MIPT_TUNE_BATCH_COHERENT"))
        sscanf(tv, "%d,%d,%d,%d", &dc.tune_batch_coherent.refill_lanes, &dc.tune_batch_coherent.leaf_batch_lanes,
               &dc.tune_batch_coherent.step_cap, &dc.tune_batch_coherent.pool1_below);
    clamp_tuning(dc.tune_coherent);
    clamp_tuning(dc.tune_batch_coherent);
    dc.view = s->host.view(dc.arena);
    return 0;
}

bool read_only(const mipt_scene *s) { return s->host.imported; }
const char *kReadOnly = "an imported scene cannot be edited (only its camera): build it from its description instead";

bool check_material(const mipt_scene *s, int m) { return m >= 0 && m < (int)s->host.materials.size(); }

void fill_counters(DeviceCtx &dc, mipt_stats *st) {
    unsigned long long c[8] = {0};
    (void)hipMemcpy(c, dc.counters.p, sizeof(c), hipMemcpyDeviceToHost);
    st->ctr_nodes = (int64_t)c[0]; st->ctr_pops = (int64_t)c[1]; st->ctr_refs = (int64_t)c[2];
    st->ctr_tris = (int64_t)c[3]; st->ctr_hits = (int64_t)c[4]; st->ctr_culled = (int64_t)c[5];
}

// Order the slot's private stream after everything queued so far on the caller's stream (see mipt_scene_set_caller_stream).
hipError_t wait_for_caller(DeviceCtx &dc) {
    hipError_t e = hipEventRecord(dc.ev_entry, dc.caller_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(dc.stream, dc.ev_entry, 0);
    return e;
}
}  // namespace

namespace mipt {
void set_error(const std::string &m) { g_err = m; }
}

extern "C" {

const char *mipt_last_error(void) { return g_err.c_str(); }
const char *mipt_version(void) { return "mipt 0.1 (gfx950)"; }

int mipt_scene_create(mipt_scene **out) {
    if (!out) return fail(-1, "null out pointer");
    *out = new mipt_scene();
    return 0;
}

void mipt_scene_destroy(mipt_scene *s) {
    if (!s) return;
    release_comms(s);
    for (DeviceCtx &dc : s->devs) release_device(dc);
    drop_dev_arena(s->host);
    delete s;
}

int mipt_scene_set_camera(mipt_scene *s, const float eye[3], const float lookat[3], const float vup[3], float fov) {
    if (!s) return fail(-1, "null scene");
    camera_set(s->host, eye, lookat, vup, fov);
    // the camera is part of the kernel argument view, not of the arena: refresh committed views
    for (DeviceCtx &dc : s->devs) dc.view = s->host.view(dc.arena);
    return 0;
}

int mipt_scene_add_material(mipt_scene *s, const float kd[3], const float ks[3]) {
    if (!s) return fail(-1, "null scene");
    if (read_only(s)) return fail(-1, kReadOnly);
    HostMaterial m;
    for (int k = 0; k < 3; k++) { m.kd[k] = kd[k]; m.ks[k] = ks[k]; }
    s->host.materials.push_back(m);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.materials.size() - 1;
}

int mipt_scene_add_light(mipt_scene *s, const float loc[3], const float power[3]) {
    if (!s) return fail(-1, "null scene");
    if (read_only(s)) return fail(-1, kReadOnly);
    HostLight l;
    for (int k = 0; k < 3; k++) { l.loc[k] = loc[k]; l.power[k] = power[k]; }
    s->host.lights.push_back(l);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.lights.size() - 1;
}

int mipt_scene_add_plane(mipt_scene *s, const float point[3], const float normal[3], int material) {
    if (!s) return fail(-1, "null scene");
    if (read_only(s)) return fail(-1, kReadOnly);
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostPlane p;
    const f3 n = norm3(mk3(normal[0], normal[1], normal[2]));  // Plane ctor normalises (plane.cuh:16)
    p.normal[0] = n.x; p.normal[1] = n.y; p.normal[2] = n.z;
    for (int k = 0; k < 3; k++) p.point[k] = point[k];
    p.material = material;
    s->host.planes.push_back(p);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.planes.size() - 1;
}

int mipt_scene_add_sphere(mipt_scene *s, const float center[3], float radius, int material) {
    if (!s) return fail(-1, "null scene");
    if (read_only(s)) return fail(-1, kReadOnly);
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostSphere sp;
    for (int k = 0; k < 3; k++) sp.center[k] = center[k];
    sp.radius = radius;
    sp.material = material;
    s->host.spheres.push_back(sp);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.spheres.size() - 1;
}

int mipt_scene_add_mesh(mipt_scene *s, const float *pos, const float *nrm, int64_t n_tris, int material) {
    if (!s || !pos || !nrm) return fail(-1, "null argument");
    if (read_only(s)) return fail(-1, kReadOnly);
    if (n_tris <= 0) return fail(-1, "a mesh needs at least one triangle");
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostMesh m;
    m.n_tris = n_tris;
    m.material = material;
    m.pos.assign(pos, pos + n_tris * 9);
    m.nrm.assign(nrm, nrm + n_tris * 9);
    s->host.meshes.push_back(std::move(m));
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.meshes.size() - 1;
}

int mipt_scene_add_mesh_obj(mipt_scene *s, const char *path, const float position[3], const float rotate[3],
                            const float scale[3], int material) {
    if (!s || !path) return fail(-1, "null argument");
    if (read_only(s)) return fail(-1, kReadOnly);
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostMesh m;
    std::string err;
    if (!load_obj(path, position, rotate, scale, m, err)) return fail(-3, err);
    m.material = material;
    s->host.meshes.push_back(std::move(m));
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return (int)s->host.meshes.size() - 1;
}

int mipt_scene_set_envmap_rgbe_dir(mipt_scene *s, const char *folder) {
    if (!s || !folder) return fail(-1, "null argument");
    if (read_only(s)) return fail(-1, kReadOnly);
    static const char *names[6] = {"px.hdr", "py.hdr", "pz.hdr", "nx.hdr", "ny.hdr", "nz.hdr"};  // envMap.cuh:35-40
    HostEnv env;
    for (int f = 0; f < 6; f++) {
        int w = 0, h = 0;
        std::vector<uint32_t> tex;
        std::string err;
        if (!load_hdr_rgbe((std::string(folder) + names[f]).c_str(), w, h, tex, err)) return fail(-3, err);
        if (f == 0) { env.w = w; env.h = h; }
        else if (w != env.w || h != env.h) return fail(-3, "cube faces differ in size");
        env.rgbe.insert(env.rgbe.end(), tex.begin(), tex.end());
    }
    s->host.env = std::move(env);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return 0;
}

int mipt_scene_set_envmap_f32(mipt_scene *s, const float *faces6, int w, int h) {
    if (!s || !faces6 || w <= 0 || h <= 0) return fail(-1, "bad argument");
    if (read_only(s)) return fail(-1, kReadOnly);
    HostEnv env;
    env.w = w;
    env.h = h;
    env.f32.assign(faces6, faces6 + (size_t)6 * w * h * 3);
    s->host.env = std::move(env);
    s->committed = false;
    s->host.packed = false;  // the description changed: the next commit packs again
    return 0;
}

int mipt_scene_commit(mipt_scene *s, const int *device_ids, int n_devices) {
    if (!s) return fail(-1, "null scene");
    if (n_devices < 0 || (n_devices > 0 && !device_ids)) return fail(-1, "bad device list");
    if (s->host.packed && s->host.arena.empty()) {  // a GPU-packed scene is committed again: its only copy lives on a device
        const int rc = ensure_host_arena(s);
        if (rc != 0) return rc;
    }
    release_comms(s);
    for (DeviceCtx &dc : s->devs) release_device(dc);
    s->devs.clear();
    s->committed = false;
    std::string err;
    // (a scene that is already packed -- committed before with no edit since, or imported -- is only uploaded)
    {
        const char *kb = getenv("MIPT_KD_BUILD");  // "host": every kd-tree on the host threads; default: large meshes on the first device
        s->host.build_device = (n_devices > 0 && !(kb && strcmp(kb, "host") == 0)) ? device_ids[0] : -1;
    }
    if (!s->host.packed && !s->host.pack(0, err)) return fail(-4, err);
    for (int i = 0; i < n_devices; i++) {
        DeviceCtx dc;
        dc.device = device_ids[i];
        const int rc = setup_device(s, dc);
        if (rc != 0) {  // nothing of a half set-up slot is kept (stream, arena, counters)
            const std::string why = g_err;
            if (s->host.arena.empty()) {  // do not lose a GPU-packed arena with the slots: fetch it while one still holds it
                if (s->devs.empty() && dc.arena) { s->devs.push_back(dc); (void)ensure_host_arena(s); s->devs.pop_back(); }
                else (void)ensure_host_arena(s);
            }
            release_device(dc);
            for (DeviceCtx &d : s->devs) release_device(d);
            s->devs.clear();
            return fail(rc, why);
        }
        s->devs.push_back(dc);
    }
    drop_dev_arena(s->host);  // (no slot on the build device adopted it: cannot happen today, the build device is slot 0's)
    s->committed = true;
    return 0;
}

int mipt_scene_set_caller_stream(mipt_scene *s, int slot, void *hip_stream) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (slot < 0 || slot >= (int)s->devs.size()) return fail(-1, "scene is not committed on that device slot");
    s->devs[(size_t)slot].caller_stream = static_cast<hipStream_t>(hip_stream);
    return 0;
}

int mipt_trace_closest_device(mipt_scene *s, int slot, const float *d_ray_o, const float *d_ray_d, int64_t n,
                              int32_t *d_prim, float *d_t, mipt_stats *st, int flags) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (slot < 0 || slot >= (int)s->devs.size()) return fail(-1, "scene is not committed on that device slot");
    if (n < 0 || n > (int64_t)1 << 30) return fail(-1, "ray count out of range");
    if (st) memset(st, 0, sizeof(*st));
    if (n == 0) return 0;
    if (!d_ray_o || !d_ray_d || !d_prim || !d_t) return fail(-1, "null buffer");
    DeviceCtx &dc = s->devs[slot];
    HIP_TRY(hipSetDevice(dc.device));
    HIP_TRY(dc.ray_o.reserve((size_t)n * sizeof(f4)));
    HIP_TRY(dc.ray_d.reserve((size_t)n * sizeof(f4)));
    HIP_TRY(dc.hits.reserve((size_t)n * sizeof(HitRec)));
    dc.events.reset();
    hipEvent_t e0 = dc.events.next(), e1 = dc.events.next(), e2 = dc.events.next(), e3 = dc.events.next();
    const bool count = (flags & MIPT_FLAG_COUNTERS) != 0;
    const int mode = extend_mode(flags);
    HIP_TRY(wait_for_caller(dc));
    HIP_TRY(hipMemsetAsync(dc.counters.p, 0, kCounterBytes, dc.stream));
    const int blocks = (int)((n + 255) / 256);
    HIP_TRY(hipEventRecord(e0, dc.stream));
    int *probe = reinterpret_cast<int *>(dc.counters.as<char>() + kCtrProbe);
    hipLaunchKernelGGL(k_make_rays, dim3(blocks), dim3(256), 0, dc.stream, dc.view, d_ray_o, d_ray_d, n, dc.ray_o.as<f4>(),
                       dc.ray_d.as<f4>(), dc.hits.as<HitRec>(), probe);
    HIP_TRY(hipEventRecord(e1, dc.stream));
    launch_extend(dc, mode, dc.ray_o.as<f4>(), dc.ray_d.as<f4>(), nullptr, nullptr, (int)n, n, dc.hits.as<HitRec>(),
                  reinterpret_cast<int *>(dc.counters.as<char>() + kCtrCursor), dc.counters.as<unsigned long long>(), false, 0, probe);
    HIP_TRY(hipEventRecord(e2, dc.stream));
    hipLaunchKernelGGL(k_split_hits, dim3(blocks), dim3(256), 0, dc.stream, dc.hits.as<HitRec>(), n, d_prim, d_t);
    HIP_TRY(hipEventRecord(e3, dc.stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(dc.stream));
    if (st) {
        float ms_all = 0, ms_ext = 0;
        HIP_TRY(hipEventElapsedTime(&ms_all, e0, e3));
        HIP_TRY(hipEventElapsedTime(&ms_ext, e1, e2));
        st->rays_total = n;
        st->rays_extended = n;
        st->extend_launches = 1;
        st->ms_total = ms_all;
        st->ms_extend = ms_ext;
        st->ms_other = ms_all - ms_ext;
        if (count) fill_counters(dc, st);
        if (mode == kCountWork) fill_work_counters(dc, st);
    }
#if defined(MIPT_PHASE_STATS) || defined(MIPT_PHASE_CLOCKS)
    if (!count) dump_phase_stats(dc, "trace_closest");
#endif
    return 0;
}

int mipt_trace_closest_device_noalloc(mipt_scene *s, int slot, const float *d_ray_o, const float *d_ray_d, int64_t n,
                                      int32_t *d_prim, float *d_t) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (slot < 0 || slot >= (int)s->devs.size()) return fail(-1, "scene is not committed on that device slot");
    if (n < 0 || n > kNoAllocMaxRays)
        return fail(-1, "ray count beyond the fixed scratch of the no-alloc entry point (at most 1048576 rays per call)");
    if (n == 0) return 0;
    if (!d_ray_o || !d_ray_d || !d_prim || !d_t) return fail(-1, "null buffer");
    DeviceCtx &dc = s->devs[slot];
    if (!dc.noalloc_scratch) return fail(-2, "scratch array address unknown on this device");
    HIP_TRY(hipSetDevice(dc.device));
    unsigned char *base = dc.noalloc_scratch;  // (the device address; `scratch` itself must not be used here)
    f4 *ro = reinterpret_cast<f4 *>(base + kNoAllocHead);
    f4 *rd = ro + kNoAllocMaxRays;
    HitRec *hits = reinterpret_cast<HitRec *>(rd + kNoAllocMaxRays);
    HIP_TRY(wait_for_caller(dc));
    HIP_TRY(hipMemsetAsync(base, 0, kNoAllocHead, dc.stream));  // the cursors must start at zero; the rest is overwritten
    const int blocks = (int)((n + 255) / 256);
    int *probe = reinterpret_cast<int *>(base + 8 * 256 + kCtrProbe);  // (inside the zeroed head of the scratch)
    hipLaunchKernelGGL(k_make_rays, dim3(blocks), dim3(256), 0, dc.stream, dc.view, d_ray_o, d_ray_d, n, ro, rd, hits, probe);
    launch_extend(dc, kProd, ro, rd, nullptr, nullptr, (int)n, n, hits, reinterpret_cast<int *>(base),
                  reinterpret_cast<unsigned long long *>(base + 8 * 256), false, 0, probe);
    hipLaunchKernelGGL(k_split_hits, dim3(blocks), dim3(256), 0, dc.stream, hits, n, d_prim, d_t);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(dc.stream));
    return 0;
}

int mipt_trace_closest(mipt_scene *s, const float *ro, const float *rd, int64_t n, int32_t *prim, float *t, mipt_stats *st,
                       int flags) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (s->devs.empty()) return fail(-1, "scene is not committed on any device (HIP device required)");
    if (st) memset(st, 0, sizeof(*st));
    if (n == 0) return 0;
    if (n < 0 || !ro || !rd || !prim || !t) return fail(-1, "bad argument");
    DeviceCtx &dc = s->devs[0];
    HIP_TRY(hipSetDevice(dc.device));
    HIP_TRY(dc.io_a.reserve((size_t)n * 12));
    HIP_TRY(dc.io_b.reserve((size_t)n * 12));
    HIP_TRY(dc.io_c.reserve((size_t)n * 4));
    HIP_TRY(dc.io_d.reserve((size_t)n * 4));
    HIP_TRY(hipMemcpy(dc.io_a.p, ro, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.io_b.p, rd, (size_t)n * 12, hipMemcpyHostToDevice));
    int rc = mipt_trace_closest_device(s, 0, dc.io_a.as<float>(), dc.io_b.as<float>(), n, dc.io_c.as<int32_t>(),
                                       dc.io_d.as<float>(), st, flags);
    if (rc != 0) return rc;
    HIP_TRY(hipMemcpy(prim, dc.io_c.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, dc.io_d.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mipt_render_device(mipt_scene *s, int slot, const mipt_render_config *cfg, float *d_accum, mipt_stats *st) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (slot < 0 || slot >= (int)s->devs.size()) return fail(-1, "scene is not committed on that device slot");
    if (!cfg || !d_accum) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(s->devs[slot].device));
    HIP_TRY(wait_for_caller(s->devs[slot]));
    std::string err;
    int rc = wavefront_render(s->host, s->devs[slot], *cfg, d_accum, st, err);
    if (rc != 0) return fail(rc, err);
    return 0;
}

int mipt_render(mipt_scene *s, const mipt_render_config *cfg, float *out, mipt_stats *st) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (s->devs.empty()) return fail(-1, "scene is not committed on any device (HIP device required)");
    if (!cfg || !out) return fail(-1, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(-1, "bad image size");
    const size_t npix = (size_t)cfg->width * cfg->height;
    const int G = (int)s->devs.size();
    std::vector<mipt_stats> stats((size_t)G);
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    // every slot renders its samples (s mod G) into its own device framebuffer, kept from call to call
    auto work = [&](int g) {
        DeviceCtx &dc = s->devs[(size_t)g];
        mipt_render_config c = *cfg;
        const int stride = c.sample_stride > 0 ? c.sample_stride : 1;
        c.sample_begin = cfg->sample_begin + g * stride;  // samples s = begin + (g + k*G)*stride
        c.sample_stride = stride * G;
        hipError_t e = hipSetDevice(dc.device);
        if (e == hipSuccess) e = dc.accum.reserve(npix * 12);
        if (e == hipSuccess) e = hipMemsetAsync(dc.accum.p, 0, npix * 12, dc.stream);
        if (e != hipSuccess) { rcs[(size_t)g] = -2; errs[(size_t)g] = std::string("framebuffer set-up: ") + hipGetErrorString(e); return; }
        rcs[(size_t)g] = wavefront_render(s->host, dc, c, dc.accum.as<float>(), &stats[(size_t)g], errs[(size_t)g]);
    };
    if (G == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(work, g);
        for (std::thread &t : th) t.join();
    }
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != 0) return fail(rcs[(size_t)g], errs[(size_t)g]);
    // combine on the devices: ONE RCCL reduce to slot 0 over xGMI (replaces G device-to-host copies and a host sum); slots
    // that share a physical device (a test set-up) are added by a kernel on that device instead -- RCCL wants distinct ones
    bool distinct = true;
    for (int a = 0; a < G; a++)
        for (int b = a + 1; b < G; b++) distinct &= s->devs[(size_t)a].device != s->devs[(size_t)b].device;
    const char *force = getenv("MIPT_FORCE_RCCL");
    if (distinct && (G > 1 || (force && atoi(force) != 0))) {
        std::string err;
        if (!g_rccl.load(err)) return fail(-6, err);
        if (s->comms.empty()) {
            std::vector<int> ids;
            for (const DeviceCtx &dc : s->devs) ids.push_back(dc.device);
            s->comms.assign((size_t)G, nullptr);
            const ncclResult_t r = g_rccl.comm_init_all(s->comms.data(), G, ids.data());
            if (r != ncclSuccess) { s->comms.clear(); return fail(-6, std::string("ncclCommInitAll: ") + g_rccl.error_string(r)); }
        }
        // (a group that was started is always ended, whatever fails inside it: an open group would swallow every later
        //  collective of the process)
        ncclResult_t r = g_rccl.group_start();
        if (r != ncclSuccess) return fail(-6, std::string("ncclGroupStart: ") + g_rccl.error_string(r));
        hipError_t he = hipSuccess;
        for (int g = 0; g < G && r == ncclSuccess && he == hipSuccess; g++) {
            DeviceCtx &dc = s->devs[(size_t)g];
            he = hipSetDevice(dc.device);
            if (he == hipSuccess)
                r = g_rccl.reduce(dc.accum.p, dc.accum.p, npix * 3, ncclFloat, ncclSum, 0, s->comms[(size_t)g], dc.stream);
        }
        const ncclResult_t r2 = g_rccl.group_end();
        if (he != hipSuccess) return fail(-2, std::string("hipSetDevice before ncclReduce: ") + hipGetErrorString(he));
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return fail(-6, std::string("ncclReduce: ") + g_rccl.error_string(r));
        for (int g = G - 1; g >= 0; g--) {
            HIP_TRY(hipSetDevice(s->devs[(size_t)g].device));
            HIP_TRY(hipStreamSynchronize(s->devs[(size_t)g].stream));
        }
    } else if (G > 1) {
        DeviceCtx &root = s->devs[0];
        HIP_TRY(hipSetDevice(root.device));
        for (int g = 1; g < G; g++) {
            if (s->devs[(size_t)g].device != root.device)
                return fail(-1, "device slots must be all distinct (RCCL reduce) or all on one device");
            hipLaunchKernelGGL(k_add_into, dim3((unsigned)((npix * 3 + 255) / 256)), dim3(256), 0, root.stream, root.accum.as<float>(),
                               s->devs[(size_t)g].accum.as<float>(), npix * 3);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(root.stream));
    }
    HIP_TRY(hipSetDevice(s->devs[0].device));
    HIP_TRY(hipMemcpy(out, s->devs[0].accum.p, npix * 12, hipMemcpyDeviceToHost));  // kernel.cu:416
    if (st) {
        *st = stats[0];
        for (int g = 1; g < G; g++) {
            const mipt_stats &o = stats[(size_t)g];
            st->rays_total += o.rays_total; st->rays_camera += o.rays_camera;
            st->rays_shadow += o.rays_shadow; st->rays_light += o.rays_light;
            st->samples += o.samples; st->extend_launches += o.extend_launches;
            st->rays_extended += o.rays_extended; st->nonfinite_samples += o.nonfinite_samples;
            st->ctr_nodes += o.ctr_nodes; st->ctr_pops += o.ctr_pops; st->ctr_refs += o.ctr_refs; st->ctr_tris += o.ctr_tris;
            st->ctr_hits += o.ctr_hits; st->ctr_culled += o.ctr_culled;
            st->work_steps += o.work_steps; st->work_pops += o.work_pops; st->work_sphere_pairs += o.work_sphere_pairs;
            st->work_planes += o.work_planes; st->work_verts += o.work_verts; st->work_leaves += o.work_leaves;
            st->ms_total = std::max(st->ms_total, o.ms_total);
            st->ms_extend = std::max(st->ms_extend, o.ms_extend);
            st->ms_other = std::max(st->ms_other, o.ms_other);
            st->shade_launches += o.shade_launches;
            st->ms_shade = std::max(st->ms_shade, o.ms_shade);
        }
    }
    return 0;
}

int mipt_env_lookup(mipt_scene *s, const float *dirs, int64_t n, float *rgb) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (s->devs.empty()) return fail(-1, "scene is not committed on any device (HIP device required)");
    if (!s->host.env.present()) return fail(-1, "scene has no environment map");
    if (n == 0) return 0;
    if (n < 0 || !dirs || !rgb) return fail(-1, "bad argument");
    DeviceCtx &dc = s->devs[0];
    HIP_TRY(hipSetDevice(dc.device));
    HIP_TRY(dc.io_a.reserve((size_t)n * 12));
    HIP_TRY(dc.io_b.reserve((size_t)n * 12));
    HIP_TRY(hipMemcpy(dc.io_a.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    std::string err;
    int rc = wavefront_env_lookup(dc, dc.io_a.as<float>(), n, dc.io_b.as<float>(), err);
    if (rc != 0) return fail(rc, err);
    HIP_TRY(hipMemcpy(rgb, dc.io_b.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return 0;
}

int mipt_queue_bin_device(mipt_scene *s, int slot, const uint32_t *d_keys, const int32_t *d_vals, const int32_t *d_count, int64_t capacity,
                          int key_bits, int32_t *d_out) {
    if (!s || !s->committed) return fail(-1, "scene not committed");
    if (slot < 0 || slot >= (int)s->devs.size()) return fail(-1, "scene is not committed on that device slot");
    if (!d_keys || !d_vals || !d_count || !d_out) return fail(-1, "null buffer");
    if (capacity <= 0 || capacity > ((int64_t)1 << 30)) return fail(-1, "queue capacity out of range");
    if (key_bits < kSortFineBits || key_bits > 32) return fail(-1, "key_bits must be 14..32");
    DeviceCtx &dc = s->devs[slot];
    HIP_TRY(hipSetDevice(dc.device));
    HIP_TRY(dc.io_a.reserve((size_t)capacity * 4));
    HIP_TRY(dc.io_b.reserve((size_t)capacity * 4));
    HIP_TRY(dc.io_c.reserve((size_t)kSortScratchInts * sizeof(int)));
    HIP_TRY(wait_for_caller(dc));
    launch_queue_sort(dc.stream, dc.num_cus, d_keys, d_vals, d_count, (unsigned)key_bits, dc.io_a.as<uint32_t>(), dc.io_b.as<int>(),
                      dc.io_c.as<int>(), d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(dc.stream));
    return 0;
}

int mipt_hdr_load_rgbe(const char *path, int32_t *w, int32_t *h, uint32_t *texels) {
    if (!path || !w || !h) return fail(-1, "bad argument");
    std::vector<uint32_t> tex;
    std::string err;
    int ww = 0, hh = 0;
    if (!load_hdr_rgbe(path, ww, hh, tex, err)) return fail(-3, err);
    *w = ww;
    *h = hh;
    if (texels) memcpy(texels, tex.data(), tex.size() * sizeof(uint32_t));
    return 0;
}

int mipt_scene_gpu_built_meshes(mipt_scene *s) { return s ? s->host.meshes_built_on_gpu : fail(-1, "null scene"); }

int mipt_mesh_count(mipt_scene *s) { return s ? (int)s->host.meshes.size() : fail(-1, "null scene"); }

int mipt_kd_sizes(mipt_scene *s, int mesh, int64_t *nn, int64_t *nr, int64_t *nt) {
    if (!s || !s->host.packed) return fail(-1, "scene not committed");
    if (mesh < 0 || mesh >= (int)s->host.meshes.size()) return fail(-1, "bad mesh index");
    const HostMesh &m = s->host.meshes[(size_t)mesh];
    if (nn) *nn = m.n_nodes;
    if (nr) *nr = m.n_refs;
    if (nt) *nt = m.n_tris;
    return 0;
}

int mipt_kd_export(mipt_scene *s, int mesh, float *boxes, int32_t *axis, int32_t *left, int32_t *right, int32_t *first,
                   int32_t *count, int32_t *refs) {
    if (!s || !s->host.packed) return fail(-1, "scene not committed");
    if (mesh < 0 || mesh >= (int)s->host.meshes.size()) return fail(-1, "bad mesh index");
    if (ensure_host_arena(s) != 0) return -2;
    const HostMesh &m = s->host.meshes[(size_t)mesh];
    const DScene v = s->host.view(s->host.arena.data());
    for (int64_t i = 0; i < m.n_nodes; i++) {
        const KdNode k = v.nodes[m.node_base + i];
        const bool leaf = (k.b >> 30) == 3u;
        if (boxes) {
            const f4 b0 = v.boxes[2 * (m.node_base + i)], b1 = v.boxes[2 * (m.node_base + i) + 1];
            const float b[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
            memcpy(boxes + 6 * i, b, 24);
        }
        if (axis) axis[i] = leaf ? -1 : (int32_t)(k.b >> 30);
        if (left) left[i] = leaf ? -1 : (int32_t)((k.b & kIdxMask) - (uint32_t)m.node_base);
        if (right) right[i] = leaf ? -1 : (int32_t)((k.b & kIdxMask) - (uint32_t)m.node_base + 1u);
        if (first) first[i] = leaf ? (int32_t)(k.a - (uint32_t)m.ref_base) : 0;
        if (count) count[i] = leaf ? (int32_t)(k.b & kIdxMask) : 0;
    }
    if (refs)
        for (int64_t i = 0; i < m.n_refs; i++) refs[i] = v.refs[m.ref_base + i] - (int32_t)m.tri_base;
    return 0;
}

int mipt_mesh_triangles(mipt_scene *s, int mesh, float *pos9, float *nrm9) {
    if (!s) return fail(-1, "null scene");
    if (mesh < 0 || mesh >= (int)s->host.meshes.size()) return fail(-1, "bad mesh index");
    const HostMesh &m = s->host.meshes[(size_t)mesh];
    if (!s->host.imported) {
        if (pos9) memcpy(pos9, m.pos.data(), m.pos.size() * 4);
        if (nrm9) memcpy(nrm9, m.nrm.data(), m.nrm.size() * 4);
        return 0;
    }
    // an imported scene keeps no triangle arrays: read them back from the packed float4 records
    if (ensure_host_arena(s) != 0) return -2;
    const DScene v = s->host.view(s->host.arena.data());
    for (int64_t i = 0; i < m.n_tris * 3; i++) {
        const f4 p = v.tri_pos[3 * m.tri_base + i], n = v.tri_nrm[3 * m.tri_base + i];
        if (pos9) { pos9[3 * i] = p.x; pos9[3 * i + 1] = p.y; pos9[3 * i + 2] = p.z; }
        if (nrm9) { nrm9[3 * i] = n.x; nrm9[3 * i + 1] = n.y; nrm9[3 * i + 2] = n.z; }
    }
    return 0;
}

int64_t mipt_scene_export_size(mipt_scene *s) {
    if (!s || !s->host.packed) return fail(-1, "scene not committed");
    if (ensure_host_arena(s) != 0) return -2;
    return (int64_t)export_blob(s->host, nullptr);
}

int mipt_scene_export(mipt_scene *s, void *blob, int64_t size) {
    if (!s || !s->host.packed) return fail(-1, "scene not committed");
    if (ensure_host_arena(s) != 0) return -2;
    if (!blob || size < (int64_t)export_blob(s->host, nullptr)) return fail(-1, "export buffer too small");
    export_blob(s->host, static_cast<uint8_t *>(blob));
    return 0;
}

int mipt_scene_import(mipt_scene *s, const void *blob, int64_t size) {
    if (!s || !blob || size <= 0) return fail(-1, "bad argument");
    release_comms(s);
    for (DeviceCtx &dc : s->devs) release_device(dc);
    s->devs.clear();
    s->committed = false;
    std::string err;
    HostScene fresh;
    if (!import_blob(fresh, static_cast<const uint8_t *>(blob), (size_t)size, err)) return fail(-5, err);
    drop_dev_arena(s->host);
    s->host = std::move(fresh);
    return 0;
}

int64_t mipt_scene_device_bytes(mipt_scene *s) { return s && s->host.packed ? (int64_t)s->host.layout.total : 0; }
int mipt_scene_packed_on_gpu(mipt_scene *s) { return s ? (s->host.packed_on_gpu ? 1 : 0) : fail(-1, "null scene"); }
double mipt_scene_build_seconds(mipt_scene *s) { return s ? s->host.build_seconds : 0.0; }
double mipt_scene_kd_seconds(mipt_scene *s) { return s ? s->host.kd_seconds : 0.0; }

}  // extern "C"
