// api.hip -- the C-ABI of libmipt (include/mipt.h) over the HIP runtime.
#include "../../include/mipt.h"

#include <hip/hip_runtime.h>

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
#include "scene_host.h"
#include "wavefront.h"

using namespace mipt;

struct mipt_scene {
    HostScene host;
    std::vector<DeviceCtx> devs;
    bool committed = false;
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

void release_device(DeviceCtx &dc) {
    if (dc.device < 0) return;
    (void)hipSetDevice(dc.device);
    if (dc.stream) (void)hipStreamSynchronize(dc.stream);
    dc.ray_o.release(); dc.ray_d.release(); dc.hits.release();
    dc.io_a.release(); dc.io_b.release(); dc.io_c.release(); dc.io_d.release(); dc.counters.release();
    for (DevBuf &b : dc.wf) b.release();
    dc.events.destroy();
    if (dc.arena) (void)hipFree(dc.arena);
    if (dc.stream) (void)hipStreamDestroy(dc.stream);
    dc = DeviceCtx();
}

bool check_material(const mipt_scene *s, int m) { return m >= 0 && m < (int)s->host.materials.size(); }

void fill_counters(DeviceCtx &dc, mipt_stats *st) {
    unsigned long long c[8] = {0};
    (void)hipMemcpy(c, dc.counters.p, sizeof(c), hipMemcpyDeviceToHost);
    st->ctr_nodes = (int64_t)c[0]; st->ctr_pops = (int64_t)c[1]; st->ctr_refs = (int64_t)c[2];
    st->ctr_tris = (int64_t)c[3]; st->ctr_hits = (int64_t)c[4]; st->ctr_culled = (int64_t)c[5];
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
    for (DeviceCtx &dc : s->devs) release_device(dc);
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
    HostMaterial m;
    for (int k = 0; k < 3; k++) { m.kd[k] = kd[k]; m.ks[k] = ks[k]; }
    s->host.materials.push_back(m);
    s->committed = false;
    return (int)s->host.materials.size() - 1;
}

int mipt_scene_add_light(mipt_scene *s, const float loc[3], const float power[3]) {
    if (!s) return fail(-1, "null scene");
    HostLight l;
    for (int k = 0; k < 3; k++) { l.loc[k] = loc[k]; l.power[k] = power[k]; }
    s->host.lights.push_back(l);
    s->committed = false;
    return (int)s->host.lights.size() - 1;
}

int mipt_scene_add_plane(mipt_scene *s, const float point[3], const float normal[3], int material) {
    if (!s) return fail(-1, "null scene");
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostPlane p;
    const f3 n = norm3(mk3(normal[0], normal[1], normal[2]));  // Plane ctor normalises (plane.cuh:16)
    p.normal[0] = n.x; p.normal[1] = n.y; p.normal[2] = n.z;
    for (int k = 0; k < 3; k++) p.point[k] = point[k];
    p.material = material;
    s->host.planes.push_back(p);
    s->committed = false;
    return (int)s->host.planes.size() - 1;
}

int mipt_scene_add_sphere(mipt_scene *s, const float center[3], float radius, int material) {
    if (!s) return fail(-1, "null scene");
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostSphere sp;
    for (int k = 0; k < 3; k++) sp.center[k] = center[k];
    sp.radius = radius;
    sp.material = material;
    s->host.spheres.push_back(sp);
    s->committed = false;
    return (int)s->host.spheres.size() - 1;
}

int mipt_scene_add_mesh(mipt_scene *s, const float *pos, const float *nrm, int64_t n_tris, int material) {
    if (!s || !pos || !nrm) return fail(-1, "null argument");
    if (n_tris <= 0) return fail(-1, "a mesh needs at least one triangle");
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostMesh m;
    m.n_tris = n_tris;
    m.material = material;
    m.pos.assign(pos, pos + n_tris * 9);
    m.nrm.assign(nrm, nrm + n_tris * 9);
    s->host.meshes.push_back(std::move(m));
    s->committed = false;
    return (int)s->host.meshes.size() - 1;
}

int mipt_scene_add_mesh_obj(mipt_scene *s, const char *path, const float position[3], const float rotate[3],
                            const float scale[3], int material) {
    if (!s || !path) return fail(-1, "null argument");
    if (!check_material(s, material)) return fail(-1, "unknown material id");
    HostMesh m;
    std::string err;
    if (!load_obj(path, position, rotate, scale, m, err)) return fail(-3, err);
    m.material = material;
    s->host.meshes.push_back(std::move(m));
    s->committed = false;
    return (int)s->host.meshes.size() - 1;
}

int mipt_scene_set_envmap_rgbe_dir(mipt_scene *s, const char *folder) {
    if (!s || !folder) return fail(-1, "null argument");
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
    return 0;
}

int mipt_scene_set_envmap_f32(mipt_scene *s, const float *faces6, int w, int h) {
    if (!s || !faces6 || w <= 0 || h <= 0) return fail(-1, "bad argument");
    HostEnv env;
    env.w = w;
    env.h = h;
    env.f32.assign(faces6, faces6 + (size_t)6 * w * h * 3);
    s->host.env = std::move(env);
    s->committed = false;
    return 0;
}

int mipt_scene_commit(mipt_scene *s, const int *device_ids, int n_devices) {
    if (!s) return fail(-1, "null scene");
    if (n_devices < 0 || (n_devices > 0 && !device_ids)) return fail(-1, "bad device list");
    for (DeviceCtx &dc : s->devs) release_device(dc);
    s->devs.clear();
    s->committed = false;
    std::string err;
    if (!s->host.pack(0, err)) return fail(-4, err);
    for (int i = 0; i < n_devices; i++) {
        DeviceCtx dc;
        dc.device = device_ids[i];
        HIP_TRY(hipSetDevice(dc.device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dc.device));
        dc.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_TRY(hipStreamCreateWithFlags(&dc.stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&dc.arena, s->host.arena.size()));
        HIP_TRY(hipMemcpyAsync(dc.arena, s->host.arena.data(), s->host.arena.size(), hipMemcpyHostToDevice, dc.stream));
        HIP_TRY(hipStreamSynchronize(dc.stream));
        HIP_TRY(dc.counters.reserve(512));
        HIP_TRY(hipMemset(dc.counters.p, 0, 512));  // [0,64) traversal counters, [64,128) work cursor of the persistent kernel
        dc.extend_blocks_per_cu = query_extend_blocks_per_cu();
        const char *ev = getenv("MIPT_EXTEND");
        dc.extend_simple = ev && std::string(ev) == "simple";
        if (const char *sv = getenv("MIPT_SHADOW_SORT")) dc.shadow_sort_from = atoi(sv);
        if (const char *tv = getenv("MIPT_TUNE")) sscanf(tv, "%d,%d,%d", &dc.tune.refill_lanes, &dc.tune.leaf_batch_lanes, &dc.tune.step_cap);
        if (const char *tv = getenv("MIPT_TUNE_COHERENT"))
            sscanf(tv, "%d,%d,%d", &dc.tune_coherent.refill_lanes, &dc.tune_coherent.leaf_batch_lanes, &dc.tune_coherent.step_cap);
        dc.view = s->host.view(dc.arena);
        s->devs.push_back(dc);
    }
    s->committed = true;
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
    HIP_TRY(hipMemsetAsync(dc.counters.p, 0, 512, dc.stream));
    const int blocks = (int)((n + 255) / 256);
    HIP_TRY(hipEventRecord(e0, dc.stream));
    hipLaunchKernelGGL(k_make_rays, dim3(blocks), dim3(256), 0, dc.stream, dc.view, d_ray_o, d_ray_d, n, dc.ray_o.as<f4>(),
                       dc.ray_d.as<f4>(), dc.hits.as<HitRec>());
    HIP_TRY(hipEventRecord(e1, dc.stream));
    launch_extend(dc, count, dc.ray_o.as<f4>(), dc.ray_d.as<f4>(), nullptr, nullptr, (int)n, n, dc.hits.as<HitRec>(),
                  reinterpret_cast<int *>(dc.counters.as<char>() + 64), dc.counters.as<unsigned long long>());
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
    }
#ifdef MIPT_PHASE_STATS
    if (!count) dump_phase_stats(dc, "trace_closest");
#endif
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
    std::vector<std::vector<float>> partial((size_t)G);
    std::vector<mipt_stats> stats((size_t)G);
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    auto work = [&](int g) {
        DeviceCtx &dc = s->devs[(size_t)g];
        mipt_render_config c = *cfg;
        const int stride = c.sample_stride > 0 ? c.sample_stride : 1;
        c.sample_begin = cfg->sample_begin + g * stride;  // samples s = begin + (g + k*G)*stride
        c.sample_stride = stride * G;
        if (hipSetDevice(dc.device) != hipSuccess) { rcs[(size_t)g] = -2; errs[(size_t)g] = "hipSetDevice failed"; return; }
        float *d_acc = nullptr;
        if (hipMalloc((void **)&d_acc, npix * 12) != hipSuccess) { rcs[(size_t)g] = -2; errs[(size_t)g] = "hipMalloc(accum) failed"; return; }
        (void)hipMemsetAsync(d_acc, 0, npix * 12, dc.stream);
        rcs[(size_t)g] = wavefront_render(s->host, dc, c, d_acc, &stats[(size_t)g], errs[(size_t)g]);
        if (rcs[(size_t)g] == 0) {
            partial[(size_t)g].resize(npix * 3);
            if (hipMemcpy(partial[(size_t)g].data(), d_acc, npix * 12, hipMemcpyDeviceToHost) != hipSuccess) {
                rcs[(size_t)g] = -2;
                errs[(size_t)g] = "hipMemcpy(accum) failed";
            }
        }
        (void)hipFree(d_acc);
    };
    if (G == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(work, g);
        for (std::thread &t : th) t.join();
    }
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != 0) return fail(rcs[(size_t)g], errs[(size_t)g]);
    memcpy(out, partial[0].data(), npix * 12);
    for (int g = 1; g < G; g++)
        for (size_t i = 0; i < npix * 3; i++) out[i] += partial[(size_t)g][i];
    if (st) {
        *st = stats[0];
        for (int g = 1; g < G; g++) {
            st->rays_total += stats[(size_t)g].rays_total; st->rays_camera += stats[(size_t)g].rays_camera;
            st->rays_shadow += stats[(size_t)g].rays_shadow; st->rays_light += stats[(size_t)g].rays_light;
            st->samples += stats[(size_t)g].samples; st->extend_launches += stats[(size_t)g].extend_launches;
            st->rays_extended += stats[(size_t)g].rays_extended;
            st->ms_total = std::max(st->ms_total, stats[(size_t)g].ms_total);
            st->ms_extend = std::max(st->ms_extend, stats[(size_t)g].ms_extend);
            st->ms_other = std::max(st->ms_other, stats[(size_t)g].ms_other);
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
    if (pos9) memcpy(pos9, m.pos.data(), m.pos.size() * 4);
    if (nrm9) memcpy(nrm9, m.nrm.data(), m.nrm.size() * 4);
    return 0;
}

int64_t mipt_scene_device_bytes(mipt_scene *s) { return s && s->host.packed ? (int64_t)s->host.arena.size() : 0; }
double mipt_scene_build_seconds(mipt_scene *s) { return s ? s->host.build_seconds : 0.0; }

}  // extern "C"
