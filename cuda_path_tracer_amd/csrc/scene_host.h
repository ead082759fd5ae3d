// scene_host.h -- host-side scene description and packed arena (internal to libmipt).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "kd_build.h"
#include "scene_types.h"

namespace mipt {

struct HostMaterial { float kd[3], ks[3]; };
struct HostLight { float loc[3], power[3]; };
struct HostPlane { float point[3], normal[3]; int material; };
struct HostSphere { float center[3]; float radius; int material; };
struct HostMesh {
    std::vector<float> pos, nrm;  // n_tris*9 each
    int material = 0;
    int64_t n_tris = 0;
    // filled by commit
    int64_t tri_base = 0, node_base = 0, ref_base = 0, n_nodes = 0, n_refs = 0;
    int64_t degenerate_nodes = 0, n_leaves = 0, max_leaf = 0;
    int max_depth = 0;
};
struct HostEnv {
    int w = 0, h = 0;
    std::vector<uint32_t> rgbe;  // 6*w*h when loaded from .hdr
    std::vector<float> f32;      // 6*w*h*3 when given as floats
    bool present() const { return w > 0; }
};
// The env section of the arena: the six faces, then the sampling tables of the map as a light (use_envmap == 2; scene.cpp
// build_env_light): marg[6 h + 1] (CDF over all rows; the extra entry is 1.0f when the map can be sampled, 0.0f when it is
// black), cond[6 h][w] (CDF inside a row).  Offsets relative to the section, each 256-byte aligned.
struct EnvSection {
    size_t marg = 0, cond = 0, bytes = 0;
};
inline EnvSection env_section(int w, int h, bool is_rgbe) {
    EnvSection e;
    if (w <= 0 || h <= 0) return e;
    const size_t faces = (size_t)6 * w * h * (is_rgbe ? 4 : 12);
    e.marg = (faces + 255) & ~(size_t)255;
    e.cond = (e.marg + ((size_t)6 * h + 1) * 4 + 255) & ~(size_t)255;
    e.bytes = e.cond + (size_t)6 * h * w * 4;
    return e;
}

// Byte offsets of each section inside the arena (all 256-byte aligned).
struct ArenaLayout {
    size_t nodes = 0, boxes = 0, tight = 0, tight_pairs = 0, refs = 0, leaf_recs = 0, leaf_spheres = 0, tri_pos = 0, tri_nrm = 0, meshes = 0, mesh_order = 0, spheres = 0, planes = 0, lights = 0,
           materials = 0, env = 0, total = 0;
};

struct HostScene {
    bool cam_set = false;
    float cam_eye[3], cam_lookat[3], cam_right[3], cam_up[3];
    std::vector<HostMaterial> materials;
    std::vector<HostLight> lights;
    std::vector<HostPlane> planes;
    std::vector<HostSphere> spheres;
    std::vector<HostMesh> meshes;
    HostEnv env;

    // kd-trees of meshes with at least kGpuBuildMinTris triangles are built on this HIP device (kd_build_gpu.hip); -1: all on
    // the host.  Set by mipt_scene_commit when it has a device and MIPT_KD_BUILD is not "host".  Both builders make the same tree.
    int build_device = -1;
    int meshes_built_on_gpu = 0;
    bool packed = false;
    bool imported = false;       // packed state came from mipt_scene_import: no triangle arrays on the host, nothing to rebuild
    // host copy of the device arena.  EMPTY after a commit that packed on the GPU (pack_gpu.hip): the arena then exists on the
    // device only (dev_arena until a device slot adopts it) and is fetched when something on the host asks for it
    // (mipt_scene_export, mipt_kd_export, a second device slot) -- api.hip: ensure_host_arena().
    std::vector<uint8_t> arena;
    void *dev_arena = nullptr;   // GPU-packed arena on device `dev_arena_device`, not yet adopted by a device slot
    int dev_arena_device = -1;
    bool packed_on_gpu = false;  // the last pack() made the big sections on the device
    ArenaLayout layout;
    int64_t total_tris = 0, total_nodes = 0, total_refs = 0;
    double build_seconds = 0.0;  // kd-trees + pack
    double kd_seconds = 0.0;     // kd-trees alone

    // Build kd-trees and pack everything into `arena`.  Returns false and sets err on failure.
    bool pack(int n_threads, std::string &err);
    // View whose pointers are `base` + section offsets (base = arena.data() for the host view).
    DScene view(const void *base) const;
};

// Flat serialised form of a packed scene (mipt_scene_export / _import): header + tables + the arena.  With out == nullptr
// only the size is computed.  import_blob leaves `s` packed and `imported` (no triangle arrays, commit only uploads).
size_t export_blob(const HostScene &s, uint8_t *out);
bool import_blob(HostScene &s, const uint8_t *blob, size_t size, std::string &err);

void camera_set(HostScene &s, const float eye[3], const float lookat[3], const float vup[3], float fov);
bool load_obj(const char *path, const float position[3], const float rotate[3], const float scale[3], HostMesh &out,
              std::string &err);
bool load_hdr_rgbe(const char *path, int &w, int &h, std::vector<uint32_t> &texels, std::string &err);

}  // namespace mipt
