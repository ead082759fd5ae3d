// pack_gpu.h -- the packer's device side (pack_gpu.hip), called from scene.cpp (plain C++: no HIP types here).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "scene_host.h"

namespace mipt {

struct GpuPackMesh {
    const float *pos, *nrm;  // host arrays, n_tris * 9 each
    int64_t n_tris;
    const KdTree *tree;      // flat tree on the host (either builder)
    int64_t tri_base, node_base, ref_base;
    const void *d_pos;       // the positions if they are on the device already (uploaded once for the kd build), or null
};

// Allocates the scene's device arena (L.total bytes) on `device` and fills its big sections [0, L.meshes) -- nodes, boxes, cull
// boxes, step records, leaf records, triangle arrays -- byte for byte as HostScene::pack does on the host.  root_tight receives
// the root cull box (8 floats) of every mesh.  On failure nothing is left allocated and the caller packs on the host.
bool gpu_pack_scene(int device, const std::vector<GpuPackMesh> &meshes, const ArenaLayout &L, int64_t total_refs, bool force_recheck,
                    void **d_arena_out, float *root_tight, std::string &err);
bool gpu_arena_write(int device, void *d_arena, size_t off, const void *src, size_t bytes, std::string &err);
bool gpu_arena_read(int device, const void *d_arena, size_t bytes, uint8_t *dst, std::string &err);
void *gpu_upload(int device, const void *src, size_t bytes, std::string &err);
void gpu_free(int device, void *p);

}  // namespace mipt
