/*
 * mipt.h -- C-ABI of the MI355X-native path-tracer core (libmipt.so).
 *
 * This is the drop-in boundary for the rendering path of fazakasreka/cuda-path-tracer.
 * The reference has no FFI; its boundary is the C++ scene-construction surface plus the
 * `render` kernel launch inside one translation unit.  Every entry point below names the
 * reference interface it replaces (paths relative to /root/reference/cuda-path-tracer/).
 *
 * Conventions: plain pointers and sizes only; 0 (or an id >= 0) = OK, negative = error and
 * mipt_last_error() (thread-local) describes it; host buffers are owned by the caller,
 * device memory by the scene; one host thread drives a scene at a time.  All arithmetic on
 * the path is fp32 in the reference's operation order (no FMA contraction).
 */
#ifndef MIPT_H
#define MIPT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mipt_scene mipt_scene;

/* Replaces the compile-time constants of constants/constants.cuh:6-9 and the launch geometry
 * of kernel.cu:397 with a runtime POD. */
typedef struct mipt_render_config {
    int32_t width, height;   /* screenWidth/screenHeight (constants.cuh:6)                        */
    int32_t spp;             /* nSamples (constants.cuh:9): normalisation constant, kernel.cu:343 */
    int32_t cam_depth;       /* camera sub-path vertices (maxdepth, constants.cuh:8)              */
    int32_t light_depth;     /* light sub-path vertices; 0 = "simple" path tracing (README.md:17) */
    int32_t use_envmap;      /* 1: add cube-map radiance on a camera-ray miss (img/environment_mapping.png); 2 (build-defined,
                                no counterpart in the reference): ... and sample the map as a LIGHT at every camera vertex
                                (texels drawn in proportion to radiance x solid angle, one shadow ray each; a ray that leaves
                                the scene then collects the map only at the first vertex and behind a mirror bounce) */
    uint32_t seed;           /* counter-based RNG key (replaces curand_init, kernel.cu:331)       */
    int32_t sample_begin, sample_end, sample_stride; /* this call renders s = begin, begin+stride, ... < end */
    int32_t samples_per_pass; /* samples of every pixel in flight per wavefront pass; 0 = auto    */
    int32_t flags;            /* MIPT_FLAG_COUNTERS: also fill the traversal counters (slower kernel) */
} mipt_render_config;

/* Replaces the wall-clock print of kernel.cu:395-412 with per-stage GPU timings and ray counts. */
typedef struct mipt_stats {
    int64_t rays_total;      /* closest-hit queries traced (all are closest-hit, kernel.cu:249,268) */
    int64_t rays_camera, rays_shadow, rays_light;
    int64_t samples;         /* pixel-samples rendered by this call                              */
    int64_t extend_launches; /* launches of the kd closest-hit kernel                            */
    int64_t rays_extended;   /* of rays_total: rays traced by that kernel (the rest miss every mesh's root box and are
                                answered by the stage that generated them)                           */
    double ms_total;         /* GPU time of the whole call (HIP events on the scene's stream)    */
    double ms_extend;        /* summed GPU time of the closest-hit kernel launches               */
    double ms_other;         /* raygen/shade/connect/resolve/accumulate                          */
    /* traversal counters, filled only when MIPT_FLAG_COUNTERS is requested (slower kernel):   */
    int64_t ctr_nodes, ctr_pops, ctr_refs, ctr_tris, ctr_hits;
    int64_t ctr_culled;      /* of ctr_tris: candidates skipped by the per-leaf cull box (never changes a result) */
    /* samples whose radiance came out non-finite (the reference's arithmetic can divide by a vanishing pdf or take the
     * sqrt of a tiny negative, samplers.cuh:19, kernel.cu:254,276; such a sample poisons image[id], kernel.cu:343) */
    int64_t nonfinite_samples;
    /* what the PRODUCTION closest-hit kernel read, filled only with MIPT_FLAG_WORK_COUNTERS (same results, slower): */
    int64_t work_steps;      /* node steps: one 64-byte step record each (both children's cull boxes and nodes)  */
    int64_t work_pops;       /* stack pops + mesh roots opened: one 8-byte node + one 32-byte box each         */
    int64_t work_sphere_pairs; /* 32-byte candidate-sphere pairs streamed by the leaf scans                    */
    int64_t work_planes;     /* marked candidates: one 16-byte plane record each                               */
    int64_t work_verts;      /* of those, candidates that reached the edge tests: 48 bytes of vertices each    */
    int64_t work_leaves;     /* leaf scans (one 4-byte reference -> triangle id lookup per scan with a hit)    */
    /* the second-largest kernel of a render, timed like the closest-hit launches (HIP events on the scene's stream): */
    int64_t shade_launches;  /* launches of the camera-vertex shading kernel (one per camera vertex and pass)   */
    double ms_shade;         /* their summed GPU time (part of ms_other)                                        */
} mipt_stats;

enum {
    MIPT_FLAG_COUNTERS = 1,        /* reference-walk kernel + its counters (ctr_*)                                */
    MIPT_FLAG_DROP_NONFINITE = 2,  /* opt-in deviation: a non-finite sample adds nothing to the image (still counted) */
    MIPT_FLAG_WORK_COUNTERS = 4    /* production kernel + what it read (work_*)                                  */
};

/* ---- scene construction: replaces Scene::build (kernel.cu:52-142) --------------------- */
int  mipt_scene_create(mipt_scene **out);                        /* `Scene scene;` kernel.cu:383 */
void mipt_scene_destroy(mipt_scene *);                           /* reference never frees (cudaDeviceReset, kernel.cu:425) */
int  mipt_scene_set_camera(mipt_scene *, const float eye[3], const float lookat[3], const float vup[3],
                           float fov_rad);                       /* Camera::set camera.cuh:9 */
int  mipt_scene_add_material(mipt_scene *, const float diffuse[3], const float mirror[3]); /* Material() material.cuh:9; returns id */
int  mipt_scene_add_light(mipt_scene *, const float location[3], const float power[3]);    /* Light() light.cuh:15 */
int  mipt_scene_add_plane(mipt_scene *, const float point[3], const float normal[3], int material);  /* Plane() plane.cuh:14 */
int  mipt_scene_add_sphere(mipt_scene *, const float center[3], float radius, int material);         /* Sphere() sphere.cuh:15 */
/* De-indexed world-space triangles: pos/nrm are n_tris*9 floats (a,b,c).  Replaces Mesh(std::vector<Triangle*>) mesh.cuh:240 */
int  mipt_scene_add_mesh(mipt_scene *, const float *pos, const float *nrm, int64_t n_tris, int material);
/* Replaces MeshObject(objPath, position, rotate, scale, Material*) mesh.cuh:322 -> readObjIntoMesh mesh.cu:232-317
 * (v / vn / f with p//n or p/t/n, SRT baked in, including the y/z swap of vec4(vec3,w) math.cuh:55). */
int  mipt_scene_add_mesh_obj(mipt_scene *, const char *path, const float position[3], const float rotate[3],
                             const float scale[3], int material);
/* Replaces EnvMap(folder) envMap.cuh:34-49 + HDRLoader::load HDRLoader.cu:29-97 (px,py,pz,nx,ny,nz.hdr).
 * Faces are kept as raw RGBE (4 B/texel) and decoded in-kernel, bit-exact w.r.t. convertComponent HDRLoader.cu:99-104. */
int  mipt_scene_set_envmap_rgbe_dir(mipt_scene *, const char *folder);
int  mipt_scene_set_envmap_f32(mipt_scene *, const float *faces6, int w, int h);  /* six w*h*3 fp32 faces, order px,py,pz,nx,ny,nz */
/* Build once, ship everywhere: the packed scene (one arena + its section table) as a flat blob.  A multi-process run
 * builds the kd-trees on ONE rank, broadcasts the blob and imports it on the others instead of G contended host builds
 * (the reference rebuilds everything per process: Scene::build kernel.cu:52-142 has no serialised form).  An imported scene
 * is ready for mipt_scene_commit (which then only uploads); the blob is only valid for the library version that wrote it. */
int64_t mipt_scene_export_size(mipt_scene *);                     /* bytes, after a commit (any n_devices); < 0 on error */
int  mipt_scene_export(mipt_scene *, void *blob, int64_t size);
int  mipt_scene_import(mipt_scene *, const void *blob, int64_t size);
/* kd build (Mesh::getKdTree/Node::makeNodeTree mesh.cu:32-155) + flat SoA packing (8-byte nodes,
 * triangle index lists) + one arena upload per device.  Replaces every copyToDevice call site
 * (copyToDevice.h:8-55; kernel.cu:61-139, mesh.cu:35-36,176,200, envMap.cuh:27,31, intersectable.cuh:12).
 * n_devices == 0 packs on the host only (no GPU needed; for inspecting the packed tree). */
int  mipt_scene_commit(mipt_scene *, const int *device_ids, int n_devices);

/* ---- rendering: replaces render<<<>>> + cudaMemcpy D2H (kernel.cu:310-346,397,416) ---- */
/* out_rgb_f32: host W*H*3, row r <-> camera Y=r (camera.cuh:18), value = sum_s trace_s / spp. With
 * several committed devices the sample set is partitioned s mod G, every device renders its share into a
 * device framebuffer and ONE RCCL reduce (sum, fp32, W*H*3) over xGMI brings them to slot 0 before the single
 * device-to-host copy that replaces kernel.cu:416 (no host-side sum; librccl missing = error). */
int  mipt_render(mipt_scene *, const mipt_render_config *, float *out_rgb_f32, mipt_stats *out);
/* Same on committed device slot `dev_slot`, ADDING into a caller-provided DEVICE buffer (W*H*3 fp32),
 * e.g. the tensor a caller then all-reduces over RCCL. Synchronises the scene stream before returning. */
int  mipt_render_device(mipt_scene *, int dev_slot, const mipt_render_config *, float *d_accum_rgb, mipt_stats *out);
/* Stream ordering of the *_device entry points (the reference has one stream, the default one: kernel.cu:397-416).  The
 * library works on a private stream per device slot.  On entry it makes that stream wait for everything the caller has
 * queued so far on `hip_stream` (a hipStream_t; NULL = the default stream, which is also what PyTorch uses unless told
 * otherwise), so a producer kernel or a zero-fill of the accumulation buffer needs no host synchronisation; on return the
 * library's work is complete (the stream has been synchronised), so consumers on any stream may follow.  */
int  mipt_scene_set_caller_stream(mipt_scene *, int dev_slot, void *hip_stream);

/* Closest hit for explicit rays: Scene::firstIntersect kernel.cu:145-160 on Ray(o, d) (ray.cuh:10 normalises d).
 * prim_id: -1 = miss, else triangles in mesh/face order, then spheres, then planes (Hit carries no id, hit.cuh:9-15). */
int  mipt_trace_closest(mipt_scene *, const float *ray_o, const float *ray_d, int64_t n, int32_t *prim_id, float *t,
                        mipt_stats *out, int flags);
/* Same with all four buffers resident on device slot `dev_slot` (n*3, n*3, n, n elements). */
int  mipt_trace_closest_device(mipt_scene *, int dev_slot, const float *d_ray_o, const float *d_ray_d, int64_t n,
                               int32_t *d_prim_id, float *d_t, mipt_stats *out, int flags);
/* No counterpart in the reference.  The same closest hit as mipt_trace_closest_device for callers that cannot allocate
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): it calls no hipMalloc / hipFree or
 * any allocating routine.  Its work buffers (padded rays, hit records, ray cursors) live in a fixed device array inside
 * the library (~40 MiB per GPU, held until the process ends), so n is limited to 1,048,576 rays per call (larger n: error).
 * No statistics.  Every call on one GPU shares that array: calls that overlap on one GPU race -- serialise them. */
int  mipt_trace_closest_device_noalloc(mipt_scene *, int dev_slot, const float *d_ray_o, const float *d_ray_d, int64_t n,
                                       int32_t *d_prim_id, float *d_t);
/* No counterpart in the reference (its megakernel has no queues, kernel.cu:310-346): the queue-binning stage the renderer runs
 * in front of the closest-hit kernel (csrc/queue_sort.h), exposed for tests and tools.  Orders d_vals[0, *d_count) by the
 * key_bits (14..32) significant bits of d_keys[] into d_out[0, *d_count): a permutation of the input in which the top 14 key
 * bits decrease at most *d_count / 262144 times (slices of a bucket that one crowded key range made huge) and neighbouring
 * keys are in order up to a tolerance that shrinks where the keys are dense (a sample sort with splitters from the queue's own
 * histogram; entries with equal keys come out in no particular order).  The length is read from DEVICE memory
 * (*d_count <= capacity); all four buffers live on device slot `dev_slot`. */
int  mipt_queue_bin_device(mipt_scene *, int dev_slot, const uint32_t *d_keys, const int32_t *d_vals, const int32_t *d_count,
                           int64_t capacity, int key_bits, int32_t *d_out);
/* EnvMap::getPixelColor envMap.cuh:50-89 evaluated on the device for n directions (host buffers). */
int  mipt_env_lookup(mipt_scene *, const float *dirs, int64_t n, float *rgb);

/* HDRLoader::load (HDRLoader/HDRLoader.cu:29-97) as a host-side service: decodes one Radiance .hdr file (adaptive RLE or flat
 * scanlines, `-Y h +X w`) to packed texels R | G<<8 | B<<16 | E<<24, rows in file order (top row first); fp32 value of a
 * component = val/256 * 2^(E-128) (convertComponent, :99-104; done in-kernel by the renderer).  texels == NULL: only w, h.
 * Unlike the reference, a truncated file, a run that leaves its scanline or a wrong scanline width is an error. */
int  mipt_hdr_load_rgbe(const char *path, int32_t *w, int32_t *h, uint32_t *texels /* w*h or NULL */);

/* ---- introspection of the packed scene (host side; valid after commit) ---------------- */
/* How many meshes' kd-trees the last commit built on the GPU (Node::makeNodeTree mesh.cu:32-136 level by level, kd_build_gpu.hip)
 * rather than on the host threads: meshes of >= 16,384 triangles when the commit has a device, unless MIPT_KD_BUILD=host.
 * Both builders produce the same tree, bit for bit. */
int  mipt_scene_gpu_built_meshes(mipt_scene *);
/* 1 when the last commit made the arena's big sections (nodes, cull boxes, step records, leaf records, triangle arrays) on the
 * GPU (pack_gpu.hip: the replacement of copyToDevice.h:8-55 and the per-leaf uploads of mesh.cu:35-36 for scenes with a mesh of
 * >= 16,384 triangles, unless MIPT_PACK=host), 0 when the host threads did.  The arena is the same bytes either way; a
 * GPU-packed scene keeps no host copy of it until an export / introspection call asks for one. */
int  mipt_scene_packed_on_gpu(mipt_scene *);
int  mipt_mesh_count(mipt_scene *);
int  mipt_kd_sizes(mipt_scene *, int mesh, int64_t *n_nodes, int64_t *n_refs, int64_t *n_tris);
/* Unpacks the 8-byte nodes of one mesh: boxes n*6 (minX,maxX,minY,maxY,minZ,maxZ), axis, left/right child
 * (-1 for leaves; indices local to the mesh), first/count of the leaf's reference range, refs (local triangle ids). */
int  mipt_kd_export(mipt_scene *, int mesh, float *boxes, int32_t *axis, int32_t *left, int32_t *right,
                    int32_t *first, int32_t *count, int32_t *refs);
int  mipt_mesh_triangles(mipt_scene *, int mesh, float *pos9, float *nrm9);
int64_t mipt_scene_device_bytes(mipt_scene *);   /* size of the packed arena */
double  mipt_scene_build_seconds(mipt_scene *);  /* kd build + pack time of the last commit */
double  mipt_scene_kd_seconds(mipt_scene *);     /* ... of which the kd-trees (host threads and / or GPU) */

const char *mipt_last_error(void);   /* replaces fprintf(stderr, ...) and continue (copyToDevice.h:10-17) */
const char *mipt_version(void);

#ifdef __cplusplus
}
#endif
#endif
