/*
 * oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a from-scratch CPU restatement of the
 * rendering semantics of fazakasreka/cuda-path-tracer (reference file:line
 * citations are on every function in oracle.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (cuda_path_tracer_amd/csrc -> libmipt.so) never includes,
 * links or calls anything in this directory.
 *
 * PARITY: the reference ships no tests, golden vectors or fixtures and building / running its
 * sources was denied by the environment, so at the level of single hits and samples this
 * restatement is UNPINNED by the reference: it rests on (a) analytic known-answer tests, (b) its own brute-force
 * cross-checks and (c) the published Philox4x32-10 vectors.  At the level of IMAGES it is pinned by the only outputs of
 * the reference that exist for this path, the two 200-sample 1024x1024 renders of its README: channel means within
 * 0.05 %, 64-pixel block means within 0.85 % rms (Monte-Carlo noise of two independent pictures included), wall blocks
 * within 0.2 % in sum (tests/golden/make_readme_fixture.py, tests/readme_util.py, tests/test_oracle_golden.py,
 * tests/test_gpu_golden.py).  That agreement needed ONE thing beyond the source text: the point on a ray,
 * `ray.start + ray.dir * t`, is evaluated with a fused multiply-add per component, as nvcc's default -fmad=true does
 * (ray_at in oracle.cpp); with two roundings the pictures were 2.5 % too bright in red and blue (the min-X face of a
 * cube, `minX <= position.x` without epsilon in mesh.cuh:132, lets fewer rays through).  Everything else is compiled
 * with -ffp-contract=off.
 */
#ifndef MIPT_ORACLE_H
#define MIPT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* mirrors mipt_render_config (include/mipt.h) field for field */
typedef struct orc_render_config {
    int32_t width, height;
    int32_t spp;            /* total samples per pixel (normalisation constant, kernel.cu:343) */
    int32_t cam_depth;      /* camera sub-path vertices (maxdepth, constants.cuh:8) */
    int32_t light_depth;    /* light sub-path vertices; 0 = "simple" path tracing */
    int32_t use_envmap;     /* 1: add env radiance on camera-ray miss (build-defined); 2: ... and sample the map as
                               a light at every camera vertex (the miss term then only at vertex 0 and behind mirror bounces) */
    uint32_t seed;
    int32_t sample_begin, sample_end, sample_stride; /* samples s = begin, begin+stride, ... < end */
    int32_t pixel_stride;   /* oracle only: render every k-th pixel (others left 0); 1 = all */
} orc_render_config;

typedef struct orc_counters {   /* traversal counters, totals over a call */
    int64_t rays;
    int64_t nodes;          /* kd nodes made current (each costs one 8-byte node read) */
    int64_t pops;           /* stack pops incl. the root (each costs one box re-read)  */
    int64_t refs;           /* leaf triangle references read                            */
    int64_t tris;           /* triangle position fetches (= refs in this traversal)     */
    int64_t hits;           /* rays with a mesh-triangle closest hit                    */
    int64_t box_tests;      /* Box::intersect evaluations (reference layout cost)       */
    int64_t stack_overflow; /* rays that would overflow the reference's int[15] (Q9)    */
    int64_t max_stack;
} orc_counters;

orc_scene *orc_scene_create(void);
void orc_scene_destroy(orc_scene *);
int orc_scene_set_camera(orc_scene *, const float eye[3], const float lookat[3], const float vup[3], float fov_rad);
int orc_scene_add_material(orc_scene *, const float diffuse[3], const float mirror[3]);
int orc_scene_add_light(orc_scene *, const float location[3], const float power[3]);
int orc_scene_add_plane(orc_scene *, const float point[3], const float normal[3], int material);
int orc_scene_add_sphere(orc_scene *, const float center[3], float radius, int material);
int orc_scene_add_mesh(orc_scene *, const float *pos, const float *nrm, int64_t n_tris, int material);
int orc_scene_add_mesh_obj(orc_scene *, const char *path, const float position[3], const float rotate[3],
                           const float scale[3], int material);
int orc_scene_set_envmap_rgbe_dir(orc_scene *, const char *folder);
int orc_scene_set_envmap_f32(orc_scene *, const float *faces6, int w, int h);
int orc_scene_commit(orc_scene *, int n_threads);

/* closest hit on n rays; dir is normalised the way the reference Ray ctor does.
 * prim_id: -1 miss, else global id (triangles in mesh order, then spheres, then planes).
 * brute != 0: test every triangle of every mesh instead of walking the kd-tree.       */
int orc_trace_closest(orc_scene *, const float *ray_o, const float *ray_d, int64_t n, int32_t *prim_id,
                      float *t, orc_counters *ctr, int brute, int n_threads);
int orc_render(orc_scene *, const orc_render_config *, float *out_rgb, orc_counters *ctr, int n_threads);

/* Why is this sample non-finite?  Re-traces ONE (pixel, sample) of orc_render and reports the first contribution
 * that was not finite and which operand of the reference's expression caused it. */
enum {
    ORC_NF_NONE = 0,
    ORC_NF_DIRECT = 1,        /* kernel.cu:254  (cam_throughput / cam_pdf) * (radiance * brdf * cos)           */
    ORC_NF_LIGHT_WEIGHT = 2,  /* kernel.cu:277  light_throughput[j] / light_pdf[j]  (pdf == 0: x/0 or 0/0)     */
    ORC_NF_GEOMETRY = 3,      /* kernel.cu:274  cosCam * cosLight / (dist * dist)                               */
    ORC_NF_CAMERA_WEIGHT = 4, /* kernel.cu:276  cam_throughput * brdf * cosCam / cam_pdf                        */
    ORC_NF_PRODUCT = 5        /* every factor finite, the product overflows / inf * 0                           */
};
typedef struct orc_sample_probe {
    int32_t first_cause, light_vertex, cam_vertex;
    int32_t sqrt_negative;     /* SampleDiffuse calls with 1 - x*x - y*y < 0 (samplers.cuh:19 -> z = NaN)      */
    int32_t zero_pdf_vertices; /* light vertices stored with pdf == 0                                           */
    int32_t rng_draws;
    float operand[4];          /* DIRECT: Tc.x, pc, radiance.x, cos;  connections: T[j].x, pdf[j], G, cw.x       */
} orc_sample_probe;
int orc_explain_sample(orc_scene *, const orc_render_config *, int64_t pixel, int32_t sample, float rgb[3], orc_sample_probe *out);

/* kd-tree export for structural goldens. Arrays may be NULL to query sizes only. */
int orc_mesh_count(orc_scene *);
int orc_kd_sizes(orc_scene *, int mesh, int64_t *n_nodes, int64_t *n_refs, int64_t *n_tris);
int orc_kd_export(orc_scene *, int mesh, float *boxes /*n*6*/, int32_t *axis, int32_t *left, int32_t *right,
                  int32_t *first, int32_t *count, int32_t *refs);
int orc_mesh_triangles(orc_scene *, int mesh, float *pos9, float *nrm9);

/* unit-level entry points for known-answer tests */
void orc_normalize(const float v[3], float out[3]);
float orc_triangle_intersect(const float pos9[9], const float o[3], const float d_unit[3]);
float orc_box_intersect(const float box6[6], const float o[3], const float d_unit[3]);
int orc_box_contains(const float box6[6], const float p[3]);
float orc_plane_intersect(const float point[3], const float normal_unit[3], const float o[3], const float d_unit[3]);
float orc_sphere_intersect(const float center[3], float radius, const float o[3], const float d_unit[3]);
void orc_camera_ray(orc_scene *, float X, float Y, int W, int H, float o[3], float d[3]);
void orc_sincos(float x, float *s, float *c);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_rng_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t draw);
int orc_env_lookup(orc_scene *, const float *dirs, int64_t n, float *rgb, int32_t *face, int32_t *pos);
/* the cube map as a light (use_envmap == 2, build-defined): sampling tables marg[6 H], cond[6 H W] (may be NULL); 1 = usable */
int orc_env_light_tables(orc_scene *, float *marg, float *cond);
int orc_env_light_sample(orc_scene *, float u1, float u2, int32_t frc[3], float pmf_sa[2], float dir[3]);
int orc_hdr_load(const char *path, int *w, int *h, float *cols /* may be NULL */);
void orc_srt_matrix(const float scale[3], const float rotate[3], const float translate[3], float m16[16],
                    float inv_t16[16]);
const char *orc_last_error(void);
/* Mutation tests (tests/test_pin_mutations.py): bit i flips quirk i (enum MUT_* in oracle.cpp) to what a clean implementation
 * would do.  Honoured only by the -DORC_MUTATIONS builds (liboracle_mut*.so); the checker refuses a non-zero mask. */
int orc_set_mutations(uint32_t mask);
int orc_mutation_count(void);

#ifdef __cplusplus
}
#endif
#endif
