// scene.cpp -- host side of libmipt: scene description, asset readers, kd build orchestration and
// packing of the whole scene into one flat arena (the copyToDevice replacement).
//
// Reference behaviour restated here (paths relative to /root/reference/cuda-path-tracer/):
//   Camera::set                camera.cuh:9-16
//   (OBJ / SRT and Radiance RGBE readers: assets.cpp)
//   copyToDevice call sites    copyToDevice/copyToDevice.h:8-55 -> ONE arena, uploaded by api.hip
#include "scene_host.h"
#include "pack_gpu.h"
#include "shade.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace mipt {

// Static chunking of an index range over the builder's threads (the pack loops are independent per node / reference).
template <class F>
static void parallel_for(int64_t n, int n_threads, F &&body) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n / 4096));
    if (nt <= 1) { body((int64_t)0, n); return; }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (int k = 0; k < nt; k++) {
        const int64_t b = k * chunk, e = std::min<int64_t>(n, b + chunk);
        if (b >= e) break;
        th.emplace_back([&body, b, e] { body(b, e); });
    }
    for (auto &x : th) x.join();
}

// ------------------------------------------------------------------------------------------ cube map as a light
// Sampling tables of the cube map (use_envmap == 2; shade.h env_light_sample reads them): weight of a texel = average(rgb) x
// solid angle, summed in double in index order, stored as float CDFs whose last entry is 1.  marg has one entry more than
// rows: 1.0f when the map can be sampled, 0.0f when it is black (or absent).  NaN, negative and infinite texels get no
// probability.  Rows are independent: they are spread over the builder's threads, the result does not depend on how.
static void build_env_light(const HostEnv &env, float *marg, float *cond, int n_threads) {
    const int w = env.w, h = env.h;
    const size_t rows = (size_t)6 * h;
    std::vector<double> row_sum(rows, 0.0);
    DScene texels;  // just enough of a view for env_texel()
    memset(&texels, 0, sizeof(texels));
    if (env.rgbe.empty()) texels.env_f32 = env.f32.data();
    else texels.env_rgbe = env.rgbe.data();
    auto do_rows = [&](size_t r_begin, size_t r_end) {
        std::vector<double> acc((size_t)w);
        for (size_t R = r_begin; R < r_end; R++) {
            const int face = (int)(R / h), row = (int)(R % h);
            double run = 0.0;
            for (int c = 0; c < w; c++) {
                f3 dir;
                float sa;
                env_texel_geometry(face, row, c, w, h, dir, sa);
                const f3 px = env_texel(texels, (size_t)face * w * h + (size_t)row * w + c);
                const double lum = ((double)px.x + (double)px.y + (double)px.z) / 3.0;
                run += (lum > 0.0 && lum < (double)INFINITY) ? lum * (double)sa : 0.0;
                acc[(size_t)c] = run;
            }
            row_sum[R] = run;
            float *cr = cond + R * w;
            for (int c = 0; c < w; c++) cr[c] = 1.0f;
            if (run > 0.0)
                for (int c = 0; c + 1 < w; c++) cr[c] = (float)(acc[(size_t)c] / run);
        }
    };
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), rows / 64));
    if (nt <= 1) {
        do_rows(0, rows);
    } else {
        std::vector<std::thread> th;
        const size_t chunk = (rows + nt - 1) / nt;
        for (size_t k = 0; k < nt; k++) {
            const size_t b = k * chunk, e = std::min(rows, b + chunk);
            if (b < e) th.emplace_back([&do_rows, b, e] { do_rows(b, e); });
        }
        for (auto &x : th) x.join();
    }
    for (size_t R = 0; R <= rows; R++) marg[R] = 1.0f;
    double total = 0.0;
    for (size_t R = 0; R < rows; R++) total += row_sum[R];
    if (!(total > 0.0)) { marg[rows] = 0.0f; return; }  // a black map is no light
    double run = 0.0;
    for (size_t R = 0; R + 1 < rows; R++) {
        run += row_sum[R];
        marg[R] = (float)(run / total);
    }
}

// ------------------------------------------------------------------------------------------ camera
void camera_set(HostScene &s, const float eye[3], const float lookat[3], const float vup[3], float fov) {
    const f3 e = mk3(eye[0], eye[1], eye[2]), l = mk3(lookat[0], lookat[1], lookat[2]), up = mk3(vup[0], vup[1], vup[2]);
    const f3 w = e - l;
    const float f = len3(w);
    const f3 right = norm3(cross3(up, w)) * f * tanf(fov / 2.0f);
    const f3 upv = norm3(cross3(w, right)) * f * tanf(fov / 2.0f);
    const f3 v[4] = {e, l, right, upv};
    float *dst[4] = {s.cam_eye, s.cam_lookat, s.cam_right, s.cam_up};
    for (int i = 0; i < 4; i++) { dst[i][0] = v[i].x; dst[i][1] = v[i].y; dst[i][2] = v[i].z; }
    s.cam_set = true;
}

// ------------------------------------------------------------------------------------------ packing
namespace {
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
}  // namespace

bool HostScene::pack(int n_threads, std::string &err) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads <= 0) {
        const char *e = getenv("MIPT_BUILD_THREADS");
        n_threads = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (n_threads <= 0) n_threads = 1;
    }
    for (const HostPlane &p : planes)
        if (p.material < 0 || p.material >= (int)materials.size()) { err = "plane references unknown material"; return false; }
    // kd-trees.  The top levels of a build are serial (one thread sorts the root's coordinates, two the next level's,
    // ...), so several meshes are built side by side, each with its share of the thread budget: 8 x 1.3 M triangles on
    // 16 threads: 4.5 s one after another with 16 threads each, 1.7 s side by side with 2 each.
    std::vector<KdTree> trees(meshes.size());
    total_tris = total_nodes = total_refs = 0;
    for (const HostMesh &hm : meshes)
        if (hm.n_tris <= 0) { err = "empty mesh"; return false; }
    // positions of the large meshes go to the device ONCE: the GPU kd build reads them there and the GPU packer after it
    std::vector<void *> d_pos(meshes.size(), nullptr);
    struct PosGuard {
        std::vector<void *> &v; int dev;
        ~PosGuard() { for (void *p : v) gpu_free(dev, p); }
    } pos_guard{d_pos, build_device};
    constexpr int64_t kGpuBuildMinTris = 16384;  // below that the host build takes a few milliseconds
    {
        std::vector<size_t> on_host;
        meshes_built_on_gpu = 0;
        for (size_t m = 0; m < meshes.size(); m++) {
            if (build_device >= 0 && meshes[m].n_tris >= kGpuBuildMinTris) {
                std::string gpu_err;
                d_pos[m] = gpu_upload(build_device, meshes[m].pos.data(), meshes[m].pos.size() * sizeof(float), gpu_err);
                if (d_pos[m] && build_kd_tree_gpu(build_device, meshes[m].pos.data(), meshes[m].n_tris, trees[m], gpu_err, d_pos[m])) {
                    meshes_built_on_gpu++;
                } else {  // e.g. out of device memory: the host builder makes the identical tree, only slower
                    if (getenv("MIPT_TIMING")) fprintf(stderr, "[mipt] GPU kd build of mesh %zu failed (%s): building it on the host\n", m, gpu_err.c_str());
                    trees[m] = KdTree();
                    on_host.push_back(m);
                }
            } else {
                on_host.push_back(m);
            }
        }
        const int side_by_side = (int)std::max<size_t>(1, std::min<size_t>(on_host.size(), (size_t)n_threads));
        const int threads_each = std::max(1, n_threads / side_by_side);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < on_host.size(); k = next.fetch_add(1))
                build_kd_tree(meshes[on_host[k]].pos.data(), meshes[on_host[k]].n_tris, trees[on_host[k]], threads_each);
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < side_by_side; w++) pool.emplace_back(worker);
        worker();
        for (std::thread &th : pool) th.join();
    }
    for (size_t m = 0; m < meshes.size(); m++) {
        HostMesh &hm = meshes[m];
        hm.tri_base = total_tris;
        hm.node_base = total_nodes;
        hm.ref_base = total_refs;
        hm.n_nodes = (int64_t)trees[m].nodes.size();
        hm.n_refs = (int64_t)trees[m].refs.size();
        hm.degenerate_nodes = trees[m].degenerate_nodes;
        hm.n_leaves = trees[m].n_leaves;
        hm.max_leaf = trees[m].max_leaf;
        hm.max_depth = trees[m].max_depth;
        total_tris += hm.n_tris;
        total_nodes += hm.n_nodes;
        total_refs += hm.n_refs;
    }
    if (total_nodes >= (int64_t)kIdxMask || total_refs >= (int64_t)INT32_MAX || total_tris >= (int64_t)(INT32_MAX / 4)) {
        err = "scene too large for 30-bit node / 31-bit triangle indices";
        return false;
    }
    const double t_trees = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    kd_seconds = t_trees;
    ArenaLayout L;
    size_t off = 0;
    auto section = [&](size_t bytes) { size_t o = off; off = align_up(off + std::max<size_t>(bytes, 16), 256); return o; };
    L.nodes = section((size_t)total_nodes * sizeof(KdNode));
    L.boxes = section((size_t)total_nodes * 32);
    L.tight = section((size_t)total_nodes * 32);
    L.tight_pairs = section((size_t)(total_nodes / 2 + 1) * 64);
    L.refs = section((size_t)total_refs * 4);
    L.leaf_recs = section((size_t)total_refs * 64);
    L.leaf_spheres = section(((size_t)total_refs / 2 + 4) * sizeof(SpherePair));  // + padding the scan may over-read
    L.tri_pos = section((size_t)total_tris * 48);
    L.tri_nrm = section((size_t)total_tris * 48);
    L.meshes = section(meshes.size() * sizeof(DMesh));
    L.mesh_order = section(3 * meshes.size() * 4);
    L.spheres = section(spheres.size() * 32);
    L.planes = section(planes.size() * 32);
    L.lights = section(lights.size() * 32);
    L.materials = section(materials.size() * 32);
    const size_t env_face_bytes = env.present() ? (env.rgbe.empty() ? env.f32.size() * 4 : env.rgbe.size() * 4) : 0;
    const EnvSection env_sec = env_section(env.w, env.h, !env.rgbe.empty());
    const size_t env_bytes = env.present() ? env_sec.bytes : 0;
    L.env = section(env_bytes);
    L.total = off;
    // The big sections [0, L.meshes).  With a device and a large mesh they are made ON that device (pack_gpu.hip: the same
    // bytes, asserted by tests/test_gpu_kd_build.py) and the host keeps no copy of the arena; else -- or should the device
    // refuse -- on the host threads below.  MIPT_PACK=host forces the host packer.
    const bool force_recheck = getenv("MIPT_FORCE_RECHECK") && atoi(getenv("MIPT_FORCE_RECHECK")) != 0;
    std::vector<float> root_tight(8 * meshes.size() + 8, 0.0f);  // root cull box of every mesh (for mesh_order)
    if (dev_arena) { gpu_free(dev_arena_device, dev_arena); dev_arena = nullptr; }
    packed_on_gpu = false;
    {
        bool large = false;
        for (const HostMesh &hm : meshes) large |= hm.n_tris >= kGpuBuildMinTris;
        const char *pk = getenv("MIPT_PACK");
        if (build_device >= 0 && large && !(pk && strcmp(pk, "host") == 0)) {
            std::vector<GpuPackMesh> gm(meshes.size());
            for (size_t m = 0; m < meshes.size(); m++)
                gm[m] = GpuPackMesh{meshes[m].pos.data(), meshes[m].nrm.data(), meshes[m].n_tris, &trees[m], meshes[m].tri_base,
                                    meshes[m].node_base, meshes[m].ref_base, d_pos[m]};
            std::string gpu_err;
            void *d_arena = nullptr;
            if (gpu_pack_scene(build_device, gm, L, total_refs, force_recheck, &d_arena, root_tight.data(), gpu_err)) {
                dev_arena = d_arena;
                dev_arena_device = build_device;
                packed_on_gpu = true;
            } else if (getenv("MIPT_TIMING")) {
                fprintf(stderr, "[mipt] GPU pack failed (%s): packing on the host\n", gpu_err.c_str());
            }
        }
    }
    std::vector<uint8_t> tail(L.total - L.meshes, 0);  // the small sections behind the big ones, always made here
    uint8_t *A = nullptr;
    if (!packed_on_gpu) {
    arena.assign(L.total, 0);
    A = arena.data();
    {  // slots no reference owns (padding) must never pass: R^2 = -inf
        float *ls = reinterpret_cast<float *>(A + L.leaf_spheres);
        const size_t n_pairs = (size_t)total_refs / 2 + 4;
        for (size_t i = 0; i < n_pairs; i++) ls[8 * i + 6] = ls[8 * i + 7] = -INFINITY;
    }
    for (size_t m = 0; m < meshes.size(); m++) {
        const HostMesh &hm = meshes[m];
        const KdTree &t = trees[m];
        KdNode *nodes = reinterpret_cast<KdNode *>(A + L.nodes) + hm.node_base;
        for (int64_t i = 0; i < hm.n_nodes; i++) {
            KdNode k = t.nodes[(size_t)i];
            if ((k.b >> 30) == 3u) k.a += (uint32_t)hm.ref_base;           // leaf: first ref -> global
            else k.b = (k.b & ~kIdxMask) | ((k.b & kIdxMask) + (uint32_t)hm.node_base);  // interior: left child -> global
            nodes[i] = k;
        }
        memcpy(A + L.boxes + (size_t)hm.node_base * 32, t.boxes.data(), (size_t)hm.n_nodes * 32);
        // MIPT_FORCE_RECHECK=1 (tests): flag every interior node, so the full re-evaluation path is exercised everywhere
        for (int64_t i = 0; i < hm.n_nodes; i++) {  // kRecheckBit: split plane outside the node's own extent (see scene_types.h)
            const KdNode k = t.nodes[(size_t)i];
            const uint32_t ax = k.b >> 30;
            if (ax == 3u) continue;
            const float *nb = t.boxes.data() + 8 * (size_t)i;
            float sv;
            memcpy(&sv, &k.a, 4);
            if (force_recheck || !(nb[2 * ax] <= sv && sv <= nb[2 * ax + 1])) nodes[i].b |= kRecheckBit;
        }
        // per-leaf cull box: (AABB of the leaf's triangles) intersected with the leaf box.  An accepted hit point lies
        // on one of those triangles AND inside the leaf box (mesh.cuh:293-296), so a ray that misses this box -- tested
        // conservatively, see leaf_culled() -- cannot find anything in the leaf and its candidates are skipped.
        float *tight = reinterpret_cast<float *>(A + L.tight) + (size_t)hm.node_base * 8;
        parallel_for(hm.n_nodes, n_threads, [&](int64_t i_begin, int64_t i_end) {
        for (int64_t i = i_begin; i < i_end; i++) {
            const KdNode k = t.nodes[(size_t)i];
            float *tb = tight + 8 * i;
            const float *nb = t.boxes.data() + 8 * (size_t)i;
            for (int a = 0; a < 8; a++) tb[a] = nb[a];
            if ((k.b >> 30) != 3u) continue;
            const int cnt = (int)(k.b & kIdxMask);
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int r = 0; r < cnt; r++) {
                const float *v = hm.pos.data() + (size_t)t.refs[(size_t)k.a + r] * 9;
                for (int c = 0; c < 9; c++) {
                    lo[c % 3] = std::min(lo[c % 3], v[c]);
                    hi[c % 3] = std::max(hi[c % 3], v[c]);
                }
            }
            for (int a = 0; a < 3; a++) {  // empty leaf: lo=+inf, hi=-inf -> every ray is culled (nothing to find)
                tb[2 * a] = std::max(nb[2 * a], lo[a]);
                tb[2 * a + 1] = std::min(nb[2 * a + 1], hi[a]);
            }
        }
        });
        // interior nodes: union of the children's cull boxes (children are stored after their parent), so a ray that
        // misses it cannot find anything in the whole subtree and the stacked subtree can be dropped unvisited
        for (int64_t i = hm.n_nodes - 1; i >= 0; i--) {
            const KdNode k = t.nodes[(size_t)i];
            if ((k.b >> 30) == 3u) continue;
            const float *l = tight + 8 * (size_t)(k.b & kIdxMask), *r = l + 8;
            float *tb = tight + 8 * i;
            for (int a = 0; a < 3; a++) {
                tb[2 * a] = std::min(l[2 * a], r[2 * a]);
                tb[2 * a + 1] = std::max(l[2 * a + 1], r[2 * a + 1]);
            }
        }
        // the same boxes once more, interleaved per sibling pair for the packed cull test of descend_fast()
        if ((hm.node_base & 1) != 0) { err = "internal: odd node base"; return false; }
        float *tpair = reinterpret_cast<float *>(A + L.tight_pairs);
        parallel_for(hm.n_nodes, n_threads, [&](int64_t i_begin, int64_t i_end) {
        for (int64_t i = i_begin; i < i_end; i++) {
            const KdNode k = t.nodes[(size_t)i];
            if ((k.b >> 30) == 3u) continue;
            const size_t lc = (size_t)(k.b & kIdxMask);
            const float *l = tight + 8 * lc, *r = l + 8;
            float *rec = tpair + 16 * (((size_t)hm.node_base + lc) >> 1);
            // stored GROWN by 2^-12 x max(|lo|, |hi|, 1), rounded outwards (see pair_culled for why that suffices); an
            // empty box stays (+inf, -inf)
            auto grow = [](float lo, float hi, float &glo, float &ghi) {
                if (!(lo <= hi)) { glo = INFINITY; ghi = -INFINITY; return; }
                const float mg = 2.44140625e-4f * std::max(std::max(std::fabs(lo), std::fabs(hi)), 1.0f);
                glo = std::nextafter(lo - mg, -INFINITY);
                ghi = std::nextafter(hi + mg, INFINITY);
            };
            const bool l_empty = !(l[0] <= l[1]) || !(l[2] <= l[3]) || !(l[4] <= l[5]);
            const bool r_empty = !(r[0] <= r[1]) || !(r[2] <= r[3]) || !(r[4] <= r[5]);
            for (int a = 0; a < 3; a++) {
                grow(l_empty ? INFINITY : l[2 * a], l_empty ? -INFINITY : l[2 * a + 1], rec[4 * a + 0], rec[4 * a + 2]);
                grow(r_empty ? INFINITY : r[2 * a], r_empty ? -INFINITY : r[2 * a + 1], rec[4 * a + 1], rec[4 * a + 3]);
            }
            memcpy(rec + 12, &nodes[lc], 2 * sizeof(KdNode));  // the children's nodes (global indices, flags set) ride along
        }
        });
        int32_t *refs = reinterpret_cast<int32_t *>(A + L.refs) + hm.ref_base;
        for (int64_t i = 0; i < hm.n_refs; i++) refs[i] = t.refs[(size_t)i] + (int32_t)hm.tri_base;
        // leaf-ordered plane records: N = cross(b-a, c-a), dd = -dot(N, a) exactly as Triangle::intersect
        // evaluates them per candidate (mesh.cuh:40-42,52); same fp32 operations, done once here
        float *lr = reinterpret_cast<float *>(A + L.leaf_recs) + (size_t)hm.ref_base * 16;  // 64 B per reference: plane, a, b, c
        parallel_for(hm.n_refs, n_threads, [&](int64_t i_begin, int64_t i_end) {
        for (int64_t i = i_begin; i < i_end; i++) {
            const float *v = hm.pos.data() + (size_t)t.refs[(size_t)i] * 9;
            const f3 a = mk3(v[0], v[1], v[2]), b = mk3(v[3], v[4], v[5]), c = mk3(v[6], v[7], v[8]);
            const f3 N = cross3(b - a, c - a);
            float *rec = lr + 16 * i;
            rec[0] = N.x; rec[1] = N.y; rec[2] = N.z;
            rec[3] = -dot3(N, a);
            for (int q = 0; q < 3; q++)
                for (int k = 0; k < 3; k++) rec[4 + 4 * q + k] = v[3 * q + k];
            // a.w: the GLOBAL triangle id (int bits) -- the winner of a leaf scan reads it from the line its test just touched
            // instead of from `refs`, a cold array (b.w, c.w stay 0)
            const int32_t gtri = t.refs[(size_t)i] + (int32_t)hm.tri_base;
            memcpy(rec + 7, &gtri, 4);
            rec[11] = 0.0f;
            rec[15] = 0.0f;
        }
        });
        // leaf-ordered candidate spheres (no counterpart in the reference; they change no result): a hit the leaf
        // loop accepts lies on the triangle (up to the rounding of the edge tests) and inside the leaf box (up to its
        // 1e-5 slack), i.e. inside both the triangle's smallest enclosing sphere and the sphere around
        // AABB(triangle) ∩ cull box of the leaf.  The smaller of the two, grown by 2^-10 relative + 4e-5 + 2^-16 x (largest
        // coordinate, at least 1) absolute, is what leaf_scan() tests the ray against before it evaluates a candidate.
        float *ls = reinterpret_cast<float *>(A + L.leaf_spheres);  // indexed by GLOBAL reference (pairs interleaved)
        parallel_for(hm.n_nodes, n_threads, [&](int64_t i_begin, int64_t i_end) {
        for (int64_t i = i_begin; i < i_end; i++) {
            const KdNode k = t.nodes[(size_t)i];
            if ((k.b >> 30) != 3u) continue;
            const float *tb = tight + 8 * i;
            const int cnt = (int)(k.b & kIdxMask);
            for (int r = 0; r < cnt; r++) {
                const size_t ri = (size_t)k.a + r;
                const size_t gi = (size_t)hm.ref_base + ri;
                const float *v = hm.pos.data() + (size_t)t.refs[ri] * 9;
                double P[3][3], mag = 1.0;
                for (int q = 0; q < 3; q++)
                    for (int c = 0; c < 3; c++) { P[q][c] = v[3 * q + c]; mag = std::max(mag, std::fabs(P[q][c])); }
                auto dist2 = [](const double *x, const double *y) {
                    return (x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) + (x[2] - y[2]) * (x[2] - y[2]);
                };
                // smallest enclosing sphere of the triangle: around the longest edge if the opposite vertex fits, else
                // the circumscribed one (acute triangle)
                double cen[3], rad2;
                {
                    int e0 = 0, e1 = 1, eo = 2;
                    double best = dist2(P[0], P[1]);
                    if (dist2(P[1], P[2]) > best) { best = dist2(P[1], P[2]); e0 = 1; e1 = 2; eo = 0; }
                    if (dist2(P[2], P[0]) > best) { best = dist2(P[2], P[0]); e0 = 2; e1 = 0; eo = 1; }
                    for (int c = 0; c < 3; c++) cen[c] = 0.5 * (P[e0][c] + P[e1][c]);
                    rad2 = 0.25 * best;
                    if (dist2(cen, P[eo]) > rad2) {
                        const double *a = P[0], *b = P[1], *c = P[2];
                        double ab[3], ac[3], n[3];
                        for (int q = 0; q < 3; q++) { ab[q] = b[q] - a[q]; ac[q] = c[q] - a[q]; }
                        n[0] = ab[1] * ac[2] - ab[2] * ac[1]; n[1] = ab[2] * ac[0] - ab[0] * ac[2]; n[2] = ab[0] * ac[1] - ab[1] * ac[0];
                        const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                        const double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], ac2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
                        if (n2 > 0.0) {
                            // circumcentre = a + (|ac|^2 (n x ab) + |ab|^2 (ac x n)) / (2 |n|^2)
                            double u[3], w[3];
                            u[0] = n[1] * ab[2] - n[2] * ab[1]; u[1] = n[2] * ab[0] - n[0] * ab[2]; u[2] = n[0] * ab[1] - n[1] * ab[0];  // n x ab
                            w[0] = ac[1] * n[2] - ac[2] * n[1]; w[1] = ac[2] * n[0] - ac[0] * n[2]; w[2] = ac[0] * n[1] - ac[1] * n[0];  // ac x n
                            double cc[3];
                            for (int q = 0; q < 3; q++) cc[q] = a[q] + (ac2 * u[q] + ab2 * w[q]) / (2.0 * n2);
                            const double r2 = std::max(dist2(cc, a), std::max(dist2(cc, b), dist2(cc, c)));
                            if (r2 <= 2.0 * rad2) {  // an acute triangle's circumradius^2 is <= 4/3 of (longest edge / 2)^2
                                for (int q = 0; q < 3; q++) cen[q] = cc[q];
                                rad2 = r2;
                            } else {
                                rad2 = std::max(rad2, dist2(cen, P[eo]));
                            }
                        } else {
                            rad2 = std::max(rad2, dist2(cen, P[eo]));
                        }
                    }
                }
                // sphere around AABB(triangle) ∩ leaf cull box (good for triangles much larger than their leaf)
                {
                    double lo[3], hi[3];
                    bool empty = false;
                    for (int c = 0; c < 3; c++) {
                        lo[c] = std::max((double)tb[2 * c], std::min(P[0][c], std::min(P[1][c], P[2][c])));
                        hi[c] = std::min((double)tb[2 * c + 1], std::max(P[0][c], std::max(P[1][c], P[2][c])));
                        mag = std::max(mag, std::max(std::fabs((double)tb[2 * c]), std::fabs((double)tb[2 * c + 1])));
                        if (!(lo[c] <= hi[c])) empty = true;
                    }
                    if (!empty) {
                        double bc[3], br2 = 0.0;
                        for (int c = 0; c < 3; c++) { bc[c] = 0.5 * (lo[c] + hi[c]); br2 += 0.25 * (hi[c] - lo[c]) * (hi[c] - lo[c]); }
                        if (br2 < rad2) { rad2 = br2; for (int c = 0; c < 3; c++) cen[c] = bc[c]; }
                    }
                }
                if (!std::isfinite(mag)) mag = 1.0;
                // margin: sqrt(3) x the 1e-5 box slack (1.8e-5) + the rounding of P = o + d t and of the edge tests (a few
                // ulp of the coordinates, < 1e-6 x mag for origins within the scene's scale; leaf_scan() adds a term for
                // far origins) -- covered 2x and 15x by 4e-5 + 2^-16 mag.  (2^-13 mag, the first choice, made the spheres
                // of a 1.3 M-triangle mesh 10 % wider than the triangles need: 15 % more candidates to evaluate.)
                const double R = std::sqrt(rad2) * (1.0 + 1.0 / 1024.0) + 4e-5 + mag / 65536.0;
                float R2 = (float)(R * R);
                R2 = std::nextafter(R2, INFINITY);
                if (!(R2 >= 0.0f) || !std::isfinite(R2) || !std::isfinite(cen[0]) || !std::isfinite(cen[1]) || !std::isfinite(cen[2])) {
                    cen[0] = cen[1] = cen[2] = 0.0;  // non-finite input: never reject (centre 0, infinite radius)
                    R2 = INFINITY;
                }
                float *rec = ls + 8 * (gi >> 1) + (gi & 1);
                rec[0] = (float)cen[0]; rec[2] = (float)cen[1]; rec[4] = (float)cen[2];
                rec[6] = R2;
            }
        }
        });
        float *tp = reinterpret_cast<float *>(A + L.tri_pos) + (size_t)hm.tri_base * 12;
        float *tn = reinterpret_cast<float *>(A + L.tri_nrm) + (size_t)hm.tri_base * 12;
        parallel_for(hm.n_tris * 3, n_threads, [&](int64_t i_begin, int64_t i_end) {
        for (int64_t i = i_begin; i < i_end; i++) {
            for (int k = 0; k < 3; k++) {
                tp[4 * i + k] = hm.pos[(size_t)(3 * i + k)];
                tn[4 * i + k] = hm.nrm[(size_t)(3 * i + k)];
            }
        }
        });
        memcpy(root_tight.data() + 8 * m, tight, 32);
    }
    } else {
        arena.clear();
        arena.shrink_to_fit();
    }
    uint8_t *T = tail.data() - L.meshes;  // T + L.<small section> addresses the tail
    for (size_t m = 0; m < meshes.size(); m++) {
        const HostMesh &hm = meshes[m];
        DMesh dm;
        dm.root = (int32_t)hm.node_base;
        dm.tri_base = (int32_t)hm.tri_base;
        dm.n_tris = (int32_t)hm.n_tris;
        dm.material = hm.material;
        reinterpret_cast<DMesh *>(T + L.meshes)[m] = dm;
    }
    {  // evaluation order of the meshes for the extend kernel (any order gives Scene::firstIntersect's answer as long
       // as equal hits go to the lower triangle id, i.e. the earlier mesh, kernel.cu:147-150)
        int32_t *mo = reinterpret_cast<int32_t *>(T + L.mesh_order);
        std::vector<int> idx(meshes.size());
        for (int ax = 0; ax < 3; ax++) {
            for (size_t m = 0; m < meshes.size(); m++) idx[m] = (int)m;
            auto centre = [&](int m) {
                const float *b = root_tight.data() + 8 * (size_t)m;
                const float lo = b[2 * ax], hi = b[2 * ax + 1];
                return (lo <= hi) ? 0.5f * (lo + hi) : 0.0f;  // empty mesh: anywhere
            };
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return centre(a) < centre(b); });
            for (size_t m = 0; m < meshes.size(); m++) mo[(size_t)ax * meshes.size() + m] = (int32_t)meshes[(size_t)idx[m]].node_base;
        }
    }
    float *sp = reinterpret_cast<float *>(T + L.spheres);
    for (size_t i = 0; i < spheres.size(); i++) {
        const HostSphere &s = spheres[i];
        const float v[8] = {s.center[0], s.center[1], s.center[2], s.radius, i2f(s.material), 0, 0, 0};
        memcpy(sp + 8 * i, v, 32);
    }
    float *pl = reinterpret_cast<float *>(T + L.planes);
    for (size_t i = 0; i < planes.size(); i++) {
        const HostPlane &p = planes[i];
        const float v[8] = {p.point[0], p.point[1], p.point[2], i2f(p.material), p.normal[0], p.normal[1], p.normal[2], 0};
        memcpy(pl + 8 * i, v, 32);
    }
    float *li = reinterpret_cast<float *>(T + L.lights);
    for (size_t i = 0; i < lights.size(); i++) {
        const HostLight &l = lights[i];
        const float v[8] = {l.loc[0], l.loc[1], l.loc[2], 0, l.power[0], l.power[1], l.power[2], 0};
        memcpy(li + 8 * i, v, 32);
    }
    float *ma = reinterpret_cast<float *>(T + L.materials);
    for (size_t i = 0; i < materials.size(); i++) {
        const HostMaterial &m = materials[i];
        const float v[8] = {m.kd[0], m.kd[1], m.kd[2], 0, m.ks[0], m.ks[1], m.ks[2], 0};
        memcpy(ma + 8 * i, v, 32);
    }
    if (env_bytes) {
        memcpy(T + L.env, env.rgbe.empty() ? (const void *)env.f32.data() : (const void *)env.rgbe.data(), env_face_bytes);
        build_env_light(env, reinterpret_cast<float *>(T + L.env + env_sec.marg), reinterpret_cast<float *>(T + L.env + env_sec.cond), n_threads);
    }
    (void)f2u;
    if (packed_on_gpu) {
        std::string gpu_err;
        if (!gpu_arena_write(dev_arena_device, dev_arena, L.meshes, tail.data(), tail.size(), gpu_err)) {
            gpu_free(dev_arena_device, dev_arena);
            dev_arena = nullptr;
            err = gpu_err;
            return false;
        }
    } else {
        memcpy(A + L.meshes, tail.data(), tail.size());
    }
    layout = L;
    packed = true;
    build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("MIPT_TIMING")) fprintf(stderr, "[mipt] kd trees %.3f s (%d of %d meshes on the GPU), pack %.3f s (%s)\n", t_trees, meshes_built_on_gpu, (int)meshes.size(), build_seconds - t_trees, packed_on_gpu ? "on the GPU" : "host threads");
    return true;
}

// ------------------------------------------------------------------------------------------ export / import
namespace {
constexpr uint64_t kBlobMagic = 0x32424f4c4254504dull;  // "MPTBLOB2"
struct BlobHeader {
    uint64_t magic, struct_bytes, arena_bytes;
    ArenaLayout layout;
    int64_t total_tris, total_nodes, total_refs;
    int32_t n_materials, n_lights, n_planes, n_spheres, n_meshes;
    int32_t env_w, env_h, env_is_rgbe, cam_set;
    float cam[12];
    double build_seconds;
};
struct BlobMesh {
    int64_t n_tris, tri_base, node_base, ref_base, n_nodes, n_refs, degenerate_nodes, n_leaves, max_leaf;
    int32_t material, max_depth;
};
}  // namespace

size_t export_blob(const HostScene &s, uint8_t *out) {
    const size_t total = sizeof(BlobHeader) + s.meshes.size() * sizeof(BlobMesh) + s.arena.size();
    if (!out) return total;
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBlobMagic;
    h.struct_bytes = sizeof(BlobHeader) + ((uint64_t)sizeof(BlobMesh) << 32);
    h.arena_bytes = s.arena.size();
    h.layout = s.layout;
    h.total_tris = s.total_tris; h.total_nodes = s.total_nodes; h.total_refs = s.total_refs;
    h.n_materials = (int32_t)s.materials.size(); h.n_lights = (int32_t)s.lights.size(); h.n_planes = (int32_t)s.planes.size();
    h.n_spheres = (int32_t)s.spheres.size(); h.n_meshes = (int32_t)s.meshes.size();
    h.env_w = s.env.w; h.env_h = s.env.h; h.env_is_rgbe = s.env.rgbe.empty() ? 0 : 1; h.cam_set = s.cam_set ? 1 : 0;
    for (int k = 0; k < 3; k++) { h.cam[k] = s.cam_eye[k]; h.cam[3 + k] = s.cam_lookat[k]; h.cam[6 + k] = s.cam_right[k]; h.cam[9 + k] = s.cam_up[k]; }
    h.build_seconds = s.build_seconds;
    memcpy(out, &h, sizeof(h));
    uint8_t *p = out + sizeof(h);
    for (const HostMesh &m : s.meshes) {
        BlobMesh bm;
        memset(&bm, 0, sizeof(bm));
        bm.n_tris = m.n_tris; bm.tri_base = m.tri_base; bm.node_base = m.node_base; bm.ref_base = m.ref_base; bm.n_nodes = m.n_nodes;
        bm.n_refs = m.n_refs; bm.degenerate_nodes = m.degenerate_nodes; bm.n_leaves = m.n_leaves; bm.max_leaf = m.max_leaf;
        bm.material = m.material; bm.max_depth = m.max_depth;
        memcpy(p, &bm, sizeof(bm));
        p += sizeof(bm);
    }
    memcpy(p, s.arena.data(), s.arena.size());
    return total;
}

bool import_blob(HostScene &s, const uint8_t *blob, size_t size, std::string &err) {
    BlobHeader h;
    if (size < sizeof(h)) { err = "scene blob too small"; return false; }
    memcpy(&h, blob, sizeof(h));
    if (h.magic != kBlobMagic || h.struct_bytes != sizeof(BlobHeader) + ((uint64_t)sizeof(BlobMesh) << 32)) {
        err = "not a scene blob of this library version";
        return false;
    }
    if (h.n_meshes < 0 || h.n_materials < 0 || h.n_lights < 0 || h.n_planes < 0 || h.n_spheres < 0 || h.total_tris < 0 ||
        h.total_nodes < 0 || h.total_refs < 0 || h.env_w < 0 || h.env_h < 0) {
        err = "scene blob is inconsistent (negative count)";
        return false;
    }
    {   // total size without wrap-around: header + mesh table + arena
        const uint64_t rest = (uint64_t)size - sizeof(h);
        const uint64_t mesh_bytes = (uint64_t)h.n_meshes * sizeof(BlobMesh);
        if (mesh_bytes > rest || h.arena_bytes != rest - mesh_bytes || h.layout.total != h.arena_bytes) {
            err = "scene blob is truncated or inconsistent";
            return false;
        }
    }
    const ArenaLayout &L = h.layout;
    // every section must hold what the header says it holds (the tables are read below, the rest by the kernels)
    auto fits = [&](size_t off, uint64_t count, uint64_t stride) {
        return off <= h.arena_bytes && count <= (h.arena_bytes - off) / std::max<uint64_t>(stride, 1);
    };
    const uint64_t env_texels = (uint64_t)6 * (uint64_t)h.env_w * (uint64_t)h.env_h;
    const bool sections_ok =
        fits(L.nodes, (uint64_t)h.total_nodes, sizeof(KdNode)) && fits(L.boxes, (uint64_t)h.total_nodes, 32) &&
        fits(L.tight, (uint64_t)h.total_nodes, 32) && fits(L.tight_pairs, (uint64_t)h.total_nodes / 2 + 1, 64) &&
        fits(L.refs, (uint64_t)h.total_refs, 4) && fits(L.leaf_recs, (uint64_t)h.total_refs, 64) &&
        fits(L.leaf_spheres, (uint64_t)h.total_refs / 2 + 4, sizeof(SpherePair)) &&
        fits(L.tri_pos, (uint64_t)h.total_tris, 48) && fits(L.tri_nrm, (uint64_t)h.total_tris, 48) &&
        fits(L.meshes, (uint64_t)h.n_meshes, sizeof(DMesh)) && fits(L.mesh_order, (uint64_t)3 * (uint64_t)h.n_meshes, 4) &&
        fits(L.spheres, (uint64_t)h.n_spheres, 32) && fits(L.planes, (uint64_t)h.n_planes, 32) &&
        fits(L.lights, (uint64_t)h.n_lights, 32) && fits(L.materials, (uint64_t)h.n_materials, 32) &&
        fits(L.env, env_texels, h.env_is_rgbe ? 4 : 12) &&
        (env_texels == 0 || fits(L.env, (uint64_t)env_section(h.env_w, h.env_h, h.env_is_rgbe != 0).bytes, 1));
    if (!sections_ok) { err = "scene blob: a section does not fit in the arena"; return false; }
    if (h.total_nodes >= (int64_t)kIdxMask || h.total_refs >= (int64_t)INT32_MAX || h.total_tris >= (int64_t)(INT32_MAX / 4)) {
        err = "scene blob: counts beyond the index widths of the layout";
        return false;
    }
    {   // per-mesh ranges: inside the totals, in order, node bases even (sibling pairs)
        const uint8_t *q = blob + sizeof(h);
        int64_t tri_at = 0, node_at = 0, ref_at = 0;
        for (int32_t m = 0; m < h.n_meshes; m++, q += sizeof(BlobMesh)) {
            BlobMesh bm;
            memcpy(&bm, q, sizeof(bm));
            if (bm.n_tris <= 0 || bm.n_nodes <= 0 || bm.n_refs < 0 || bm.tri_base != tri_at || bm.node_base != node_at ||
                bm.ref_base != ref_at || (bm.node_base & 1) != 0 || bm.n_tris > h.total_tris - tri_at || bm.n_nodes > h.total_nodes - node_at ||
                bm.n_refs > h.total_refs - ref_at || bm.material < 0 || bm.material >= h.n_materials) {
                err = "scene blob: mesh table inconsistent with the arena";
                return false;
            }
            tri_at += bm.n_tris; node_at += bm.n_nodes; ref_at += bm.n_refs;
        }
        if (tri_at != h.total_tris || node_at != h.total_nodes || ref_at != h.total_refs) {
            err = "scene blob: mesh table does not add up to the totals";
            return false;
        }
    }
    s = HostScene();
    // the small tables are read back from the arena sections the packer wrote them to
    const uint8_t *p = blob + sizeof(h);
    s.meshes.resize((size_t)h.n_meshes);
    for (HostMesh &m : s.meshes) {
        BlobMesh bm;
        memcpy(&bm, p, sizeof(bm));
        p += sizeof(bm);
        m.n_tris = bm.n_tris; m.tri_base = bm.tri_base; m.node_base = bm.node_base; m.ref_base = bm.ref_base; m.n_nodes = bm.n_nodes;
        m.n_refs = bm.n_refs; m.degenerate_nodes = bm.degenerate_nodes; m.n_leaves = bm.n_leaves; m.max_leaf = bm.max_leaf;
        m.material = bm.material; m.max_depth = bm.max_depth;
    }
    s.arena.assign(p, p + h.arena_bytes);
    const uint8_t *A = s.arena.data();
    auto f = [&](size_t off, size_t i) { float v; memcpy(&v, A + off + 4 * i, 4); return v; };
    auto iv = [&](size_t off, size_t i) { int32_t v; memcpy(&v, A + off + 4 * i, 4); return v; };
    s.materials.resize((size_t)h.n_materials);
    for (size_t i = 0; i < s.materials.size(); i++)
        for (int k = 0; k < 3; k++) { s.materials[i].kd[k] = f(L.materials, 8 * i + k); s.materials[i].ks[k] = f(L.materials, 8 * i + 4 + k); }
    s.lights.resize((size_t)h.n_lights);
    for (size_t i = 0; i < s.lights.size(); i++)
        for (int k = 0; k < 3; k++) { s.lights[i].loc[k] = f(L.lights, 8 * i + k); s.lights[i].power[k] = f(L.lights, 8 * i + 4 + k); }
    s.planes.resize((size_t)h.n_planes);
    for (size_t i = 0; i < s.planes.size(); i++) {
        for (int k = 0; k < 3; k++) { s.planes[i].point[k] = f(L.planes, 8 * i + k); s.planes[i].normal[k] = f(L.planes, 8 * i + 4 + k); }
        s.planes[i].material = iv(L.planes, 8 * i + 3);
    }
    s.spheres.resize((size_t)h.n_spheres);
    for (size_t i = 0; i < s.spheres.size(); i++) {
        for (int k = 0; k < 3; k++) s.spheres[i].center[k] = f(L.spheres, 8 * i + k);
        s.spheres[i].radius = f(L.spheres, 8 * i + 3);
        s.spheres[i].material = iv(L.spheres, 8 * i + 4);
    }
    s.env.w = h.env_w;
    s.env.h = h.env_h;
    if (h.env_w > 0) {  // view() only asks which of the two forms is present; the texels stay in the arena
        if (h.env_is_rgbe) s.env.rgbe.assign(1, 0u);
        else s.env.f32.assign(1, 0.0f);
    }
    s.cam_set = h.cam_set != 0;
    for (int k = 0; k < 3; k++) { s.cam_eye[k] = h.cam[k]; s.cam_lookat[k] = h.cam[3 + k]; s.cam_right[k] = h.cam[6 + k]; s.cam_up[k] = h.cam[9 + k]; }
    s.layout = L;
    s.total_tris = h.total_tris; s.total_nodes = h.total_nodes; s.total_refs = h.total_refs;
    s.build_seconds = h.build_seconds;
    s.packed = true;
    s.imported = true;
    return true;
}

DScene HostScene::view(const void *base_v) const {
    const uint8_t *base = static_cast<const uint8_t *>(base_v);
    DScene d;
    memset(&d, 0, sizeof(d));
    d.nodes = reinterpret_cast<const KdNode *>(base + layout.nodes);
    d.boxes = reinterpret_cast<const f4 *>(base + layout.boxes);
    d.tight = reinterpret_cast<const f4 *>(base + layout.tight);
    d.tight_pairs = reinterpret_cast<const f4 *>(base + layout.tight_pairs);
    d.refs = reinterpret_cast<const int32_t *>(base + layout.refs);
    d.leaf_recs = reinterpret_cast<const f4 *>(base + layout.leaf_recs);
    d.leaf_spheres = reinterpret_cast<const SpherePair *>(base + layout.leaf_spheres);
    d.tri_pos = reinterpret_cast<const f4 *>(base + layout.tri_pos);
    d.tri_nrm = reinterpret_cast<const f4 *>(base + layout.tri_nrm);
    d.meshes = reinterpret_cast<const DMesh *>(base + layout.meshes);
    d.mesh_order = reinterpret_cast<const int32_t *>(base + layout.mesh_order);
    d.spheres = reinterpret_cast<const f4 *>(base + layout.spheres);
    d.planes = reinterpret_cast<const f4 *>(base + layout.planes);
    d.lights = reinterpret_cast<const f4 *>(base + layout.lights);
    d.materials = reinterpret_cast<const f4 *>(base + layout.materials);
    if (env.present()) {
        if (env.rgbe.empty()) d.env_f32 = reinterpret_cast<const float *>(base + layout.env);
        else d.env_rgbe = reinterpret_cast<const uint32_t *>(base + layout.env);
        const EnvSection es = env_section(env.w, env.h, !env.rgbe.empty());
        d.env_marg = reinterpret_cast<const float *>(base + layout.env + es.marg);
        d.env_cond = reinterpret_cast<const float *>(base + layout.env + es.cond);
    }
    d.n_meshes = (int32_t)meshes.size();
    d.n_spheres = (int32_t)spheres.size();
    d.n_planes = (int32_t)planes.size();
    d.n_lights = (int32_t)lights.size();
    d.n_materials = (int32_t)materials.size();
    d.total_tris = (int32_t)total_tris;
    d.env_w = env.w;
    d.env_h = env.h;
    for (int k = 0; k < 3; k++) {
        d.cam_eye[k] = cam_eye[k]; d.cam_lookat[k] = cam_lookat[k]; d.cam_right[k] = cam_right[k]; d.cam_up[k] = cam_up[k];
    }
    return d;
}

}  // namespace mipt
