// shade.h -- device functions of the shading stages: RNG, camera, surface normals, light, samplers,
// cube-map lookup.  Arithmetic follows the reference line by line in fp32 (no contraction):
//   Camera::getRay            camera.cuh:17-20
//   Triangle::intersect tail  mesh.cuh:90-110 (barycentric vertex normal, flipped toward the ray)
//   Sphere/Plane normals      sphere.cuh:33-35, plane.cuh:26-27
//   Light::radianceAt         light.cuh:25-29;  Light::randomSampleRay light.cuh:30-39 (Q12)
//   SampleDiffuse/Mirror      samplers.cuh:8-23, 46-49
//   EnvMap::getPixelColor     envMap.cuh:50-89 (Q19) + convertComponent HDRLoader.cu:99-104
//   Rand::random              random.cuh:7-9 -> Philox4x32-10 keyed on (seed, pixel, sample); uniforms in (0,1]
#pragma once
#include "scene_types.h"
#include "traverse.h"

namespace mipt {

constexpr uint32_t kRngKey1 = 0x6d697074u;  // 'mipt'

PT_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// 32 x 32 -> 64-bit product, both halves from ONE v_mad_u64_u32 on the device (the compiler splits the product into
// v_mul_hi_u32 + v_mul_lo_u32: 40 multiplies per Philox block instead of 20; scripts/micro/mulbench.hip)
PT_HD void mul_wide(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
    hi = (uint32_t)(r >> 32);
    lo = (uint32_t)r;
#else
    const uint64_t r = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(r >> 32);
    lo = (uint32_t)r;
#endif
}
PT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0, l0, h1, l1;
        mul_wide(0xD2511F53u, c0, h0, l0);
        mul_wide(0xCD9E8D57u, c2, h1, l1);
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Sequential draw stream d = 0,1,2,... of one (seed, pixel, sample); block d>>2 is regenerated on demand,
// so a path can resume its stream in a later kernel from the stored draw counter alone.
struct Rng {
    uint32_t seed, pixel, sample, draw, blk;
    uint32_t v0, v1, v2, v3;
    PT_HD void init(uint32_t s, uint32_t p, uint32_t smp, uint32_t d) {
        seed = s; pixel = p; sample = smp; draw = d; blk = 0xffffffffu;
        v0 = v1 = v2 = v3 = 0;
    }
    PT_HD float u() {
        const uint32_t b = draw >> 2;
        if (b != blk) {
            uint32_t o[4];
            philox4x32_10(b, sample, pixel, 0u, seed, kRngKey1, o);
            v0 = o[0]; v1 = o[1]; v2 = o[2]; v3 = o[3];
            blk = b;
        }
        const uint32_t k = draw & 3u;
        const uint32_t x = k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3));
        draw++;
        return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
    }
};

PT_HD f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
PT_HD f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }
PT_HD f4 mk4(f3 v, float w) { f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = w; return r; }

// Camera::getRay; returns the un-normalised direction (the Ray constructor normalises)
PT_HD f3 camera_dir(const DScene &sc, float X, float Y, int W, int H) {
    const f3 eye = ld3(sc.cam_eye), lookat = ld3(sc.cam_lookat), right = ld3(sc.cam_right), up = ld3(sc.cam_up);
    return lookat + right * (2.0f * X / (float)W - 1.0f) + up * (2.0f * Y / (float)H - 1.0f) - eye;
}

struct Surface {
    f3 pos, nrm;
    f3 kd, ks;
};

// Resolve a HitRec into position / normal / material (what the reference's Hit carries, hit.cuh:9-15).
PT_HD Surface surface_at(const DScene &sc, f3 o, f3 d, float t, int prim) {
    Surface s;
    s.pos = ray_at(o, d, t);
    int mat;
    if (prim < sc.total_tris) {
        const f4 pa = sc.tri_pos[3 * (size_t)prim], pb = sc.tri_pos[3 * (size_t)prim + 1], pc = sc.tri_pos[3 * (size_t)prim + 2];
        const f4 na = sc.tri_nrm[3 * (size_t)prim], nb = sc.tri_nrm[3 * (size_t)prim + 1], nc = sc.tri_nrm[3 * (size_t)prim + 2];
        const f3 va = xyz(pa) - xyz(pb), vc = xyz(pc) - xyz(pb), vh = s.pos - xyz(pb);
        const float daa = dot3(va, va), dac = dot3(va, vc), dcc = dot3(vc, vc);
        const float denom = daa * dcc - dac * dac;
        const float dha = dot3(vh, va), dhc = dot3(vh, vc);
        const float b0 = (dcc * dha - dac * dhc) / denom;
        const float b1 = (daa * dhc - dac * dha) / denom;
        const float b2 = 1.0f - b0 - b1;
        s.nrm = norm3(xyz(na) * b0 + xyz(nb) * b2 + xyz(nc) * b1);
        int m = 0;
        for (int k = 1; k < sc.n_meshes; k++)
            if (prim >= ldk_i32(&sc.meshes[k].tri_base)) m = k;
        mat = sc.meshes[m].material;
    } else if (prim < sc.total_tris + sc.n_spheres) {
        const int i = prim - sc.total_tris;
        const f4 c = sc.spheres[2 * i], e = sc.spheres[2 * i + 1];
        s.nrm = (s.pos - xyz(c)) / c.w;
        mat = (int)as_u32(e.x);
    } else {
        const int i = prim - sc.total_tris - sc.n_spheres;
        const f4 p = sc.planes[2 * i], n = sc.planes[2 * i + 1];
        s.nrm = xyz(n);
        mat = (int)as_u32(p.w);
    }
    if (dot3(s.nrm, d) > 0.0f) s.nrm = s.nrm * (-1.0f);
    s.kd = xyz(sc.materials[2 * mat]);
    s.ks = xyz(sc.materials[2 * mat + 1]);
    return s;
}

PT_HD f3 radiance_at(const DScene &sc, int light, f3 p) {
    const f3 loc = xyz(ldk4(sc.lights, 2 * (size_t)light)), power = xyz(ldk4(sc.lights, 2 * (size_t)light + 1));
    float d2 = dot3(loc - p, loc - p);
    if (d2 < kEps) d2 = kEps;
    return power / d2 / 4.0f / kPi;
}

PT_HD f3 light_sample_dir(Rng &rng) {
    const f3 X = mk3(0, 0, 1), Y = mk3(0, 1, 0), Z = mk3(1, 0, 0);
    const float alpha = rng.u() * 2.0f * kPi;
    const float beta = rng.u() * 2.0f * kPi;
    return norm3((X * det_cosf(alpha) + Z * det_sinf(alpha)) * det_sinf(beta) + Y * det_cosf(beta));
}

PT_HD float sample_diffuse(f3 N, f3 &out, Rng &rng) {
    f3 T = cross3(N, mk3(1.0f, 0.0f, 0.0f));
    if (len3(T) < kEps) T = cross3(N, mk3(0.0f, 0.0f, 1.0f));
    T = norm3(T);
    const f3 B = cross3(N, T);
    float x, y;
    do {
        x = 2.0f * rng.u() - 1.0f;
        y = 2.0f * rng.u() - 1.0f;
    } while (x * x + y * y > 1.0f);
    const float z = sqrtf(1.0f - x * x - y * y);
    out = N * z + T * x + B * y;
    return z / kPi;
}

PT_HD float sample_mirror(f3 N, f3 in, f3 &out) {
    out = in - N * dot3(N, in) * 2.0f;
    return 1.0f;
}

// One texel of the cube map, decoded as HDRLoader's convertComponent does (HDRLoader.cu:99-104)
PT_HD f3 env_texel(const DScene &sc, size_t texel) {
    if (sc.env_rgbe) {
        const uint32_t px = sc.env_rgbe[texel];
        const int expo = (int)(px >> 24) - 128;
        const float scale = ldexpf(1.0f, expo);  // (float)pow(2, expo), exact
        return mk3(((int)(px & 255u) / 256.0f) * scale, ((int)((px >> 8) & 255u) / 256.0f) * scale,
                   ((int)((px >> 16) & 255u) / 256.0f) * scale);
    }
    const float *c = sc.env_f32 + texel * 3;
    return mk3(c[0], c[1], c[2]);
}

// ---- The cube map as a LIGHT (mipt_render_config.use_envmap == 2).  No counterpart in the reference, which only looks
// radiance up; a build-defined extension like the miss term.  A texel stands for the direction through its
// centre and is drawn with probability proportional to average(rgb) x solid angle from two CDF tables in the arena (packed
// by scene.cpp build_env_light: env_marg over all 6 h rows, env_cond inside a row).
// Direction (not normalised) and solid angle of texel (face, row, col): the inverse of env_lookup's index formulas, face by
// face -- env_lookup(dir) lands in this texel again (tests/test_env_light.py).
PT_HD void env_texel_geometry(int face, int row, int col, int w, int h, f3 &dir, float &solid_angle) {
    const float fu = ((float)col + 0.5f) / (float)w, fv = ((float)row + 0.5f) / (float)h;
    const float a = 2.0f * fu - 1.0f, b = 2.0f * fv - 1.0f;
    const float len2 = 1.0f + a * a + b * b;
    solid_angle = (4.0f / ((float)w * (float)h)) / (len2 * sqrtf(len2));
    switch (face) {
        case 0: dir = mk3(1.0f, -b, -a); break;
        case 3: dir = mk3(-1.0f, -b, a); break;
        case 1: dir = mk3(a, 1.0f, b); break;
        case 4: dir = mk3(a, -1.0f, -b); break;
        case 2: dir = mk3(a, -b, 1.0f); break;
        default: dir = mk3(-a, -b, -1.0f); break;
    }
}
// first index whose CDF entry is >= u (u in (0, 1]; the last entry is 1)
PT_HD int cdf_search(const float *cdf, int n, float u) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] < u) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
PT_HD bool env_light_usable(const DScene &sc) { return sc.env_marg != nullptr && sc.env_marg[6 * sc.env_h] != 0.0f; }
// One importance sample: direction (not normalised), solid angle, probability and radiance of the texel (u1, u2) select.
PT_HD bool env_light_sample(const DScene &sc, float u1, float u2, f3 &dir, float &solid_angle, float &pmf, f3 &radiance) {
    const int w = sc.env_w, h = sc.env_h;
    const int R = cdf_search(sc.env_marg, 6 * h, u1);
    const float *cond = sc.env_cond + (size_t)R * w;
    const int c = cdf_search(cond, w, u2);
    const float p_row = sc.env_marg[R] - (R > 0 ? sc.env_marg[R - 1] : 0.0f);
    const float p_col = cond[c] - (c > 0 ? cond[c - 1] : 0.0f);
    pmf = p_row * p_col;
    const int face = R / h, row = R - face * h;
    env_texel_geometry(face, row, c, w, h, dir, solid_angle);
    radiance = env_texel(sc, (size_t)face * w * h + (size_t)row * w + c);
    return pmf > 0.0f;
}

// EnvMap::getPixelColor: nearest texel, flattened index clamped only above (Q19); faces px,py,pz,nx,ny,nz
PT_HD f3 env_lookup(const DScene &sc, f3 d) {
    const int W = sc.env_w, H = sc.env_h;
    int face, pos;
    if ((fabsf(d.x) >= fabsf(d.y)) && (fabsf(d.x) >= fabsf(d.z))) {
        face = d.x > 0.0f ? 0 : 3;
        pos = ((int)((1.0f - ((d.y / fabsf(d.x) + 1.0f) / 2.0f)) * H)) * 3 * W + ((int)((1.0f - ((d.z / d.x + 1.0f) / 2.0f)) * W)) * 3;
    } else if ((fabsf(d.y) >= fabsf(d.x)) && (fabsf(d.y) >= fabsf(d.z))) {
        face = d.y > 0.0f ? 1 : 4;
        pos = ((int)((d.z / d.y + 1.0f) / 2.0f * H)) * 3 * W + ((int)((d.x / fabsf(d.y) + 1.0f) / 2.0f * W)) * 3;
    } else {
        face = d.z > 0.0f ? 2 : 5;
        pos = ((int)((1.0f - ((d.y / fabsf(d.z) + 1.0f) / 2.0f)) * H)) * 3 * W + ((int)((d.x / d.z + 1.0f) / 2.0f * W)) * 3;
    }
    if (pos > H * W * 3 - 3) pos = H * W * 3 - 3;
    if (pos < 0) pos = 0;
    return env_texel(sc, (size_t)face * W * H + (size_t)(pos / 3));
}

}  // namespace mipt
