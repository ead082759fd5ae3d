// assets.cpp -- asset readers of libmipt's host side: Wavefront OBJ meshes and Radiance RGBE cube faces.
//
// Reference behaviour restated (paths relative to /root/reference/cuda-path-tracer/):
//   readObjIntoMesh            shapes/mesh.cu:232-317   `v` / `vn` / `f p//n | p/t/n`, triangles only, normals mandatory,
//                                                       position -> vec4(p,1) * SRT, normal -> vec4(n,0) * SRT.inverse().transpose()
//   SRTmtx / mat4 / vec4       math/math.cuh:50-158, math/math.cu:3-39  (row vectors; vec4(vec3,w) swaps y/z: Q2)
//   HDRLoader::load            HDRLoader/HDRLoader.cu:29-97   header, `-Y h +X w`, rows in file order
//   decrunch / oldDecrunch     HDRLoader/HDRLoader.cu:118-191 the two scanline encodings
//
// What is different on purpose: the OBJ file is memory-mapped and parsed by all builder threads at once (a 1.3 M-face file
// is ~100 MB of text); the RGBE texels stay packed (decoded in-kernel, shade.h::env_lookup); and every malformed input is
// an ERROR here -- the reference prints nothing and keeps a half-decoded (black) face, or indexes out of bounds.
#include "scene_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace mipt {
namespace {

// ------------------------------------------------------------------------------------------ read-only file mapping
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    ~MappedFile() {
        if (data && size) munmap(const_cast<char *>(data), size);
    }
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        if (size) {
            void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) { ::close(fd); size = 0; return false; }
            madvise(p, size, MADV_SEQUENTIAL);
            data = static_cast<const char *>(p);
        }
        ::close(fd);
        return true;
    }
};

// ------------------------------------------------------------------------------------------ SRT (math.cuh:127-158)
struct Row4 { float c[4]; };
struct Mat44 { Row4 r[4]; };

Row4 row_times(const Row4 &v, const Mat44 &m) {  // row-vector * matrix: v0*row0 + v1*row1 + v2*row2 + v3*row3
    Row4 acc;
    for (int j = 0; j < 4; j++) acc.c[j] = m.r[0].c[j] * v.c[0];
    for (int i = 1; i < 4; i++)
        for (int j = 0; j < 4; j++) acc.c[j] = acc.c[j] + m.r[i].c[j] * v.c[i];
    return acc;
}
Mat44 mat_mul(const Mat44 &a, const Mat44 &b) {
    Mat44 o;
    for (int i = 0; i < 4; i++) o.r[i] = row_times(a.r[i], b);
    return o;
}
Mat44 identity() {
    Mat44 m;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m.r[i].c[j] = (i == j) ? 1.0f : 0.0f;
    return m;
}
Mat44 rotation(float angle, f3 w) {
    const float c = cosf(angle), s = sinf(angle);
    w = norm3(w);
    Mat44 m = identity();
    m.r[0] = Row4{{c * (1.0f - w.x * w.x) + w.x * w.x, w.x * w.y * (1.0f - c) + w.z * s, w.x * w.z * (1.0f - c) - w.y * s, 0.0f}};
    m.r[1] = Row4{{w.x * w.y * (1.0f - c) - w.z * s, c * (1.0f - w.y * w.y) + w.y * w.y, w.y * w.z * (1.0f - c) + w.x * s, 0.0f}};
    m.r[2] = Row4{{w.x * w.z * (1.0f - c) + w.y * s, w.y * w.z * (1.0f - c) - w.x * s, c * (1.0f - w.z * w.z) + w.z * w.z, 0.0f}};
    return m;
}
Mat44 srt(const float scale[3], const float rot[3], const float tr[3]) {
    Mat44 S = identity(), T = identity();
    S.r[0].c[0] = scale[0]; S.r[1].c[1] = scale[1]; S.r[2].c[2] = scale[2];
    T.r[3].c[0] = tr[0]; T.r[3].c[1] = tr[1]; T.r[3].c[2] = tr[2];
    Mat44 m = mat_mul(S, rotation(rot[0], mk3(1, 0, 0)));
    m = mat_mul(m, rotation(rot[1], mk3(0, 1, 0)));
    m = mat_mul(m, rotation(rot[2], mk3(0, 0, 1)));
    return mat_mul(m, T);
}
// inverse().transpose() exactly as math.cu:3-39 + math.cuh:104-112 define them
Mat44 normal_matrix(const Mat44 &m) {
    const f3 a = mk3(m.r[0].c[0], m.r[0].c[1], m.r[0].c[2]), b = mk3(m.r[1].c[0], m.r[1].c[1], m.r[1].c[2]),
             c = mk3(m.r[2].c[0], m.r[2].c[1], m.r[2].c[2]), t = mk3(m.r[3].c[0], m.r[3].c[1], m.r[3].c[2]);
    const float det = dot3(cross3(a, b), c);
    Mat44 inv = identity();
    if (!(fabsf(det) < kEps)) {
        const float id = 1.0f / det;
        const f3 r0 = cross3(b, c) * id, r1 = cross3(c, a) * id, r2 = cross3(a, b) * id;
        inv.r[0] = Row4{{r0.x, r0.y, r0.z, 0.0f}};
        inv.r[1] = Row4{{r1.x, r1.y, r1.z, 0.0f}};
        inv.r[2] = Row4{{r2.x, r2.y, r2.z, 0.0f}};
        inv.r[3] = Row4{{-dot3(r0, t), -dot3(r1, t), -dot3(r2, t), 1.0f}};
    }
    Mat44 tr;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) tr.r[i].c[j] = inv.r[j].c[i];
    return tr;
}

// ------------------------------------------------------------------------------------------ OBJ text
inline bool blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\r'; }

// What one thread found in its slice of the file (whole lines).  Face corners keep the file's 1-based indices.
struct ObjSlice {
    std::vector<float> v, vn;
    std::vector<int64_t> corner_p, corner_n;
    std::string err;
    int64_t err_line_offset = 0;
};

// One line [p, eol): never reads at or past eol (strtof / strtoll stop at the '\n' that ends every line but the last of
// the file, which the caller hands over as a NUL-terminated copy).
bool parse_three(const char *p, const char *eol, std::vector<float> &dst) {
    for (int k = 0; k < 3; k++) {
        while (p < eol && blank(*p)) p++;
        if (p >= eol || isspace((unsigned char)*p)) return false;  // strtof would skip it and read into the next line
        char *end;
        const float f = strtof(p, &end);
        if (end == p || end > eol) return false;
        dst.push_back(f);
        p = end;
    }
    return true;
}

bool parse_face(const char *p, const char *eol, ObjSlice &out) {
    int corners = 0;
    for (;;) {
        while (p < eol && blank(*p)) p++;
        if (p >= eol) break;
        int64_t slot[3] = {0, 0, 0};
        bool have[3] = {false, false, false};
        for (int k = 0; k < 3; k++) {  // p, p/t, p//n, p/t/n
            if (p < eol && !isspace((unsigned char)*p)) {  // (strtoll skips white space, the line's end included)
                char *end;
                const long long v = strtoll(p, &end, 10);
                if (end != p && end <= eol) { slot[k] = v; have[k] = true; p = end; }
            }
            if (p < eol && *p == '/') p++; else break;
        }
        if (!have[0]) { out.err = "OBJ face corner is not an index"; return false; }
        if (!have[2]) { out.err = "OBJ face without normal index (normals are mandatory, mesh.cu:277-279,303)"; return false; }
        if (p < eol && !blank(*p)) { out.err = "OBJ face corner has trailing characters"; return false; }
        out.corner_p.push_back(slot[0]);
        out.corner_n.push_back(slot[2]);
        corners++;
    }
    if (corners != 3) { out.err = "OBJ faces must be triangles (mesh.cu:296-301 groups corners by three)"; return false; }
    return true;
}

void parse_slice(const char *begin, const char *end, ObjSlice &out) {
    const char *p = begin;
    while (p < end) {
        const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        const char *q = p;
        while (q < eol && blank(*q)) q++;
        bool ok = true;
        if (eol - q >= 2 && q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
            ok = parse_three(q + 1, eol, out.v);
            if (!ok) out.err = "OBJ `v` line without three numbers";
        } else if (eol - q >= 3 && q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) {
            ok = parse_three(q + 2, eol, out.vn);
            if (!ok) out.err = "OBJ `vn` line without three numbers";
        } else if (eol - q >= 2 && q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
            ok = parse_face(q + 1, eol, out);
        }  // every other prefix (vt, o, g, s, usemtl, mtllib, #) is skipped like mesh.cu:259-293 does
        if (!ok) { out.err_line_offset = p - begin; return; }
        p = eol + 1;
    }
}

}  // namespace

bool load_obj(const char *path, const float position[3], const float rotate[3], const float scale[3], HostMesh &out,
              std::string &err) {
    MappedFile file;
    if (!file.open(path)) { err = std::string("cannot open OBJ file ") + path; return false; }
    if (file.size == 0) { err = "OBJ has no faces"; return false; }
    const char *const b = file.data, *const e = file.data + file.size;
    // the last line may end at the end of the mapping without '\n': it is parsed from a terminated copy
    const char *tail = e;
    while (tail > b && tail[-1] != '\n') tail--;
    const std::string last_line(tail, e);

    unsigned hw = std::thread::hardware_concurrency();
    if (const char *env = getenv("MIPT_BUILD_THREADS")) hw = (unsigned)std::max(1, atoi(env));
    const size_t body = (size_t)(tail - b);
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(std::min(16u, std::max(1u, hw)), body / (1u << 20)));
    std::vector<const char *> cut(nt + 1);
    cut[0] = b;
    cut[nt] = tail;
    for (int k = 1; k < nt; k++) {  // slice boundaries moved forward to the next line start
        const char *p = b + body / nt * k;
        if (p < cut[k - 1]) p = cut[k - 1];
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(tail - p)));
        cut[k] = nl ? nl + 1 : tail;
    }
    std::vector<ObjSlice> slice(nt + 1);
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < nt; k++) pool.emplace_back([&, k] { parse_slice(cut[k], cut[k + 1], slice[k]); });
        parse_slice(cut[0], cut[1], slice[0]);
        for (std::thread &t : pool) t.join();
        parse_slice(last_line.data(), last_line.data() + last_line.size(), slice[nt]);
    }
    size_t nv = 0, nn = 0, nc = 0;
    for (const ObjSlice &s : slice) {
        if (!s.err.empty()) { err = s.err; return false; }
        nv += s.v.size(); nn += s.vn.size(); nc += s.corner_p.size();
    }
    if (nc == 0) { err = "OBJ has no faces"; return false; }
    std::vector<float> P(nv), N(nn);
    std::vector<size_t> corner_at(slice.size());
    {
        size_t av = 0, an = 0, ac = 0;
        for (size_t k = 0; k < slice.size(); k++) {
            if (!slice[k].v.empty()) memcpy(&P[av], slice[k].v.data(), slice[k].v.size() * sizeof(float));
            if (!slice[k].vn.empty()) memcpy(&N[an], slice[k].vn.data(), slice[k].vn.size() * sizeof(float));
            av += slice[k].v.size(); an += slice[k].vn.size();
            corner_at[k] = ac;
            ac += slice[k].corner_p.size();
        }
    }
    const int64_t np = (int64_t)nv / 3, nnrm = (int64_t)nn / 3;
    const Mat44 M = srt(scale, rotate, position), NM = normal_matrix(M);
    out.n_tris = (int64_t)nc / 3;
    out.pos.resize(nc * 3);
    out.nrm.resize(nc * 3);
    std::vector<char> bad(slice.size(), 0);
    auto bake = [&](size_t k) {  // mesh.cu:296-315 for the corners of slice k
        const ObjSlice &s = slice[k];
        for (size_t c = 0; c < s.corner_p.size(); c++) {
            const int64_t ip = s.corner_p[c] - 1, in = s.corner_n[c] - 1;  // "we index from 0" (:275,278)
            if (ip < 0 || ip >= np || in < 0 || in >= nnrm) { bad[k] = 1; return; }
            const float *v = &P[3 * (size_t)ip], *n = &N[3 * (size_t)in];
            const Row4 pv = row_times(Row4{{v[0], v[2], v[1], 1.0f}}, M);   // y/z swapped (math.cuh:55, Q2)
            const Row4 nv4 = row_times(Row4{{n[0], n[2], n[1], 0.0f}}, NM);
            const f3 nu = norm3(mk3(nv4.c[0], nv4.c[1], nv4.c[2]));
            const size_t o = 3 * (corner_at[k] + c);
            out.pos[o] = pv.c[0]; out.pos[o + 1] = pv.c[1]; out.pos[o + 2] = pv.c[2];
            out.nrm[o] = nu.x; out.nrm[o + 1] = nu.y; out.nrm[o + 2] = nu.z;
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < slice.size(); k++) pool.emplace_back(bake, k);
        bake(0);
        for (std::thread &t : pool) t.join();
    }
    for (char f : bad)
        if (f) { err = "OBJ index out of range (negative / relative indices are not supported, as in mesh.cu:275)"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------ Radiance RGBE
namespace {

// Reader over the bytes of one .hdr file.  Every `take` past the end, every run that leaves its scanline and every
// marker in an impossible place is a hard error (`fail`), reported with the byte offset.
class RadianceReader {
public:
    RadianceReader(const unsigned char *p, size_t n) : p_(p), n_(n) {}
    const std::string &error() const { return err_; }

    // "#?RADIANCE" + one byte, header lines up to the empty line, then "-Y h +X w" (HDRLoader.cu:39-72)
    bool header(int &w, int &h) {
        static const char magic[] = "#?RADIANCE";
        if (n_ < 11 || memcmp(p_, magic, 10) != 0) return fail("not a Radiance file (no #?RADIANCE signature)");
        at_ = 11;
        for (int prev = 0;;) {
            if (at_ >= n_) return fail("truncated header");
            const int ch = p_[at_++];
            if (ch == '\n' && prev == '\n') break;
            prev = ch;
        }
        char line[64];
        size_t k = 0;
        for (;;) {
            if (at_ >= n_) return fail("truncated resolution line");
            const char ch = (char)p_[at_++];
            if (ch == '\n') break;
            if (k + 1 >= sizeof(line)) return fail("resolution line too long");
            line[k++] = ch;
        }
        line[k] = 0;
        long hh = 0, ww = 0;
        if (sscanf(line, "-Y %ld +X %ld", &hh, &ww) != 2) return fail("unsupported orientation (only -Y h +X w, HDRLoader.cu:68)");
        if (ww <= 0 || hh <= 0 || ww > 0x7fff || hh > 0x7fff) return fail("resolution out of range (1..32767)");
        // every scanline takes at least four bytes of the file (a flat texel, or the run-length prefix); its WIDTH is not
        // bounded by the file's size: three (1,1,1,n) markers behind one texel repeat it up to 255 + 255 * 256 + ... times
        if ((size_t)hh > (n_ - at_) / 4) return fail("more scanlines than the file can hold");
        w = (int)ww;
        h = (int)hh;
        return true;
    }

    // One scanline of `w` texels, packed R | G<<8 | B<<16 | E<<24.
    bool scanline(uint32_t *dst, int w) {
        if (w >= 8 && at_ + 4 <= n_ && p_[at_] == 2 && p_[at_ + 1] == 2 && !(p_[at_ + 2] & 128)) {
            // adaptive run-length encoding: (2, 2, w_hi, w_lo), then the four components one after another (:131-157)
            const int coded = (p_[at_ + 2] << 8) | p_[at_ + 3];
            if (coded != w) return fail("run-length scanline of a different width");
            at_ += 4;
            for (int x = 0; x < w; x++) dst[x] = 0u;
            for (int comp = 0; comp < 4; comp++) {
                const int shift = 8 * comp;
                for (int x = 0; x < w;) {
                    if (at_ >= n_) return fail("truncated run-length data");
                    int count = p_[at_++];
                    if (count > 128) {  // a run of one value
                        count -= 128;
                        if (x + count > w) return fail("run leaves the scanline");
                        if (at_ >= n_) return fail("truncated run-length data");
                        const uint32_t v = (uint32_t)p_[at_++] << shift;
                        for (int r = 0; r < count; r++) dst[x++] |= v;
                    } else {            // literal bytes
                        if (count == 0) return fail("zero-length literal block");
                        if (x + count > w) return fail("literal block leaves the scanline");
                        if (at_ + (size_t)count > n_) return fail("truncated run-length data");
                        for (int r = 0; r < count; r++) dst[x++] |= (uint32_t)p_[at_++] << shift;
                    }
                }
            }
            return true;
        }
        // flat texels with (1,1,1,n) repeat markers; consecutive markers extend the count by 8 bits each (:161-191)
        int x = 0, marker_shift = 0;
        while (x < w) {
            if (at_ + 4 > n_) return fail("truncated scanline");
            const unsigned char *t = p_ + at_;
            at_ += 4;
            if (t[0] == 1 && t[1] == 1 && t[2] == 1) {
                if (x == 0) return fail("repeat marker before the first texel of a scanline");
                const long count = (long)t[3] << marker_shift;
                if (count > w - x) return fail("repeat leaves the scanline");
                for (long r = 0; r < count; r++, x++) dst[x] = dst[x - 1];
                marker_shift += 8;
                if (marker_shift > 24) return fail("too many consecutive repeat markers");
            } else {
                dst[x++] = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                marker_shift = 0;
            }
        }
        return true;
    }

private:
    bool fail(const char *what) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s (byte %zu)", what, at_);
        err_ = buf;
        return false;
    }
    const unsigned char *p_;
    size_t n_, at_ = 0;
    std::string err_;
};

}  // namespace

bool load_hdr_rgbe(const char *path, int &w, int &h, std::vector<uint32_t> &texels, std::string &err) {
    MappedFile file;
    if (!file.open(path)) { err = std::string("cannot open HDR file ") + path; return false; }
    RadianceReader rd(reinterpret_cast<const unsigned char *>(file.data), file.size);
    if (!rd.header(w, h)) { err = std::string(path) + ": " + rd.error(); return false; }
    texels.clear();
    for (int y = 0; y < h; y++) {  // rows stay in file order: top row first (HDRLoader.cu:86-91)
        texels.resize((size_t)(y + 1) * w, 0u);  // grown row by row: a damaged header cannot ask for memory the rows never fill
        if (!rd.scanline(texels.data() + (size_t)y * w, w)) {
            err = std::string(path) + ": row " + std::to_string(y) + ": " + rd.error();
            return false;
        }
    }
    return true;
}

}  // namespace mipt
