"""Which ordering of BOUNCE rays (scattered origins, cosine-hemisphere directions) helps the extend kernel?
Third-vertex camera rays of the C2 scene traced as generated (path order) vs sorted by several keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene
desc = G.scene_icosphere_room(6)
s = desc.apply(Scene()); s.commit((0,))
W, H = 1920, 1080
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ro, rd = G.camera_rays(desc.camera, W, H, W * H)
idx = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
ro, rd = np.tile(ro[idx], (SPP, 1)), np.tile(rd[idx], (SPP, 1))  # SPP sample slots, tile order inside each (as in a pass)
c = np.array([0, -1, -2], np.float32)
rng = np.random.Generator(np.random.Philox(key=3))


def hit_points(ro, rd):
    prim, t, _ = s.trace_closest(ro, rd)
    d = rd / (np.linalg.norm(rd, axis=1, keepdims=True) + 1e-5)
    P = ro + d * t[:, None]
    n = np.where((prim < 81920)[:, None], (P - c) / 2.5, 0)
    ax = np.argmax(np.abs(P), axis=1)
    nw = np.zeros_like(P); nw[np.arange(len(P)), ax] = -np.sign(P[np.arange(len(P)), ax])
    n = np.where((prim >= 81920)[:, None], nw, n).astype(np.float32)
    return P.astype(np.float32), n, prim >= 0


def bounce(P, n, ok):
    v = rng.standard_normal(P.shape).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (P + n * 1e-5)[ok], (n + v * 0.999)[ok]


depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
o, d = ro, rd
for _ in range(depth):
    P, n, ok = hit_points(o, d)
    o, d = bounce(P, n, ok)
d = d / np.linalg.norm(d, axis=1, keepdims=True)
lo, hi = c - 2.5, c + 2.5
with np.errstate(divide="ignore", invalid="ignore"):
    t0, t1 = (lo - o) / d, (hi - o) / d
tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
keep = (tn <= tf) & (tf > 0)
o, d = np.ascontiguousarray(o[keep]), np.ascontiguousarray(d[keep]); m = len(d)
print("depth", depth, "bounce rays through the box:", m, flush=True)


def dir_cell(d, n):
    a = d / np.abs(d).sum(axis=1, keepdims=True)
    u, w = a[:, 0], a[:, 2]
    neg = a[:, 1] < 0
    u2 = np.where(neg, (1 - np.abs(w)) * np.sign(u), u); w2 = np.where(neg, (1 - np.abs(a[:, 0])) * np.sign(w), w)
    iu = np.clip(((u2 * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1); iw = np.clip(((w2 * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    return iu * n + iw


def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def morton(p, bits):
    q = np.clip(((p + 5.0) / 10.0 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
orders = [("as generated", np.arange(m)),
          ("dir 256x256", np.argsort(dir_cell(d, 256), kind="stable")),
          ("dir 16x16", np.argsort(dir_cell(d, 16), kind="stable")),
          ("origin morton 5b", np.argsort(morton(o, 5), kind="stable")),
          ("origin morton 8b", np.argsort(morton(o, 8), kind="stable")),
          ("octant, morton 5b", np.argsort(octant * (1 << 15) + morton(o, 5), kind="stable")),
          ("morton 4b, dir 8x8", np.argsort(morton(o, 4) * 64 + dir_cell(d, 8), kind="stable")),
          ("dir 8x8, morton 5b", np.argsort(dir_cell(d, 8) * (1 << 15) + morton(o, 5), kind="stable")),
          ("dir 16x16, morton 6b", np.argsort(dir_cell(d, 16) * (1 << 18) + morton(o, 6), kind="stable")),
          ("morton 5b, dir 8x8", np.argsort(morton(o, 5) * 64 + dir_cell(d, 8), kind="stable")),
          ("morton 6b, dir 8x8", np.argsort(morton(o, 6) * 64 + dir_cell(d, 8), kind="stable")),
          ("morton 7b, dir 8x8", np.argsort(morton(o, 7) * 64 + dir_cell(d, 8), kind="stable")),
          ("morton 5b, dir 16x16", np.argsort(morton(o, 5) * 256 + dir_cell(d, 16), kind="stable")),
          ("morton 6b, dir 16x16", np.argsort(morton(o, 6) * 256 + dir_cell(d, 16), kind="stable")),
          ("morton 4b, dir 32x32", np.argsort(morton(o, 4) * 1024 + dir_cell(d, 32), kind="stable")),
          ("dir 8x8, morton 7b", np.argsort(dir_cell(d, 8) * (1 << 21) + morton(o, 7), kind="stable")),
          ("shuffled", rng.permutation(m))]
d_p = torch.empty(m, dtype=torch.int32, device="cuda"); d_t = torch.empty(m, dtype=torch.float32, device="cuda")
for name, perm in orders:
    a = torch.from_numpy(np.ascontiguousarray(o[perm])).cuda(); b = torch.from_numpy(np.ascontiguousarray(d[perm])).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        st = s.trace_closest_device(a.data_ptr(), b.data_ptr(), m, d_p.data_ptr(), d_t.data_ptr())
    print("%-22s Mrays/s %.1f" % (name, m / st["ms_extend"] / 1e3), flush=True)
