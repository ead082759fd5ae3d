"""Would ordering the shadow rays of a bounce by direction help the extend kernel?  They all start at the light, so a
direction sort turns an incoherent batch into bundles like primary rays.  Second-vertex shadow rays of the C2 scene
(light -> second hit of a cosine-bounced camera path), restricted to those that pass the sphere's bounding box,
traced as generated (path order) vs binned by direction cell."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene
desc = G.scene_icosphere_room(6)
s = desc.apply(Scene()); s.commit((0,))
W, H = 1920, 1080
ro, rd = G.camera_rays(desc.camera, W, H, W * H)
idx = np.arange(W * H).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
ro, rd = ro[idx], rd[idx]
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


P1, n1, ok1 = hit_points(ro, rd)
v = rng.standard_normal(P1.shape).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
P2, n2, ok2 = hit_points((P1 + n1 * 1e-5)[ok1], (n1 + v * 0.999)[ok1])
lp = np.array([0.0, -4.0, -4.5], np.float32)
tgt = P2[ok2]
d = tgt - lp
d /= np.linalg.norm(d, axis=1, keepdims=True)
# keep the rays that pass the sphere's bounding box (the others never reach the kernel in a render)
lo, hi = c - 2.5, c + 2.5
with np.errstate(divide="ignore", invalid="ignore"):
    t0, t1 = (lo - lp) / d, (hi - lp) / d
tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
keep = (tn <= tf) & (tf > 0)
d = np.ascontiguousarray(d[keep]); m = len(d)
o = np.broadcast_to(lp, d.shape).copy()
print("shadow rays through the box:", m, flush=True)


def cell(d, n):  # octahedral map -> n x n cells
    a = d / np.abs(d).sum(axis=1, keepdims=True)
    u, w = a[:, 0], a[:, 2]
    neg = a[:, 1] < 0
    u2 = np.where(neg, (1 - np.abs(w)) * np.sign(u), u); w2 = np.where(neg, (1 - np.abs(a[:, 0])) * np.sign(w), w)
    iu = np.clip(((u2 * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1); iw = np.clip(((w2 * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    return iu * n + iw


d_p = torch.empty(m, dtype=torch.int32, device="cuda"); d_t = torch.empty(m, dtype=torch.float32, device="cuda")
orders = [("as generated", np.arange(m))]
for n in (16, 64, 256, 1024):
    orders.append(("binned %dx%d" % (n, n), np.argsort(cell(d, n), kind="stable")))
orders.append(("shuffled", rng.permutation(m)))
for name, perm in orders:
    a = torch.from_numpy(np.ascontiguousarray(o[perm])).cuda(); b = torch.from_numpy(np.ascontiguousarray(d[perm])).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        st = s.trace_closest_device(a.data_ptr(), b.data_ptr(), m, d_p.data_ptr(), d_t.data_ptr())
    print("%-18s Mrays/s %.1f" % (name, m / st["ms_extend"] / 1e3), flush=True)
