"""Would ordering secondary rays by direction octant inside 512-ray groups help the extend kernel?
First-bounce rays of the C2 scene (tile-ordered pixels, cosine-ish hemisphere directions) traced as they come vs sorted."""
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
prim, t, _ = s.trace_closest(ro, rd)
d = rd / (np.linalg.norm(rd, axis=1, keepdims=True) + 1e-5)
P = ro + d * t[:, None]
hit = prim >= 0
c = np.array([0, -1, -2], np.float32)
ntri = (P - c) / 2.5
n = np.where((prim < 81920)[:, None], ntri, 0)
# walls: normal = axis of the coordinate at +-5
wall = (prim >= 81920)
ax = np.argmax(np.abs(P), axis=1)
nw = np.zeros_like(P); nw[np.arange(len(P)), ax] = -np.sign(P[np.arange(len(P)), ax])
n = np.where(wall[:, None], nw, n).astype(np.float32)
rng = np.random.Generator(np.random.Philox(key=3))
v = rng.standard_normal(P.shape).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
nd = n + v * 0.999  # cosine-weighted hemisphere
keep = hit
o2 = (P + n * 1e-5)[keep].astype(np.float32); d2 = nd[keep].astype(np.float32)
m = (len(o2) // 512) * 512
o2, d2 = o2[:m], d2[:m]
octant = (d2[:, 0] > 0).astype(np.int32) | ((d2[:, 1] > 0).astype(np.int32) << 1) | ((d2[:, 2] > 0).astype(np.int32) << 2)
grp = np.arange(m) // 512
order_sorted = np.lexsort((np.arange(m), octant, grp))
grp64k = np.arange(m) // 65536
order_sorted_big = np.lexsort((np.arange(m), octant, grp64k))
d_p = torch.empty(m, dtype=torch.int32, device="cuda"); d_t = torch.empty(m, dtype=torch.float32, device="cuda")
for name, perm in (("as generated", np.arange(m)), ("octant-sorted in 512 groups", order_sorted), ("octant-sorted in 64k groups", order_sorted_big),
                   ("shuffled", rng.permutation(m))):
    a = torch.from_numpy(np.ascontiguousarray(o2[perm])).cuda(); b = torch.from_numpy(np.ascontiguousarray(d2[perm])).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        st = s.trace_closest_device(a.data_ptr(), b.data_ptr(), m, d_p.data_ptr(), d_t.data_ptr())
    print(name, m, "Mrays/s %.1f" % (m / st["ms_extend"] / 1e3), flush=True)
