"""How much would ordering the incoherent closest-hit batch buy?  (1,310,720-triangle icosphere, 2^24 rays)
Orders tried: as generated; by answer (hit triangle; an upper bound no key can reach); by a practical key = Morton code of the
point where the ray enters the mesh's bounding box (10 bits per axis) [+ direction octant]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene

n = 1 << 24
desc = G.scene_env_sphere(8)
desc.planes.clear()
s = desc.apply(Scene()); s.commit((0,))
ro, rd = G.incoherent_rays(n, (-2.5,) * 3, (2.5,) * 3)

def run(ro, rd, tag):
    d_o = torch.from_numpy(ro).cuda(); d_d = torch.from_numpy(rd).cuda()
    d_p = torch.empty(n, dtype=torch.int32, device="cuda"); d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for _ in range(4):
        st = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_p.data_ptr(), d_t.data_ptr())
        ms.append(st["ms_extend"])
    print("%-28s %.3f ms  %.1f Mrays/s" % (tag, np.median(ms), n / np.median(ms) / 1e3), flush=True)
    return d_p.cpu().numpy()

prim = run(ro, rd, "as generated")
order = np.argsort(prim, kind="stable")
run(ro[order], rd[order], "sorted by answer (bound)")

def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x

d = rd / (np.linalg.norm(rd, axis=1, keepdims=True) + 1e-5)
lo, hi = -2.5, 2.5
with np.errstate(divide="ignore", invalid="ignore"):
    t0 = (lo - ro) / d; t1 = (hi - ro) / d
tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
tn = np.where(tn <= tf, np.maximum(tn, 0), 0)
P = ro + d * tn[:, None]
for bits in (6, 8, 10):
    q = np.clip(((P - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    order = np.argsort(key, kind="stable")
    run(ro[order], rd[order], "entry-point Morton %d bits" % bits)
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
    order = np.argsort((key << 3) | octant, kind="stable")
    run(ro[order], rd[order], "  + direction octant")
