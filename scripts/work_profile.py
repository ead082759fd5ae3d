"""What the production closest-hit kernel reads per ray (MIPT_FLAG_WORK_COUNTERS): L8 mesh, coherent / incoherent batches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
desc = G.scene_env_sphere(level)
desc.planes.clear()
s = desc.apply(Scene())
s.commit((0,))
for kind in ("coherent", "incoherent"):
    ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,) * 3, (2.5,) * 3)
    p, t, st = s.trace_closest(ro, rd, work_counters=True)
    w = {k[5:]: st[k] / n for k in st if k.startswith("work_")}
    b = w["steps"] * 80 + w["pops"] * 40 + w["sphere_pairs"] * 32 + w["planes"] * 16 + w["verts"] * 48 + 48
    print(kind, "hit %.3f" % (p >= 0).mean(), {k: round(v, 2) for k, v in w.items()}, "product bytes/ray %.0f" % b, flush=True)
