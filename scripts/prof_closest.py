"""Profiling driver: closest hit on the 1.3M-triangle icosphere (C4 mesh), one batch kind, few launches.
usage: python scripts/prof_closest.py [coherent|incoherent] [log2 rays] [repeats] [level]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene

kind = sys.argv[1] if len(sys.argv) > 1 else "incoherent"
n = 1 << int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
level = int(sys.argv[4]) if len(sys.argv) > 4 else 8
desc = G.scene_env_sphere(level)
desc.planes.clear()
s = desc.apply(Scene())
s.commit((0,))
ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,) * 3, (2.5,) * 3)
d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
d_p = torch.empty(n, dtype=torch.int32, device="cuda")
d_t = torch.empty(n, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(rep):
    st = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_p.data_ptr(), d_t.data_ptr())
    print(kind, n, "ms_extend %.3f" % st["ms_extend"], "Mrays/s %.1f" % (n / st["ms_extend"] / 1e3), flush=True)
