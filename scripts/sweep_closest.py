"""Closest-hit batches of the 1,310,720-triangle mesh once per value of an environment variable (MIPT_TUNE for batches in
unknown order, MIPT_TUNE_BATCH_COHERENT for batches the kernel finds coherent).  One process per setting.
usage: python scripts/sweep_closest.py MIPT_TUNE_BATCH_COHERENT "48,16,4,40 40,16,4,40 ..." """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene
n = 1 << 24
desc = G.scene_env_sphere(8); desc.planes.clear()
s = desc.apply(Scene()); s.commit((0,))
out = []
for kind in ("coherent", "incoherent"):
    ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,)*3, (2.5,)*3)
    d_o = torch.from_numpy(ro).cuda(); d_d = torch.from_numpy(rd).cuda()
    d_p = torch.empty(n, dtype=torch.int32, device="cuda"); d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    ms = [s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_p.data_ptr(), d_t.data_ptr())["ms_extend"] for _ in range(5)]
    out.append("%%s %%.1f" %% (kind, n / np.median(ms[1:]) / 1e3))
print(" ".join(out))
''' % ROOT
var, values = sys.argv[1], sys.argv[2].split()
for v in values:
    r = subprocess.run([sys.executable, "-c", WORKER], env=dict(os.environ, **{var: v}), capture_output=True, text=True, timeout=200)
    print("%s=%-14s %s" % (var, v, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
