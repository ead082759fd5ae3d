"""Sweep of the persistent kernel's scheduling knobs (MIPT_TUNE=refill,batch,cap) on the closest-hit batches of the
1,310,720-triangle mesh (the knobs shipped were tuned on C2's 81,920 triangles).  One process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene
n = 1 << 23
desc = G.scene_env_sphere(8); desc.planes.clear()
s = desc.apply(Scene()); s.commit((0,))
out = []
for kind in ("coherent", "incoherent"):
    ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,)*3, (2.5,)*3)
    d_o = torch.from_numpy(ro).cuda(); d_d = torch.from_numpy(rd).cuda()
    d_p = torch.empty(n, dtype=torch.int32, device="cuda"); d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    ms = [s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_p.data_ptr(), d_t.data_ptr())["ms_extend"] for _ in range(4)]
    out.append("%%s %%.1f" %% (kind, n / np.median(ms[1:]) / 1e3))
print(" ".join(out))
''' % ROOT
tunes = sys.argv[1:] or ["32,28,4", "32,28,6", "32,28,8", "32,28,12", "24,28,6", "40,28,6", "48,28,6", "32,20,6", "32,36,6", "32,44,6", "16,28,6", "24,20,8", "40,36,8"]
import json
for t in tunes:
    tc = os.environ.get("COHERENT_REFILL", "60") + "," + ",".join(t.split(",")[1:])  # coherent queues: same batch / cap, own refill
    env = dict(os.environ, MIPT_TUNE=t, MIPT_TUNE_COHERENT=tc)
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=200)
    line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-200:]
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--no-closest"],
                       env=env, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
        c2 = "C2 %.1f Msamples/s (kernel %.1f Mrays/s)" % (d["value"] / 1e6, d["roofline"]["Mrays_per_s_in_kernel"])
    except Exception:
        c2 = "C2 FAILED " + b.stderr[-200:]
    print("TUNE=%-10s COHERENT=%-10s %s  %s" % (t, tc, line, c2), flush=True)
