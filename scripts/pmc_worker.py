"""Workload of bench.py's PMC passes: run under `rocprofv3 --pmc ... -- python3 scripts/pmc_worker.py <manifest.json>`.

Launches, in this order, the closest-hit kernel of (1) one bench step of C2 (1920x1080, icosphere L6, simple depth 6,
--spp samples per pixel), (2) a coherent and (3) an incoherent batch of 2^--log2-rays rays on the 1,310,720-triangle
icosphere, and writes how many production launches (k_extend<0>) each part made, so that the per-dispatch counter rows
can be attributed by order.  No torch: host buffers through the C-ABI (the copies are not what is being counted)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cuda_path_tracer_amd import generators as G  # noqa: E402
from cuda_path_tracer_amd.scene import Scene, make_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("manifest")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--log2-rays", type=int, default=22)
ap.add_argument("--repeats", type=int, default=2)
ap.add_argument("--no-closest", action="store_true")
a = ap.parse_args()

parts = []
s = G.scene_icosphere_room(a.level).apply(Scene())
s.commit((0,))
_, st = s.render(make_config(a.width, a.height, spp=a.spp, cam_depth=6, light_depth=0))
parts.append({"name": "c2", "launches": st["extend_launches"], "rays": st["rays_extended"]})
s.close()
if not a.no_closest:
    desc = G.scene_env_sphere(8)
    desc.planes.clear()
    s = desc.apply(Scene())
    s.commit((0,))
    n = 1 << a.log2_rays
    for kind in ("coherent", "incoherent"):
        ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,) * 3, (2.5,) * 3)
        for _ in range(a.repeats):
            s.trace_closest(ro, rd)
        parts.append({"name": kind, "launches": a.repeats, "rays": n * a.repeats})
    s.close()
json.dump({"parts": parts}, open(a.manifest, "w"))
