"""Lane use per phase of the persistent closest-hit kernel (developer build variants/libmipt_stats.so, made by
scripts/build_phase_stats.sh; the library prints one table per call on stderr) for one C2 bench step and the two
closest-hit batches, plus the refill rate per cursor word (one word took 72 M atomics/s of a ~88 M/s
cap; now 8 words).  usage: python scripts/phase_lane_use.py  2>&1 | tee profiles/r3_phase_lane_use.txt
(-DMIPT_PHASE_CLOCKS build, variants/libmipt_clocks.so: the same runs print the shader clocks per section of the loop instead)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_path_tracer_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("MIPT_LIB") or os.path.join(ROOT, "variants", "libmipt_stats.so")  # MIPT_LIB=variants/libmipt_clocks.so: clocks per section
from cuda_path_tracer_amd import generators as G  # noqa: E402
from cuda_path_tracer_amd.scene import Scene, make_config  # noqa: E402

s = G.scene_icosphere_room(6).apply(Scene())
s.commit((0,))
s.render(make_config(1920, 1080, spp=64, cam_depth=6))  # warm-up (its table is printed too)
sys.stderr.flush()
print("==== C2: one bench step (1920x1080, 64 spp, simple depth 6) ====", flush=True)
_, st = s.render(make_config(1920, 1080, spp=64, cam_depth=6))
sys.stderr.flush()
print("k_extend: %d launches, %.2f ms in all, %d rays" % (st["extend_launches"], st["ms_extend"], st["rays_extended"]), flush=True)
print("(refill events per second and cursor word = `refill exec` / 8 words / %.4f s)" % (st["ms_extend"] * 1e-3), flush=True)
s.close()
desc = G.scene_env_sphere(8)
desc.planes.clear()
s = desc.apply(Scene())
s.commit((0,))
n = 1 << 24
for kind in ("coherent", "incoherent"):
    ro, rd = G.camera_rays(desc.camera, 3840, 2160, n) if kind == "coherent" else G.incoherent_rays(n, (-2.5,) * 3, (2.5,) * 3)
    print("==== closest hit, 1,310,720 triangles, 2^24 %s rays ====" % kind, flush=True)
    _, _, st = s.trace_closest(ro, rd)
    sys.stderr.flush()
    print("k_extend: %.3f ms" % st["ms_extend"], flush=True)
