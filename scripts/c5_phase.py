import os, sys
sys.path.insert(0, os.getcwd())
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
s = G.scene_multi_sphere(8).apply(Scene()); s.commit((0,))
cfg = make_config(3840, 2160, spp=2, cam_depth=5, light_depth=5)
s.render(cfg)
sys.stderr.flush(); print("==== second render ====", flush=True)
_, st = s.render(cfg)
print({k: st[k] for k in ("rays_total", "rays_extended", "ms_extend", "ms_total", "extend_launches", "samples")})
