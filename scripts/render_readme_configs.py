"""Renders the README's two configurations (default scene, 1024x1024; simple depth 6 and bidirectional 3+3) on the GPU at
200 spp (the README's count) and at 3200 spp (converged) -> gpurun_out/readme/*.npy, for tests/golden/analyse_readme.py."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from cuda_path_tracer_amd._lib import FLAG_DROP_NONFINITE

cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
s = G.scene_default(cube).apply(Scene())
s.commit((0,))
os.makedirs("gpurun_out/readme", exist_ok=True)
for key, kw in (("simple_d6", dict(cam_depth=6, light_depth=0)), ("bidir_3_3", dict(cam_depth=3, light_depth=3))):
    for spp in (200, 3200):
        img, st = s.render(make_config(1024, 1024, spp=spp, flags=FLAG_DROP_NONFINITE, **kw))
        np.save("gpurun_out/readme/%s_%dspp.npy" % (key, spp), img.astype(np.float16) if spp == 200 else img)
        print(key, spp, "mean", img.mean(axis=(0, 1)), "nonfinite dropped", st["nonfinite_samples"], flush=True)
