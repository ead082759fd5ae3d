"""C3 (default scene, bidirectional 3+3, 1024x1024) throughput vs samples per pass."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
s = G.scene_default(cube).apply(Scene()); s.commit((0,))
acc = torch.zeros(1024 * 1024 * 3, device="cuda"); torch.cuda.synchronize()
for S in [int(x) for x in (sys.argv[1:] or ["4","8","16","32","64"])]:
    for rep in range(2):
        cfg = make_config(1024, 1024, spp=64, cam_depth=3, light_depth=3, samples_per_pass=S)
        st = s.render_device(cfg, acc.data_ptr())
    print("samples_per_pass", S, "Msamples/s %.1f" % (st["samples"] / st["ms_total"] / 1e3), "extend share %.3f" % (st["ms_extend"] / st["ms_total"]), flush=True)
