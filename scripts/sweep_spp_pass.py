import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
desc = G.scene_icosphere_room(6)
s = desc.apply(Scene()); s.commit((0,))
acc = torch.zeros(1920 * 1080 * 3, device="cuda"); torch.cuda.synchronize()
for S in (16, 32, 64, 128):
    for rep in range(2):
        cfg = make_config(1920, 1080, spp=128, cam_depth=6, samples_per_pass=S)
        st = s.render_device(cfg, acc.data_ptr())
    print("samples_per_pass", S, "Msamples/s %.1f" % (st["samples"] / st["ms_total"] / 1e3), "extend Mrays/s %.1f" % (st["rays_extended"] / st["ms_extend"] / 1e3),
          "extend share %.3f" % (st["ms_extend"] / st["ms_total"]), flush=True)
