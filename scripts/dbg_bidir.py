import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene
obj = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
desc = G.scene_default(obj)
s = desc.apply(Scene()); s.commit((0,))
o = desc.apply(OracleScene()); o.commit()
for Dc, Dl, spp in [(1,1,1),(1,2,1),(1,3,1),(2,1,1),(3,3,1),(3,3,16)]:
    cfg = make_config(64, 48, spp=spp, cam_depth=Dc, light_depth=Dl)
    img, st = s.render(cfg); ref, ctr = o.render(cfg)
    d = np.abs(img-ref)
    exact = (img.view(np.uint32)==ref.view(np.uint32)).all(axis=2).mean()
    close = np.isclose(img, ref, rtol=1e-3, atol=1e-5).all(axis=2).mean()
    print(Dc, Dl, spp, "exact", round(exact,4), "close", round(close,4), "mean gpu/ref", img.mean(), ref.mean(), "rays", st["rays_total"], ctr["rays"])
