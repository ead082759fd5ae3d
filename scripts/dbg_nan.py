import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene
cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
desc = G.scene_default(cube)
kw = dict(width=1024, height=1024, spp=200, cam_depth=3, light_depth=3)
s = desc.apply(Scene()); s.commit((0,))
o = desc.apply(OracleScene(threads=16)); o.commit()
found = 0
for smp in range(200):
    c1 = make_config(**{**kw, "sample_begin": smp, "sample_end": smp + 1})
    img, _ = s.render(c1)
    bad = ~np.isfinite(img).all(axis=2)
    if bad.any():
        ref, _ = o.render(c1)
        obad = ~np.isfinite(ref).all(axis=2)
        print("sample", smp, "gpu non-finite", int(bad.sum()), "oracle non-finite", int(obad.sum()), "same pixels", bool((bad == obad).all()),
              "rest bit-equal", bool(np.array_equal(img[~bad], ref[~bad])), "values", img[bad][:2].tolist())
        found += 1
        if found >= 3:
            break
print("done", found)
