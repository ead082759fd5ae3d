"""Where do the non-finite pixels of the long renders come from?  ("Non-finite pixels, unasserted".)

Renders the README's bidirectional configuration (C3: default scene, 1024x1024, 3+3) on the GPU one sample index at a
time, and for every pixel whose sample is not finite asks the ORACLE (orc_explain_sample) which operand of which
reference expression produced it.  Writes gpurun_out/nonfinite_c3.txt; tests/test_gpu_nonfinite.py pins what it found.
With `c4` as third argument: the open environment scene of C4 (cube map + 1,310,720-triangle mirror-ish icosphere + floor plane,
3840x2160, simple depth 6) -> gpurun_out/nonfinite_c4.txt.
usage: python scripts/find_nonfinite.py [max_samples] [want] [c3|c4]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene

max_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 200
want = int(sys.argv[2]) if len(sys.argv) > 2 else 6
which = sys.argv[3] if len(sys.argv) > 3 else "c3"
if which == "c3":
    cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
    desc = G.scene_default(cube)
    kw = dict(width=1024, height=1024, spp=200, cam_depth=3, light_depth=3)
else:
    desc = G.scene_env_sphere(8, env_f32=G.synthetic_env_faces(64))
    kw = dict(width=3840, height=2160, spp=1024, cam_depth=6, light_depth=0, use_envmap=True)
s = desc.apply(Scene())
s.commit((0,))
o = desc.apply(OracleScene(threads=16))
o.commit()
os.makedirs("gpurun_out", exist_ok=True)
out = open("gpurun_out/nonfinite_%s.txt" % which, "w")


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    out.write(line + "\n")
    out.flush()


found = 0
for smp in range(max_samples):
    cfg = make_config(**{**kw, "sample_begin": smp, "sample_end": smp + 1})
    img, st = s.render(cfg)
    bad = ~np.isfinite(img).all(axis=2)
    if not bad.any():
        continue
    for pix in np.flatnonzero(bad.reshape(-1)):
        rgb, pr = o.explain_sample(cfg, int(pix), smp)
        say("sample", smp, "pixel", int(pix), "gpu", img.reshape(-1, 3)[pix].tolist(), "stat nonfinite_samples", st["nonfinite_samples"],
            "oracle sample rgb", rgb.tolist(), "probe", pr)
        found += 1
    if found >= want:
        break
say("done: %d non-finite pixel-samples in the first %d sample indices" % (found, smp + 1))
