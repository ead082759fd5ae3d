"""`python -m cuda_path_tracer_amd` -- the reference's main() (kernel.cu:374-431) on libmipt: build the default scene
(or a procedural one), render on the GPU, write resources/out/image.tga-style output (+ optional PFM)."""
import argparse
import os
import tempfile
import time

from . import generators as G
from .image_io import write_pfm, write_tga
from .scene import Scene, make_config


def main():
    ap = argparse.ArgumentParser(prog="cuda_path_tracer_amd")
    ap.add_argument("--scene", default="default", choices=["default", "icosphere", "env-sphere", "multi-sphere"])
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--width", type=int, default=1024)      # constants.cuh:6
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=200)         # constants.cuh:9
    ap.add_argument("--cam-depth", type=int, default=3)     # constants.cuh:8
    ap.add_argument("--light-depth", type=int, default=3, help="0 = simple path tracing")
    ap.add_argument("--env-dir", default=None, help="folder with px,py,pz,nx,ny,nz.hdr")
    ap.add_argument("--seed", type=int, default=G.SEED)
    ap.add_argument("--devices", default="0", help="comma separated HIP device ids")
    ap.add_argument("--out", default="image.tga")
    ap.add_argument("--pfm", default=None)
    a = ap.parse_args()
    if a.scene == "default":
        desc = G.scene_default(G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj")))
    elif a.scene == "icosphere":
        desc = G.scene_icosphere_room(a.level)
    elif a.scene == "env-sphere":
        desc = G.scene_env_sphere(a.level, env_dir=a.env_dir, env_f32=None if a.env_dir else G.synthetic_env_faces(64))
    else:
        desc = G.scene_multi_sphere(a.level)
    s = desc.apply(Scene())
    t0 = time.time()
    s.commit(devices=[int(x) for x in a.devices.split(",")])
    print("scene committed in %.2f s (kd build %.2f s, %.1f MB)" % (time.time() - t0, s.build_seconds(), s.device_bytes() / 1e6))
    cfg = make_config(a.width, a.height, a.spp, a.cam_depth, a.light_depth, use_envmap=desc.env_dir is not None or desc.env_f32 is not None,
                      seed=a.seed)
    print("Render strated")                                  # sic, kernel.cu:394
    img, st = s.render(cfg)
    print("Render ended in %.4f" % (st["ms_total"] * 1e-3), "(%.1f Mrays/s, %.3g samples/s)" %
          (st["rays_total"] / st["ms_total"] / 1e3, st["samples"] / st["ms_total"] * 1e3))
    write_tga(a.out, img)
    if a.pfm:
        write_pfm(a.pfm, img)


if __name__ == "__main__":
    main()
