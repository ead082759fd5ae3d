"""Throughput of the BASELINE.json configurations other than the headline one (single GPU, reduced spp where the
full sample count would take minutes; samples are independent, so samples/s does not depend on spp).
usage: python scripts/bench_configs.py [C1 C3 C4 C5]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config

which = sys.argv[1:] or ["C1", "C3", "C4", "C5"]
cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))


def env_faces_1024():
    rng = np.random.Generator(np.random.Philox(key=G.SEED))
    base = rng.random((6, 64, 64, 3), dtype=np.float32)
    return np.ascontiguousarray(np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)) * np.float32(2.0)


CONFIGS = {
    "C1": lambda: (G.scene_default(cube), dict(width=640, height=480, spp=16, cam_depth=6, light_depth=0), "default scene, simple d6, 640x480, 16 spp"),
    "C3": lambda: (G.scene_default(cube), dict(width=1024, height=1024, spp=200, cam_depth=3, light_depth=3), "default scene, bidirectional 3+3, 1024x1024, 200 spp"),
    "C4": lambda: (G.scene_env_sphere(8, env_f32=env_faces_1024()), dict(width=3840, height=2160, spp=16, cam_depth=6, light_depth=0, use_envmap=True),
                   "env map + icosphere L8 (1,310,720 tris), simple d6, 4K, 16 of 1024 spp"),
    "C5": lambda: (G.scene_multi_sphere(8), dict(width=3840, height=2160, spp=4, cam_depth=5, light_depth=5),
                   "8 x icosphere L8 (10,485,760 tris), bidirectional 5+5, 4K, 4 of 4096 spp"),
}
for name in which:
    desc, kw, text = CONFIGS[name]()
    t0 = time.time()
    s = desc.apply(Scene())
    s.commit((0,))
    t_commit = time.time() - t0
    cfg = make_config(**kw)
    warm = make_config(**{**kw, "spp": kw["spp"], "sample_end": 1})
    s.render(warm)
    t0 = time.time()
    img, st = s.render(cfg)
    wall = time.time() - t0
    out = {"config": name, "workload": text, "samples_per_sec": round(st["samples"] / (st["ms_total"] * 1e-3), 1),
           "Mrays_per_s": round(st["rays_total"] / (st["ms_total"] * 1e-3) / 1e6, 1),
           "rays_per_sample": round(st["rays_total"] / st["samples"], 2), "ms_gpu": round(st["ms_total"], 1),
           "extend_share": round(st["ms_extend"] / st["ms_total"], 3), "wall_s_incl_readback": round(wall, 3),
           "commit_s": round(t_commit, 2), "kd_build_s": round(s.build_seconds(), 2), "arena_MB": round(s.device_bytes() / 1e6, 1),
           "image_mean_finite_pixels": float(img[np.isfinite(img).all(axis=2)].mean()),
           "nonfinite_pixels": int((~np.isfinite(img).all(axis=2)).sum()), "nonfinite_samples": int(st["nonfinite_samples"]),
           "nonfinite_note": "the reference's own arithmetic (DESIGN.md 4.2: light_pdf == 0 at kernel.cu:277; sqrtf of a negative at "
                             "samplers.cuh:19 through the env term); identical in the oracle, pinned by tests/test_gpu_full_configs.py"}
    print(json.dumps(out), flush=True)
    s.close()
