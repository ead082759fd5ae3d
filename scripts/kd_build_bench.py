"""kd-tree build time, GPU (kd_build_gpu.hip) vs host threads (kd_build.cpp): icosphere L6 / L8 and the 8 x L8 scene of C5.
usage: python scripts/kd_build_bench.py [gpu|host|both]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene

which = sys.argv[1] if len(sys.argv) > 1 else "both"
scenes = (("icosphere L6 (81,920 tris)", lambda: G.scene_icosphere_room(6)), ("icosphere L8 (1,310,720 tris)", lambda: G.scene_env_sphere(8)),
          ("C5: 8 x icosphere L8 (10,485,760 tris)", lambda: G.scene_multi_sphere(8)))
for name, make in scenes:
    desc = make()
    for builder in (("gpu", "host") if which == "both" else (which,)):
        if builder == "host":
            os.environ["MIPT_KD_BUILD"] = "host"
        else:
            os.environ.pop("MIPT_KD_BUILD", None)
        for rep in range(2):  # the second commit of a process has the code objects loaded
            s = desc.apply(Scene())
            t0 = time.perf_counter()
            s.commit((0,))
            wall = time.perf_counter() - t0
            kd, total, ngpu = s.kd_seconds(), s.build_seconds(), s.gpu_built_meshes()
            s.close()
        print("%-42s %-4s kd-trees %.3f s  pack %.3f s  commit wall %.3f s  (meshes built on the GPU: %d)" % (name, builder, kd, total - kd, wall, ngpu), flush=True)
