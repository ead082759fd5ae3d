"""Deviation 7 has a switch on both sides: the product built with -DMIPT_RAY_AT_UNFUSED evaluates the point on a
ray as the reference's source text says (product, then sum), the oracle's mutation switch "dev7_unfused" does the same.  This
script checks on a GPU that the two AS-WRITTEN forms agree bit for bit with each other, as the two fitted forms do in tests/.

    bash scripts/build_variant.sh unfused -DMIPT_RAY_AT_UNFUSED
    MIPT_LIB=$PWD/variants/libmipt_unfused.so python scripts/check_unfused.py
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle import binding as B


def main():
    assert "unfused" in os.environ.get("MIPT_LIB", ""), "run with MIPT_LIB=variants/libmipt_unfused.so"
    B.set_mutations(("dev7_unfused",), "mut")
    cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
    bad = 0
    for name, desc, cfgs in (
            ("default scene", G.scene_default(cube), [dict(cam_depth=6, light_depth=0), dict(cam_depth=3, light_depth=3)]),
            ("icosphere L4 room", G.scene_icosphere_room(4), [dict(cam_depth=6, light_depth=0), dict(cam_depth=4, light_depth=4)])):
        s = desc.apply(Scene())
        s.commit(devices=(0,))
        o = desc.apply(B.OracleScene(threads=16, variant="mut"))
        o.commit()
        fitted = desc.apply(B.OracleScene(threads=16))
        fitted.commit()
        rng = np.random.default_rng(5)
        n = 1 << 16
        org = rng.uniform(-3.5, 3.5, (n, 3)).astype(np.float32)
        dirs = rng.normal(size=(n, 3)).astype(np.float32)
        prim, t = s.trace_closest(org, dirs)[:2]
        oprim, ot = o.trace_closest(org, dirs)[:2]
        fprim, ft = fitted.trace_closest(org, dirs)[:2]
        same = np.array_equal(prim, oprim) and np.array_equal(t.view(np.uint32), ot.view(np.uint32))
        differs = int((t.view(np.uint32) != ft.view(np.uint32)).sum() + (prim != fprim).sum())
        print("%-18s closest hit, %d rays: as-written product == as-written oracle: %s   (rays where the fitted oracle answers differently: %d)"
              % (name, n, same, differs), flush=True)
        bad += 0 if same else 1
        for kw in cfgs:
            cfg = make_config(96, 64, spp=4, **kw)
            img, _ = s.render(cfg)
            ref, _ = o.render(cfg)
            fit, _ = fitted.render(cfg)
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            print("%-18s render %s: as-written product == as-written oracle: %s   (pixels where the fitted oracle differs: %d of %d)"
                  % (name, kw, same, int((ref.view(np.uint32) != fit.view(np.uint32)).any(axis=-1).sum()), 96 * 64), flush=True)
            bad += 0 if same else 1
        s.close()
        o.close()
        fitted.close()
    print("OK" if bad == 0 else "MISMATCH in %d checks" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
