"""Mutation test of the oracle's only reference-held pin.  CPU only.

The reference holds no vectors; what pins oracle/oracle.cpp to it are the two renders of its README (1024x1024, 200 samples:
simple depth 6, bidirectional 3+3), reduced to block statistics (tests/golden/reference_readme_renders.npz).  This script
renders those two configurations with the oracle (256x256, 200 spp, as tests/test_oracle_golden.py does) once as is and once
per quirk switch of the -DORC_MUTATIONS build, and tabulates the README-fixture metrics: a switch whose render FAILS the
thresholds of the pin test is a quirk the reference's pictures can see; one that passes is a quirk the pin is blind to --
parity on it rests on reading the source.  The same RNG streams are used for every variant, so the column "vs unmutated" (mean
absolute relative difference of the 64x64 block means against the unmutated oracle render) is almost free of Monte-Carlo
noise and shows whether the switch changes this scene's picture at all.

    python scripts/pin_mutations.py [--size 256] [--spp 200] [--threads 8] [--out profiles/r3_pin_mutations.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import make_config
from oracle import binding as B
from tests.readme_util import CONFIGS, block_stats, compare, cube_mask_of

# thresholds of tests/test_oracle_golden.py::test_oracle_against_the_reference_readme_renders
def passes(r):
    return (r["corr"] > 0.97 and r["mean_dev"] < 0.01 and r["walls"]["sum_dev"] < 0.015 and r["cubes"]["sum_dev"] < 0.03)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--spp", type=int, default=200)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r3_pin_mutations.json"))
    ap.add_argument("--only", default="", help="comma-separated subset of the switches")
    args = ap.parse_args()
    cube = G.write_cube_obj(os.path.join(tempfile.mkdtemp(), "cube.obj"))
    desc = G.scene_default(cube)
    variants = [("unmutated", "mut", ())] + [(m, "mut", (m,)) for m in B.MUTATIONS] + [
        ("fuse_everything (-ffp-contract=fast, deviation 7 kept)", "mut_fma", ()),
        ("fuse_everything, ray_at as written", "mut_fma", ("dev7_unfused",))]
    if args.only:
        keep = set(args.only.split(","))
        variants = [v for v in variants if v[0] == "unmutated" or v[0].split(" ")[0] in keep]
    rows = []
    base_blocks = {}
    S = args.size
    for name, lib, muts in variants:
        B.set_mutations((), "mut")
        B.set_mutations((), "mut_fma")
        B.set_mutations(muts, lib)
        o = desc.apply(B.OracleScene(threads=args.threads, variant=lib))
        o.commit()
        mask = cube_mask_of(o, desc.camera, S)
        row = {"switch": name, "configs": {}}
        for key, kw in CONFIGS.items():
            t0 = time.time()
            img, _ = o.render(make_config(S, S, spp=args.spp, **kw))
            img = np.nan_to_num(img, nan=0.0, posinf=0.0, neginf=0.0)
            r = compare(key, img, S, S, mask)
            blocks = block_stats(img, S, S)[0]
            if name == "unmutated":
                base_blocks[key] = blocks
            vs = float(np.abs(blocks - base_blocks[key]).sum() / base_blocks[key].sum())
            row["configs"][key] = {"mean_dev": r["mean_dev"], "rel_rms": r["rel_rms"], "corr": r["corr"], "walls_sum_dev": r["walls"]["sum_dev"],
                                   "cubes_sum_dev": r["cubes"]["sum_dev"], "passes_pin": bool(passes(r)), "vs_unmutated": vs,
                                   "seconds": round(time.time() - t0, 1)}
            print("%-58s %-18s mean %.4f walls %.4f cubes %.4f rms %.4f corr %.4f  vs unmutated %.5f  %s" % (
                name, key, r["mean_dev"], r["walls"]["sum_dev"], r["cubes"]["sum_dev"], r["rel_rms"], r["corr"], vs,
                "passes" if passes(r) else "FAILS"), flush=True)
        o.close()
        rows.append(row)
        B.set_mutations((), lib)
    json.dump({"size": S, "spp": args.spp, "thresholds": "corr > 0.97, mean_dev < 0.01, walls.sum_dev < 0.015, cubes.sum_dev < 0.03", "rows": rows},
              open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
