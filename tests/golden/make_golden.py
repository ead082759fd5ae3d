#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

PARITY UNPINNED: the reference ships no tests or fixtures and could not be built or run
here, so these vectors come from the build's own restatement of the reference (oracle/).  They pin
the oracle against regressions and give the GPU tests fixed expected outputs; the analytic known-answer tests in
tests/test_oracle_kat.py are what ties the oracle to the reference's formulas.

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuda_path_tracer_amd import generators as G  # noqa: E402
from cuda_path_tracer_amd.scene import make_config  # noqa: E402
from oracle.binding import OracleScene, hdr_load  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def golden_rays(desc, n):
    o1, d1 = G.camera_rays(desc.camera, 64, 64, n // 2, seed=11)
    o2, d2 = G.incoherent_rays(n - n // 2, (-3, -4.5, -4.5), (3, 1.5, 0.5), seed=12)
    return np.concatenate([o1, o2]), np.concatenate([d1, d2])


def kd_summary(k):
    leaf = k["left"] < 0
    interior = ~leaf
    # numbering-independent: multiset of split planes / leaf sizes, hashed in sorted order
    splits = []
    for i in np.flatnonzero(interior):
        ax = int(k["axis"][i])
        splits.append((ax, float(k["boxes"][k["left"][i]][2 * ax + 1])))
    splits.sort()
    h = hashlib.sha256(np.array(splits, np.float64).tobytes()).hexdigest()
    sizes = np.sort(k["count"][leaf])
    return {"n_interior": int(interior.sum()), "n_leaves": int(leaf.sum()), "n_refs": int(len(k["refs"])),
            "max_leaf": int(sizes.max()), "leaf_sizes_sha": hashlib.sha256(sizes.astype(np.int32).tobytes()).hexdigest(),
            "splits_sha": h, "root_box": [float(x) for x in k["boxes"][0]]}


def main():
    tmp = tempfile.mkdtemp()
    cube = G.write_cube_obj(os.path.join(tmp, "cube.obj"))
    out = {}
    import json
    kd = {}
    for name, desc in (("default", G.scene_default(cube)), ("ico3", G.scene_icosphere_room(3)), ("ico4", G.scene_icosphere_room(4))):
        o = desc.apply(OracleScene())
        o.commit()
        ro, rd = golden_rays(desc, 4096)
        prim, t, ctr = o.trace_closest(ro, rd)
        bprim, bt, _ = o.trace_closest(ro, rd, brute=True)
        out[name + "_ray_o"] = ro
        out[name + "_ray_d"] = rd
        out[name + "_prim"] = prim
        out[name + "_t"] = t
        out[name + "_brute_agree"] = np.array([(prim == bprim).mean()], np.float64)
        kd[name] = [kd_summary(o.kd_export(m)) for m in range(o.mesh_count())]
        print(name, "hit frac", (prim >= 0).mean(), "kd-vs-brute agreement", (prim == bprim).mean(), ctr)
    # env lookup goldens on small synthetic faces
    faces = G.synthetic_env_faces(8)
    desc = G.scene_env_sphere(1, env_f32=faces)
    o = desc.apply(OracleScene())
    o.commit()
    rng = np.random.default_rng(5)
    dirs = rng.standard_normal((44, 3)).astype(np.float32)
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    edges = np.array([[1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0], [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1],
                      [0, 1, 1], [0, 1, -1], [0, -1, 1], [0, -1, -1]], np.float32)
    dirs = np.concatenate([dirs, corners, edges])
    rgb, face, pos = o.env_lookup(dirs, with_index=True)
    out["env_dirs"], out["env_rgb"], out["env_face"], out["env_pos"] = dirs, rgb, face, pos
    # image golden: C1 (640x480, 16 spp, simple depth 6) as 80x60 block means + global statistics
    desc = G.scene_default(cube)
    o = desc.apply(OracleScene())
    o.commit()
    cfg = make_config(640, 480, spp=16, cam_depth=6, light_depth=0, seed=G.SEED)
    img, ctr = o.render(cfg)
    out["c1_block_means"] = img.reshape(60, 8, 80, 8, 3).mean(axis=(1, 3)).astype(np.float32)
    out["c1_stats"] = np.array([img.mean(), img.var(), float(ctr["rays"])], np.float64)
    out["c1_sha256"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)
    cfg = make_config(128, 96, spp=4, cam_depth=3, light_depth=3, seed=G.SEED)
    img, _ = o.render(cfg)
    out["bidir_small"] = img
    np.savez_compressed(os.path.join(OUT, "golden.npz"), **out)
    # checksums of the six shipped Radiance faces as decoded by the oracle loader (only when the reference is mounted)
    env = G.reference_env_dir()
    if env:
        kd["hdr_faces"] = {}
        for f in ("px", "py", "pz", "nx", "ny", "nz"):
            cols = hdr_load(env + f + ".hdr")
            kd["hdr_faces"][f] = {"shape": list(cols.shape), "sha256": hashlib.sha256(cols.tobytes()).hexdigest(),
                                  "mean": float(cols.mean(dtype=np.float64))}
    with open(os.path.join(OUT, "kd_and_assets.json"), "w") as fh:
        json.dump(kd, fh, indent=1, sort_keys=True)
    print("wrote", os.path.join(OUT, "golden.npz"))


if __name__ == "__main__":
    main()
