"""CPU: the oracle reproduces the committed golden vectors (tests/golden/make_golden.py), agrees with its own
brute-force search, and its asset readers behave as specified."""
import hashlib
import json
import os

import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import make_config
from oracle import binding as B
from tests.hdr_util import decode_rgbe, random_rgbe, write_hdr

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "kd_and_assets.json")))


def _scene(name, cube_obj):
    desc = {"default": lambda: G.scene_default(cube_obj), "ico3": lambda: G.scene_icosphere_room(3),
            "ico4": lambda: G.scene_icosphere_room(4)}[name]()
    o = desc.apply(B.OracleScene())
    o.commit()
    return desc, o


@pytest.mark.parametrize("name", ["default", "ico3", "ico4"])
def test_closest_hit_goldens(name, cube_obj):
    _, o = _scene(name, cube_obj)
    prim, t, ctr = o.trace_closest(GOLD[name + "_ray_o"], GOLD[name + "_ray_d"])
    assert np.array_equal(prim, GOLD[name + "_prim"])
    assert np.array_equal(t.view(np.uint32), GOLD[name + "_t"].view(np.uint32))
    assert ctr["stack_overflow"] == 0
    bprim, bt, _ = o.trace_closest(GOLD[name + "_ray_o"], GOLD[name + "_ray_d"], brute=True)
    agree = (prim == bprim).mean()
    assert agree == GOLD[name + "_brute_agree"][0] and agree > 0.99  # Q6/Q8 pin-holes are the remainder
    # where kd and brute force disagree, the kd answer is never nearer than brute force (it can only lose hits)
    bad = prim != bprim
    assert (np.where(t[bad] > 0, t[bad], np.inf) >= bt[bad]).all()


def test_kd_structure_goldens(cube_obj):
    from tests.golden.make_golden import kd_summary
    for name in ("default", "ico3", "ico4"):
        _, o = _scene(name, cube_obj)
        got = [kd_summary(o.kd_export(m)) for m in range(o.mesh_count())]
        assert got == META[name], name
    assert META["default"][0]["n_leaves"] == 1 and META["default"][0]["n_refs"] == 12  # +-1 cube: one leaf, 12 triangles


def test_env_lookup_goldens():
    desc = G.scene_env_sphere(1, env_f32=G.synthetic_env_faces(8))
    o = desc.apply(B.OracleScene())
    o.commit()
    rgb, face, pos = o.env_lookup(GOLD["env_dirs"], with_index=True)
    assert np.array_equal(rgb, GOLD["env_rgb"]) and np.array_equal(face, GOLD["env_face"]) and np.array_equal(pos, GOLD["env_pos"])
    # envMap.cuh:50-89: +x looks at px (face 0), -y at ny (face 4); the centre of +z maps to the middle texel
    rgb, face, pos = o.env_lookup(np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1]], np.float32), with_index=True)
    assert list(face) == [0, 4, 2]
    assert pos[2] == (4 * 8 + 4) * 3


def test_image_goldens(cube_obj):
    _, o = _scene("default", cube_obj)
    img, ctr = o.render(make_config(640, 480, spp=16, cam_depth=6, light_depth=0, seed=G.SEED))
    assert hashlib.sha256(img.tobytes()).digest() == GOLD["c1_sha256"].tobytes()
    assert np.allclose(img.reshape(60, 8, 80, 8, 3).mean(axis=(1, 3)), GOLD["c1_block_means"], rtol=1e-6)
    img, _ = o.render(make_config(128, 96, spp=4, cam_depth=3, light_depth=3, seed=G.SEED))
    assert np.array_equal(img, GOLD["bidir_small"])


def test_render_properties(cube_obj):
    _, o = _scene("default", cube_obj)
    full, _ = o.render(make_config(48, 32, spp=6, cam_depth=4))
    assert np.isfinite(full).all() and full.min() >= 0 and full.mean() > 0.05
    # sample ranges partition the estimate: sum over disjoint strided subsets == full image (up to fp32 summation order)
    parts = sum(o.render(make_config(48, 32, spp=6, cam_depth=4, sample_begin=g, sample_end=6, sample_stride=3))[0] for g in range(3))
    assert np.allclose(parts, full, rtol=2e-6, atol=1e-7)
    # different seed -> different noise, same expectation
    other, _ = o.render(make_config(48, 32, spp=6, cam_depth=4, seed=123))
    assert not np.array_equal(other, full) and abs(other.mean() - full.mean()) < 0.15 * full.mean()
    # Q11: with a light sub-path, a sample whose first light ray escapes contributes nothing; closed room -> never
    bd, _ = o.render(make_config(48, 32, spp=2, cam_depth=2, light_depth=2))
    assert bd.mean() > 0


def test_hdr_loader_rle_and_flat(tmp_path):
    rgbe = random_rgbe(9, 16, seed=4)
    for rle in (True, False):
        p = tmp_path / ("t_%d.hdr" % rle)
        write_hdr(str(p), rgbe, rle=rle)
        cols = B.hdr_load(str(p))
        assert cols.shape == (9, 16, 3)
        assert np.array_equal(cols, decode_rgbe(rgbe)), rle   # rows in file order, val/256 * 2^(E-128)
    with pytest.raises(B.OracleError):
        B.hdr_load(str(tmp_path / "missing.hdr"))


@pytest.mark.skipif(G.reference_env_dir() is None, reason="reference checkout not mounted (GPU box)")
def test_shipped_hdr_faces_checksums():
    for f, want in META["hdr_faces"].items():
        cols = B.hdr_load(G.reference_env_dir() + f + ".hdr")
        assert list(cols.shape) == want["shape"] == [1024, 1024, 3]
        assert hashlib.sha256(cols.tobytes()).hexdigest() == want["sha256"]


def test_obj_reader_swaps_y_z_and_bakes_srt(tmp_path):
    # mesh.cu:232-317 + math.cuh:55: OBJ (x,y,z) enters the transform as (x,z,y)
    p = tmp_path / "tri.obj"
    p.write_text("v 1 2 3\nv 4 5 6\nv 7 8 10\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1/9/1 3/9/1 2/9/1\n")
    o = B.OracleScene()
    o.add_material((1, 1, 1), (0, 0, 0))
    o.add_mesh_obj(str(p), (10, 20, 30), (0, 0, 0), (2, 2, 2), 0)
    o.commit()
    pos, nrm = o.mesh_triangles(0)
    assert np.allclose(pos[0], [[12, 26, 34], [18, 32, 40], [24, 40, 46]])
    assert np.allclose(pos[1], pos[0][[0, 2, 1]])
    assert np.allclose(nrm[0], [[0, 1, 0]] * 3, atol=3e-5)   # normal (0,0,1) -> swapped to (0,1,0)
    with pytest.raises(B.OracleError):
        o.add_mesh_obj(str(tmp_path / "nope.obj"), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)


@pytest.mark.parametrize("key", ["simple_d6_200spp", "bidir_3_3_200spp"])
def test_oracle_against_the_reference_readme_renders(key, cube_obj):
    """The only outputs of the reference itself that exist for this path: its two published 1024x1024, 200-sample renders
    of the default scene (8-bit, tone-clamped, 256-colour palette, cuRAND, the author's cube.obj).  The oracle's own
    200-sample render at 256x256 (a quarter of the resolution: the CPU suite's budget) against their block statistics:
    channel means within 1 %, wall blocks (no cube, hardly a saturated pixel) within 1.5 % in sum.  Measured: means 0.23 % /
    0.2 %, walls 0.23 % / 0.3 %, blocks a cube touches 1.4 % / 1.1 %.  (Round 1, before the point on a ray was evaluated with
    a fused multiply-add like nvcc does: means 2.5 %, left wall 8-14 % too bright in red and blue.)
    The full-resolution comparison runs on the GPU: tests/test_gpu_golden.py."""
    from tests.readme_util import CONFIGS, compare, cube_mask_of
    desc = G.scene_default(cube_obj)
    o = desc.apply(B.OracleScene())
    o.commit()
    img, _ = o.render(make_config(256, 256, spp=200, **CONFIGS[key]))
    img = np.nan_to_num(img, nan=0.0, posinf=0.0, neginf=0.0)
    r = compare(key, img, 256, 256, cube_mask_of(o, desc.camera, 256))
    print(key, r)
    assert r["corr"] > 0.97 and r["mean_dev"] < 0.01, r
    assert r["walls"]["blocks"] > 400 and r["walls"]["sum_dev"] < 0.015 and r["cubes"]["sum_dev"] < 0.03, r
