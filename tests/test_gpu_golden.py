"""GPU: the HIP path reproduces the committed golden fixtures, reads assets like the oracle, and holds parity at
BASELINE.json's full sizes (checked exactly on pixel / ray subsets the oracle can afford)."""
import os

import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd._lib import FLAG_DROP_NONFINITE
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene
from tests.hdr_util import decode_rgbe, random_rgbe, write_hdr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))


def _gpu(desc):
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    return s


@pytest.mark.parametrize("name", ["default", "ico3", "ico4"])
def test_closest_hit_goldens(name, cube_obj):
    desc = {"default": lambda: G.scene_default(cube_obj), "ico3": lambda: G.scene_icosphere_room(3),
            "ico4": lambda: G.scene_icosphere_room(4)}[name]()
    prim, t, _ = _gpu(desc).trace_closest(GOLD[name + "_ray_o"], GOLD[name + "_ray_d"])
    assert np.array_equal(prim, GOLD[name + "_prim"])
    assert np.array_equal(t.view(np.uint32), GOLD[name + "_t"].view(np.uint32))


def test_image_goldens(cube_obj):
    s = _gpu(G.scene_default(cube_obj))
    img, st = s.render(make_config(640, 480, spp=16, cam_depth=6, light_depth=0, seed=G.SEED))   # BASELINE config C1
    assert np.allclose(img.reshape(60, 8, 80, 8, 3).mean(axis=(1, 3)), GOLD["c1_block_means"], rtol=1e-6)
    assert abs(img.mean() - GOLD["c1_stats"][0]) < 1e-6 * GOLD["c1_stats"][0] + 1e-9
    assert st["rays_total"] <= GOLD["c1_stats"][2]
    img, _ = s.render(make_config(128, 96, spp=4, cam_depth=3, light_depth=3, seed=G.SEED))
    assert np.array_equal(img, GOLD["bidir_small"])


def test_env_lookup_goldens_and_rgbe_directory(tmp_path):
    s = _gpu(G.scene_env_sphere(1, env_f32=G.synthetic_env_faces(8)))
    assert np.array_equal(s.env_lookup(GOLD["env_dirs"]), GOLD["env_rgb"])
    # EnvMap(folder): six Radiance files, raw RGBE kept on the device and decoded in-kernel
    faces = {}
    for i, f in enumerate(("px", "py", "pz", "nx", "ny", "nz")):
        faces[f] = random_rgbe(16, 16, seed=20 + i)
        write_hdr(str(tmp_path / (f + ".hdr")), faces[f], rle=(i % 2 == 0))
    desc = G.scene_env_sphere(1, env_dir=str(tmp_path) + "/")
    s = _gpu(desc)
    o = desc.apply(OracleScene())
    o.commit()
    rng = np.random.default_rng(9)
    dirs = rng.standard_normal((2048, 3)).astype(np.float32)
    got = s.env_lookup(dirs)
    assert np.array_equal(got, o.env_lookup(dirs))
    assert np.isin(got.reshape(-1), np.concatenate([decode_rgbe(v).reshape(-1) for v in faces.values()])).all()
    cfg = make_config(48, 32, spp=2, cam_depth=3, use_envmap=True)
    assert np.array_equal(s.render(cfg)[0], o.render(cfg)[0])


def test_reference_and_persistent_kernels_agree():
    desc = G.scene_icosphere_room(5)
    ro, rd = G.incoherent_rays(1 << 16, (-3, -4, -5), (3, 2, 1))
    os.environ["MIPT_EXTEND"] = "simple"
    try:
        a = _gpu(desc).trace_closest(ro, rd, counters=True)
    finally:
        del os.environ["MIPT_EXTEND"]
    b = _gpu(desc).trace_closest(ro, rd, counters=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    for k in ("ctr_nodes", "ctr_pops", "ctr_refs", "ctr_tris", "ctr_hits", "ctr_culled"):
        assert a[2][k] == b[2][k], k
    assert b[2]["ctr_culled"] > 0
    c = _gpu(desc).trace_closest(ro, rd)  # production kernel
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))


def test_full_size_c2_frame_on_pixel_subset():
    # BASELINE config C2 at full resolution (1920x1080, 81,920 triangles), 1 spp: every 997th pixel bit-exact
    desc = G.scene_icosphere_room(6)
    s = _gpu(desc)
    o = desc.apply(OracleScene())
    o.commit()
    cfg = make_config(1920, 1080, spp=1, cam_depth=6)
    img, st = s.render(cfg)
    ref, _ = o.render(cfg, pixel_stride=997)
    sel = np.arange(0, 1920 * 1080, 997)
    assert np.array_equal(img.reshape(-1, 3)[sel], ref.reshape(-1, 3)[sel])
    assert st["samples"] == 1920 * 1080 and np.isfinite(img).all()


def test_full_size_1m_triangle_mesh_on_ray_subset():
    # the 1,310,720-triangle icosphere of C4 / the closest-hit benchmark: 2^20 rays, 2^14 of them checked exactly
    desc = G.scene_env_sphere(8)
    s = _gpu(desc)
    o = desc.apply(OracleScene())
    o.commit()
    ro, rd = G.incoherent_rays(1 << 20, (-2.5,) * 3, (2.5,) * 3)
    prim, t, st = s.trace_closest(ro, rd, counters=True)
    sel = np.arange(0, 1 << 20, 64)
    op, ot, _ = o.trace_closest(ro[sel], rd[sel])
    assert np.array_equal(prim[sel], op) and np.array_equal(t[sel].view(np.uint32), ot.view(np.uint32))
    # the production kernel (incremental descent, cull boxes, candidate spheres) against the counting one, all 2^20 rays
    fprim, ft, _ = s.trace_closest(ro, rd)
    assert np.array_equal(fprim, prim) and np.array_equal(ft.view(np.uint32), t.view(np.uint32))
    hit = prim >= 0
    assert 0.8 < hit.mean() <= 1.0 and (t[hit] > 0).all() and (t[~hit] == -1).all()
    # property: a reported triangle hit really lies on the unit-direction ray at distance t from the sphere centre ~2.5
    tri = (prim >= 0) & (prim < 1310720)
    d = rd[tri] / (np.linalg.norm(rd[tri], axis=1, keepdims=True) + 1e-5)
    r = np.linalg.norm(ro[tri] + d * t[tri, None], axis=1)
    assert np.abs(r - 2.5).max() < 2e-3


def test_full_size_10m_triangle_multi_mesh(monkeypatch):
    # BASELINE config C5's scene: 8 overlapping 1,310,720-triangle icospheres (10,485,760 triangles).  Too large for the
    # oracle inside a test: the production kernel (mesh ordering, distance pruning with its repeat path, cull boxes,
    # candidate spheres) is held against the counting kernel, which walks every mesh in the reference's order node for
    # node and is itself checked against the oracle at the smaller sizes; renders are held against their unbinned twins.
    desc = G.scene_multi_sphere(8)
    s = _gpu(desc)
    assert s.device_bytes() > 2.5e9
    ro, rd = G.incoherent_rays(1 << 18, (-4.6,) * 3, (4.6,) * 3)
    co, cd = G.camera_rays(desc.camera, 512, 512, 1 << 18)
    ro, rd = np.concatenate([ro, co]), np.concatenate([rd, cd])
    prim, t, st = s.trace_closest(ro, rd, counters=True)
    fprim, ft, _ = s.trace_closest(ro, rd)
    assert np.array_equal(fprim, prim) and np.array_equal(ft.view(np.uint32), t.view(np.uint32))
    tri = (prim >= 0) & (prim < 8 * 1310720)
    assert tri.mean() > 0.5 and len(np.unique(prim[tri] // 1310720)) == 8  # every mesh is hit by some ray
    cfgs = (make_config(256, 192, spp=2, cam_depth=5, light_depth=0), make_config(160, 120, spp=2, cam_depth=5, light_depth=5))
    imgs = [s.render(c)[0] for c in cfgs]
    monkeypatch.setenv("MIPT_SHADOW_SORT", "-1")
    plain = _gpu(desc)
    monkeypatch.delenv("MIPT_SHADOW_SORT")
    for c, img in zip(cfgs, imgs):
        assert np.isfinite(img).all() and img.max() > 0.01
        assert np.array_equal(plain.render(c)[0].view(np.uint32), img.view(np.uint32))


@pytest.mark.parametrize("key", ["simple_d6_200spp", "bidir_3_3_200spp"])
def test_readme_configurations_against_the_reference_renders(key, cube_obj):
    # The README's own configurations (1024x1024, 200 samples; BASELINE config C3 is the bidirectional one) through the
    # HIP path, against the block statistics of the reference's published renders (tests/readme_util.py: classes of blocks).
    # north_star's bar -- 1 % -- on everything the 255 clamp and the absent cube.obj do not blur:
    #   channel means            measured 0.05 % / 0.04 %   (round 1: 2.6 % / 2.3 %)
    #   64-pixel block rms       measured 0.84 % / 0.82 %   (round 1: 2.6 % / 3.7 %; ~0.7 % of it is the Monte-Carlo noise of
    #                                                        two independent 200-sample pictures)
    #   wall blocks, in sum      measured 0.15 % / 0.15 %;  blocks a cube touches 1.5 % / 1.1 % (palette, reconstructed cube.obj)
    from tests.readme_util import CONFIGS, compare, cube_mask_of
    desc = G.scene_default(cube_obj)
    s = _gpu(desc)
    cfg = make_config(1024, 1024, spp=200, **CONFIGS[key])
    cfg.flags = FLAG_DROP_NONFINITE   # (a non-finite sample would saturate its pixel: one pixel in 10^6)
    img, st = s.render(cfg)
    r = compare(key, img, 1024, 1024, cube_mask_of(s, desc.camera, 1024))
    print(key, r)
    assert st["samples"] == 1024 * 1024 * 200
    assert r["corr"] > 0.995 and r["rel_rms"] < 0.012 and r["mean_dev"] < 0.005, r
    assert r["walls"]["blocks"] > 800 and r["walls"]["sum_dev"] < 0.01 and r["walls"]["rel_rms"] < 0.025, r
    assert r["cubes"]["sum_dev"] < 0.025 and r["saturated"]["sum_dev"] < 0.005, r
