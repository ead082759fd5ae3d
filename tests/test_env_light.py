"""The cube map as a light (mipt_render_config.use_envmap == 2): a build-defined extension with no counterpart in the
reference (it only looks radiance up, envMap.cuh:50-89), so what is pinned here is (1) that the product's host side and the
oracle build the SAME sampling tables, (2) that a sampled texel's direction looks that texel up again through the
reference's index formulas, (3) that the estimator converges to the picture the miss term alone gives (oracle, CPU), and --
in tests/test_gpu_render.py -- that the HIP path equals the oracle bit for bit with it switched on."""
import struct

import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene


def _product_tables(scene, w, h, rgbe):
    """marg[6 h + 1], cond[6 h, w] out of the packed arena (BlobHeader: 17 section offsets from byte 24, env is the 16th)."""
    blob = scene.export_blob()
    env_off = struct.unpack_from("<Q", blob, 24 + 8 * 15)[0]
    arena = blob[-scene.device_bytes():]
    faces = 6 * w * h * (4 if rgbe else 12)
    marg_off = (faces + 255) & ~255
    cond_off = (marg_off + (6 * h + 1) * 4 + 255) & ~255
    marg = arena[env_off + marg_off:env_off + marg_off + (6 * h + 1) * 4].view(np.float32)
    cond = arena[env_off + cond_off:env_off + cond_off + 6 * h * w * 4].view(np.float32).reshape(6 * h, w)
    return marg, cond


def _env_scene(n, black=False, spike=False):
    desc = G.scene_env_sphere(3)
    faces = G.synthetic_env_faces(n)
    if black:
        faces = np.zeros_like(faces)
    if spike:  # a small sun, some unusable texels: the tables must skip NaN / negative / infinite radiance
        faces[1, n // 4:n // 4 + 2, n // 2:n // 2 + 2] = 4000.0
        faces[0, 0, 0] = np.nan
        faces[2, 1, 1] = -1.0
        faces[3, 2, 2] = np.inf
    desc.env_f32 = faces
    return desc


@pytest.mark.parametrize("n,spike", [(4, False), (16, True), (64, False)])
def test_tables_identical_in_product_and_oracle(n, spike):
    desc = _env_scene(n, spike=spike)
    s = desc.apply(Scene())
    s.commit(devices=())
    o = desc.apply(OracleScene(threads=4))
    o.commit()
    ok, marg, cond = o.env_light_tables(n, n)
    pm, pc = _product_tables(s, n, n, rgbe=False)
    assert ok and pm[-1] == 1.0
    assert np.array_equal(pm[:-1].view(np.uint32), marg.view(np.uint32))
    assert np.array_equal(pc.view(np.uint32), cond.view(np.uint32))
    assert np.all(np.diff(marg) >= 0) and marg[-1] == 1.0 and np.all(np.diff(cond, axis=1) >= 0) and np.all(cond[:, -1] == 1.0)


def test_black_map_is_no_light():
    desc = _env_scene(8, black=True)
    s = desc.apply(Scene())
    s.commit(devices=())
    o = desc.apply(OracleScene(threads=2))
    o.commit()
    ok, _, _ = o.env_light_tables(8, 8)
    pm, _ = _product_tables(s, 8, 8, rgbe=False)
    assert not ok and pm[-1] == 0.0
    cfg = make_config(24, 16, spp=2, cam_depth=3, use_envmap=2)   # falls back to the miss term: same picture as mode 1
    a = o.render(cfg)[0]
    cfg.use_envmap = 1
    assert np.array_equal(a, o.render(cfg)[0])


def test_sampled_texel_round_trip_and_probabilities():
    n = 16
    desc = _env_scene(n, spike=True)
    o = desc.apply(OracleScene(threads=2))
    o.commit()
    _, marg, cond = o.env_light_tables(n, n)
    rng = np.random.default_rng(3)
    total_sa = 0.0
    seen = set()
    for _ in range(3000):
        u1, u2 = rng.uniform(1e-7, 1.0, 2)
        (f, r, c), pmf, sa, d = o.env_light_sample(u1, u2)
        _, face, pos = o.env_lookup(d[None, :], with_index=True)
        assert face[0] == f and pos[0] == (r * n + c) * 3          # the reference's own index formulas find the texel again
        R = f * n + r
        p_row = marg[R] - (marg[R - 1] if R else 0.0)
        p_col = cond[R, c] - (cond[R, c - 1] if c else 0.0)
        assert pmf == np.float32(p_row) * np.float32(p_col) and pmf > 0
        assert np.isfinite(desc.env_f32[f, r, c]).all() and desc.env_f32[f, r, c].sum() > 0   # unusable texels are never drawn
        if (f, r, c) not in seen:
            seen.add((f, r, c))
            total_sa += sa
    assert (1, n // 4, n // 2) in seen                                # the sun is found
    assert total_sa < 4 * np.pi * 1.001                               # solid angles of distinct texels never exceed the sphere


def test_estimator_converges_to_the_miss_term_picture():
    """Unbiasedness, on the CPU oracle: the same scene with the map sampled as a light and with the miss term alone."""
    desc = _env_scene(32)
    desc.materials.append(((0.1, 0.1, 0.1), (0.8, 0.8, 0.8)))                # a mirror ball: the miss term must survive behind it
    desc.spheres.append(((-2.0, 0.5, -1.0), 1.0, len(desc.materials) - 1))
    o = desc.apply(OracleScene(threads=8))
    o.commit()
    imgs = {}
    for mode in (1, 2):
        cfg = make_config(48, 36, spp=500, cam_depth=4, light_depth=0, use_envmap=mode)
        imgs[mode] = o.render(cfg)[0]
    a, b = imgs[1], imgs[2]
    assert abs(a.mean() - b.mean()) / a.mean() < 0.02
    blocks = lambda x: x.reshape(3, 12, 4, 12, 3).mean(axis=(1, 3))
    assert np.abs(blocks(a) - blocks(b)).max() / blocks(a).mean() < 0.08
