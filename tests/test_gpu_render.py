"""GPU parity: wavefront renderer (mipt_render through the C-ABI) vs the CPU oracle on identical seeds.
Simple path tracing is pure +,-,*,/,sqrt in fp32 on both sides -> the images must be bit-identical.
The bidirectional integrator's sinf/cosf (light.cuh:35-38) are evaluated by one fixed fp32 polynomial on
both sides, so those images are bit-identical too."""
import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene

pytestmark = pytest.mark.gpu


def _pair(desc):
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    o = desc.apply(OracleScene())
    o.commit()
    return s, o


def _rel_rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(1e-12, np.sqrt(np.mean(b ** 2))))


def test_simple_default_scene_bit_exact(cube_obj):
    s, o = _pair(G.scene_default(cube_obj))
    cfg = make_config(96, 64, spp=4, cam_depth=6, light_depth=0)
    img, st = s.render(cfg)
    ref, ctr = o.render(cfg)
    assert ref.max() > 0.01
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"max abs diff {np.abs(img - ref).max()}"
    assert st["rays_total"] == ctr["rays"] or st["rays_total"] <= ctr["rays"]  # GPU skips shadow rays that cannot contribute
    assert st["samples"] == 96 * 64 * 4


def test_simple_icosphere_bit_exact_and_pass_split():
    s, o = _pair(G.scene_icosphere_room(4))
    cfg = make_config(80, 60, spp=5, cam_depth=6, light_depth=0)
    ref, _ = o.render(cfg)
    for spp_pass in (0, 1, 2):
        cfg.samples_per_pass = spp_pass
        img, _ = s.render(cfg)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (spp_pass, np.abs(img - ref).max())


def test_sample_partition_sums_to_full_image():
    s, o = _pair(G.scene_icosphere_room(3))
    full = make_config(64, 48, spp=8, cam_depth=4)
    ref, _ = o.render(full)
    parts = []
    for g in range(2):
        c = make_config(64, 48, spp=8, cam_depth=4, sample_begin=g, sample_end=8, sample_stride=2)
        img, _ = s.render(c)
        oimg, _ = o.render(c)
        assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32))
        parts.append(img)
    assert np.allclose(parts[0] + parts[1], ref, rtol=1e-5, atol=1e-7)


def test_bidirectional_bit_exact(cube_obj):
    s, o = _pair(G.scene_default(cube_obj))
    for dc, dl in ((3, 3), (2, 5), (5, 1)):
        cfg = make_config(64, 48, spp=4, cam_depth=dc, light_depth=dl)
        img, st = s.render(cfg)
        ref, _ = o.render(cfg)
        assert st["rays_light"] > 0 and st["rays_shadow"] > 0
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (dc, dl, np.abs(img - ref).max())


def test_bidirectional_mesh_and_mirror():
    desc = G.scene_icosphere_room(3)
    desc.materials.append(((0.1, 0.1, 0.1), (0.8, 0.8, 0.8)))
    desc.spheres.append(((-2.5, -3.5, -1.0), 1.2, len(desc.materials) - 1))
    s, o = _pair(desc)
    cfg = make_config(64, 48, spp=3, cam_depth=4, light_depth=4)
    img, _ = s.render(cfg)
    ref, _ = o.render(cfg)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), np.abs(img - ref).max()


def test_mirror_sphere_and_envmap():
    desc = G.scene_env_sphere(level=3, env_f32=G.synthetic_env_faces(16))
    desc.spheres.append(((3.0, -3.0, 0.0), 1.5, 1))
    s, o = _pair(desc)
    cfg = make_config(64, 48, spp=3, cam_depth=5, light_depth=0, use_envmap=True)
    img, _ = s.render(cfg)
    ref, _ = o.render(cfg)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), np.abs(img - ref).max()


@pytest.mark.parametrize("rgbe", [False, True])
def test_env_map_as_a_light_bit_exact(rgbe, tmp_path):
    """use_envmap == 2 (build-defined, no counterpart in the reference): texels drawn in proportion to radiance x solid angle,
    one more shadow ray per camera vertex, the miss term only at vertex 0 and behind mirror bounces.  Simple and bidirectional,
    one and several passes, a one-vertex path (the sample is taken at the LAST vertex too), a mirror ball, a mesh."""
    from tests.hdr_util import random_rgbe, write_hdr
    if rgbe:
        for i, f in enumerate(("px", "py", "pz", "nx", "ny", "nz")):
            write_hdr(str(tmp_path / (f + ".hdr")), random_rgbe(16, 16, seed=40 + i), rle=(i % 2 == 0))
        desc = G.scene_env_sphere(level=3, env_dir=str(tmp_path) + "/")
    else:
        faces = G.synthetic_env_faces(16)
        faces[1, 3:5, 8:10] = 900.0   # a sun
        desc = G.scene_env_sphere(level=3, env_f32=faces)
    desc.spheres.append(((3.0, -3.0, 0.0), 1.5, 1))
    s, o = _pair(desc)
    for dc, dl, spp_pass in ((4, 0, 0), (1, 0, 0), (5, 0, 2), (3, 2, 0), (2, 4, 1)):
        cfg = make_config(64, 48, spp=3, cam_depth=dc, light_depth=dl, use_envmap=2, samples_per_pass=spp_pass)
        img, st = s.render(cfg)
        ref, _ = o.render(cfg)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (dc, dl, spp_pass, np.abs(img - ref).max())
        cfg.use_envmap = 1
        assert not np.array_equal(ref, o.render(cfg)[0])       # the mode does change the picture ...
    cfg = make_config(64, 48, spp=3, cam_depth=4, light_depth=0, use_envmap=1)
    assert np.array_equal(s.render(cfg)[0].view(np.uint32), o.render(cfg)[0].view(np.uint32))   # ... and mode 1 is what it was


def test_env_lookup_matches_oracle():
    desc = G.scene_env_sphere(level=1, env_f32=G.synthetic_env_faces(8))
    s, o = _pair(desc)
    rng = np.random.default_rng(3)
    dirs = rng.standard_normal((4096, 3)).astype(np.float32)
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    edges = np.array([[1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0], [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1],
                      [0, 1, 1], [0, 1, -1], [0, -1, 1], [0, -1, -1]], np.float32)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    dirs = np.concatenate([dirs, corners, edges, axes])
    assert np.array_equal(s.env_lookup(dirs), o.env_lookup(dirs))


def test_overlapping_meshes_simple_and_bidirectional():
    # the C5 layout at a small subdivision level: eight mutually overlapping kd meshes tested linearly (kernel.cu:147-150)
    s, o = _pair(G.scene_multi_sphere(3))
    for kw in (dict(cam_depth=5, light_depth=0), dict(cam_depth=3, light_depth=3)):
        cfg = make_config(72, 48, spp=3, **kw)
        img, _ = s.render(cfg)
        ref, _ = o.render(cfg)
        assert ref.max() > 0.01
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (kw, np.abs(img - ref).max())


def test_deep_paths_many_passes():
    # depth 16 (the accepted maximum) and a pass size that does not divide the sample count
    s, o = _pair(G.scene_icosphere_room(3))
    cfg = make_config(40, 30, spp=7, cam_depth=16, light_depth=0, samples_per_pass=3)
    img, st = s.render(cfg)
    ref, _ = o.render(cfg)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert st["samples"] == 40 * 30 * 7


def test_scene_without_meshes():
    # planes and spheres only: every ray is decided by the stage that generates it, the extend kernel gets empty queues
    desc = G.scene_icosphere_room(2)
    desc.meshes.clear()
    desc.materials.append(((0.1, 0.1, 0.1), (0.8, 0.8, 0.8)))
    desc.spheres.append(((-1.0, -3.0, -2.0), 1.5, len(desc.materials) - 1))
    desc.spheres.append(((2.0, -4.0, -3.0), 1.0, 0))
    s, o = _pair(desc)
    for kw in (dict(cam_depth=6, light_depth=0), dict(cam_depth=3, light_depth=3)):
        cfg = make_config(64, 48, spp=3, **kw)
        img, st = s.render(cfg)
        ref, _ = o.render(cfg)
        assert ref.max() > 0.01
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (kw, np.abs(img - ref).max())
        assert st["rays_extended"] == 0 and st["rays_total"] > 0
    ro, rd = G.incoherent_rays(4096, (-3, -4, -4), (3, 0, 0))
    gp, gt, _ = s.trace_closest(ro, rd)
    op, ot, _ = o.trace_closest(ro, rd)
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))


def test_render_argument_errors():
    from cuda_path_tracer_amd._lib import MiptError
    desc = G.scene_icosphere_room(2)
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    with pytest.raises(MiptError, match="path depth"):
        s.render(make_config(16, 16, spp=1, cam_depth=17))
    with pytest.raises(MiptError, match="spp"):
        s.render(make_config(16, 16, spp=0))
    with pytest.raises(MiptError, match="image size"):
        s.render(make_config(0, 16, spp=1))
    with pytest.raises(MiptError, match="environment map"):
        s.render(make_config(16, 16, spp=1, use_envmap=True))
    with pytest.raises(MiptError, match="use_envmap must be"):
        s.render(make_config(16, 16, spp=1, use_envmap=3))
    nolight = G.scene_icosphere_room(2)
    nolight.lights.clear()
    s2 = nolight.apply(Scene())
    s2.commit(devices=(0,))
    with pytest.raises(MiptError, match="light"):
        s2.render(make_config(16, 16, spp=1))
    # an empty sample range renders nothing and is not an error
    img, st = s.render(make_config(16, 16, spp=4, sample_begin=2, sample_end=2))
    assert st["samples"] == 0 and not img.any()


def test_two_device_slots_in_one_process():
    # mipt_render over several committed device slots (one host thread and stream per slot, samples s mod G, partial
    # framebuffers combined on the devices: one RCCL reduce for distinct devices, a kernel for slots that share one).
    # With one GPU in the box both slots live on device 0, which still runs the whole multi-slot code path.
    desc = G.scene_icosphere_room(3)
    two = desc.apply(Scene())
    two.commit(devices=(0, 0))
    one = desc.apply(Scene())
    one.commit(devices=(0,))
    cfg = make_config(64, 48, spp=6, cam_depth=5)
    img2, st2 = two.render(cfg)
    parts = []
    for g in range(2):
        c = make_config(64, 48, spp=6, cam_depth=5, sample_begin=g, sample_end=6, sample_stride=2)
        parts.append(one.render(c)[0])
    assert np.array_equal(img2.view(np.uint32), (parts[0] + parts[1]).view(np.uint32))
    full, st1 = one.render(cfg)
    assert np.allclose(img2, full, rtol=2e-6, atol=1e-7)  # same samples, different fp32 summation order
    assert st2["samples"] == st1["samples"] == 64 * 48 * 6 and st2["rays_total"] == st1["rays_total"]


def test_coincident_meshes_render_bit_exact():
    """Exactly tying hits on two identical meshes + a third overlapping one: simple and bidirectional images."""
    d = G.scene_icosphere_room(3)
    kind, pos, nrm, mat = d.meshes[0]
    d.meshes.append((kind, pos + np.array([1.5, 0.25, 0.5], np.float32), nrm, mat))
    d.meshes.append((kind, pos.copy(), nrm, mat))
    s, o = _pair(d)
    for cfg in (make_config(64, 48, spp=3, cam_depth=5, light_depth=0), make_config(48, 40, spp=2, cam_depth=3, light_depth=3)):
        img, _ = s.render(cfg)
        ref, _ = o.render(cfg)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_ray_binning_changes_no_result(monkeypatch):
    """Shadow / first light rays are binned by direction before the closest-hit kernel (scenes >= 4096 triangles;
    MIPT_SHADOW_SORT=-1, read at commit, switches it off): same images bit for bit, same ray counts, equal to the oracle."""
    desc = G.scene_icosphere_room(5)  # 20,480 triangles
    binned = desc.apply(Scene())
    binned.commit(devices=(0,))
    monkeypatch.setenv("MIPT_SHADOW_SORT", "-1")
    plain = desc.apply(Scene())
    plain.commit(devices=(0,))
    monkeypatch.delenv("MIPT_SHADOW_SORT")
    o = desc.apply(OracleScene())
    o.commit()
    for cfg in (make_config(96, 64, spp=4, cam_depth=6, light_depth=0), make_config(64, 48, spp=3, cam_depth=4, light_depth=3)):
        a, sa = binned.render(cfg)
        b, sb = plain.render(cfg)
        ref, _ = o.render(cfg)
        assert a.max() > 0.01
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(a.view(np.uint32), ref.view(np.uint32))
        for k in ("rays_total", "rays_shadow", "rays_camera", "rays_light", "rays_extended", "samples"):
            assert sa[k] == sb[k], k


def test_binned_queues_follow_the_scene_from_render_to_render():
    """The binning stage takes every queue length from device memory (rounds 1-2 sized a library sort by the previous
    render's lengths).  Moving the camera between renders of one scene object makes the queues shrink and grow by large
    factors; every image equals the one a fresh scene renders."""
    desc = G.scene_icosphere_room(5)
    near = ((0.0, -1.0, 1.5), (0.0, -1.0, -2.0), (0.0, 1.0, 0.0), desc.camera[3])   # the sphere fills the frame
    away = ((-3.5, -4.0, -3.5), (-5.0, -5.0, -5.0), (0.0, 1.0, 0.0), desc.camera[3])  # a floor corner on the light's side, sphere behind
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    for cfg in (make_config(512, 384, spp=4, cam_depth=5, light_depth=0), make_config(320, 240, spp=3, cam_depth=3, light_depth=3)):
        extended = []
        for cam in (near, away, near, desc.camera, away):
            s.set_camera(*cam)
            img, st = s.render(cfg)
            fresh = desc.apply(Scene())
            fresh.set_camera(*cam)
            fresh.commit(devices=(0,))
            ref, st_ref = fresh.render(cfg)
            fresh.close()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), cam
            assert st["rays_extended"] == st_ref["rays_extended"] and st["rays_total"] == st_ref["rays_total"]
            extended.append(st["rays_extended"])
        assert max(extended) > 1.2 * min(extended), extended


def test_full_light_queue_binned_at_a_small_pass_size():
    """Round 2's library sort over-ran the first-light-ray key buffer when that queue was (nearly) full and the pass small
    (advisor finding): a light INSIDE a mesh of >= 4096 triangles queues every light ray, 64x64 at 1 spp is a 4096-path
    pass.  Rendered twice in a row (the second render is the one that over-ran) and held against the oracle."""
    desc = G.scene_icosphere_room(5)
    kind, pos, nrm, mat = desc.meshes[0]
    centre = np.asarray(desc.lights[0][0], np.float32)
    big, big_n = G.icosphere(5, 3.0, tuple(float(c) for c in centre))  # 20,480 triangles around lights[0]
    desc.meshes.append(("arrays", big, big_n, mat))
    s, o = _pair(desc)
    for cfg in (make_config(64, 64, spp=1, cam_depth=3, light_depth=2), make_config(64, 64, spp=2, cam_depth=2, light_depth=3)):
        ref, _ = o.render(cfg)
        for _ in range(2):
            img, st = s.render(cfg)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            assert st["rays_light"] >= 64 * 64 * cfg.spp


def test_bench_workload_invariants_at_full_resolution(monkeypatch):
    """C2 as bench.py runs it (1920x1080, icosphere L6, simple depth 6), too large for the oracle: size-independent
    properties instead.  The image does not depend on how the samples are split into wavefront passes, on ray binning,
    or on which half of the samples a rank renders (sample interleave sums to the whole up to fp32 addition order)."""
    desc = G.scene_icosphere_room(6)
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    W, H, SPP = 1920, 1080, 4
    full, st = s.render(make_config(W, H, spp=SPP, cam_depth=6))
    assert np.isfinite(full).all() and full.mean() > 0.5
    assert st["samples"] == W * H * SPP and st["rays_extended"] < st["rays_total"]
    one_per_pass, _ = s.render(make_config(W, H, spp=SPP, cam_depth=6, samples_per_pass=1))
    assert np.array_equal(full.view(np.uint32), one_per_pass.view(np.uint32))
    halves = [s.render(make_config(W, H, spp=SPP, cam_depth=6, sample_begin=g, sample_end=SPP, sample_stride=2))[0] for g in range(2)]
    assert np.allclose(halves[0] + halves[1], full, rtol=1e-5, atol=1e-6)
    monkeypatch.setenv("MIPT_SHADOW_SORT", "-1")
    plain = desc.apply(Scene())
    plain.commit(devices=(0,))
    monkeypatch.delenv("MIPT_SHADOW_SORT")
    unbinned, st2 = plain.render(make_config(W, H, spp=SPP, cam_depth=6))
    assert np.array_equal(full.view(np.uint32), unbinned.view(np.uint32))
    assert st2["rays_total"] == st["rays_total"] and st2["rays_extended"] == st["rays_extended"]
