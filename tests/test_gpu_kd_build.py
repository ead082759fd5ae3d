"""GPU: the kd-tree built on the device (csrc/kd_build_gpu.hip: Node::makeNodeTree, mesh.cu:32-136, level by level) is
the tree of the host builder -- which tests/test_packer.py holds equal to the oracle's, i.e. the reference's -- bit for
bit: same nodes, boxes, split planes, leaf lists, and the same packed arena, hence identical renders."""
import time

import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene

from tests.test_packer import _walk

pytestmark = pytest.mark.gpu
KEYS = ("boxes", "axis", "left", "right", "first", "count", "refs")


def _commit(desc, monkeypatch, builder):
    if builder == "host":
        monkeypatch.setenv("MIPT_KD_BUILD", "host")
    else:
        monkeypatch.delenv("MIPT_KD_BUILD", raising=False)
    s = desc.apply(Scene())
    t0 = time.perf_counter()
    s.commit(devices=(0,))
    return s, time.perf_counter() - t0


@pytest.mark.parametrize("level,shape", [(5, "sphere"), (7, "sphere"), (6, "squashed"), (8, "sphere")])
def test_gpu_tree_equals_host_tree(level, shape, monkeypatch):
    desc = G.scene_icosphere_room(level)
    if shape == "squashed":  # anisotropic + off-centre: uneven splits, the median walk takes its long branches
        kind, pos, nrm, mat = desc.meshes[0]
        desc.meshes[0] = (kind, (pos * np.array([1.0, 0.05, 3.0], np.float32) + np.array([0.5, 2.0, -1.0], np.float32)).astype(np.float32), nrm, mat)
    g, tg = _commit(desc, monkeypatch, "gpu")
    h, th = _commit(desc, monkeypatch, "host")
    assert g.gpu_built_meshes() == 1 and h.gpu_built_meshes() == 0
    kg, kh = g.kd_export(0), h.kd_export(0)
    for k in KEYS:
        assert kg[k].shape == kh[k].shape and np.array_equal(kg[k].view(np.uint32) if kg[k].dtype == np.float32 else kg[k],
                                                             kh[k].view(np.uint32) if kh[k].dtype == np.float32 else kh[k]), k
    n = g.device_bytes()  # the packed arena = the tail of the exported blob (its header carries the build's wall time)
    assert n == h.device_bytes() and np.array_equal(g.export_blob()[-n:], h.export_blob()[-n:])  # byte for byte
    print("\nlevel %d %s: %d triangles, %d nodes; commit (build + pack + upload) %.3f s with the GPU build (kd-tree %.3f s), %.3f s on the host threads (kd-tree %.3f s)"
          % (level, shape, kg["n_tris"], len(kg["axis"]), tg, g.kd_seconds(), th, h.kd_seconds()))


def test_gpu_built_multi_mesh_scene_renders_like_the_oracle(monkeypatch):
    # three meshes: two above the GPU threshold (16,384 triangles), one below (built on the host): mixed builders in one scene
    desc = G.scene_icosphere_room(5)
    kind, pos, nrm, mat = desc.meshes[0]
    p6, n6 = G.icosphere(6, 1.2, (1.5, 1.0, -1.0))
    p2, n2 = G.icosphere(2, 0.8, (-2.0, 1.5, 0.0))
    desc.meshes += [(kind, p6, n6, mat), (kind, p2, n2, mat)]
    s, _ = _commit(desc, monkeypatch, "gpu")
    assert s.gpu_built_meshes() == 2
    o = desc.apply(OracleScene())
    o.commit()
    for m in range(3):  # against the ORACLE's builder: same boxes (bitwise), axes, leaf membership and leaf order
        wg, wo = [], []
        _walk(s.kd_export(m), 0, wg)
        _walk(o.kd_export(m), 0, wo)
        assert wg == wo, m
    cfg = make_config(96, 64, spp=2, cam_depth=4, light_depth=3)
    assert np.array_equal(s.render(cfg)[0].view(np.uint32), o.render(cfg)[0].view(np.uint32))
    ro, rd = G.incoherent_rays(1 << 14, (-3, -3.5, -4.5), (3, 2.5, 0.5))
    gp, gt, _ = s.trace_closest(ro, rd)
    op, ot, _ = o.trace_closest(ro, rd)
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))


# ---- the packer on the GPU (csrc/pack_gpu.hip): the same arena, byte for byte, as the host packer (csrc/scene.cpp) ----------
def _commit_pack(desc, monkeypatch, where, recheck=False):
    monkeypatch.delenv("MIPT_KD_BUILD", raising=False)
    if where == "host":
        monkeypatch.setenv("MIPT_PACK", "host")
    else:
        monkeypatch.delenv("MIPT_PACK", raising=False)
    if recheck:
        monkeypatch.setenv("MIPT_FORCE_RECHECK", "1")
    else:
        monkeypatch.delenv("MIPT_FORCE_RECHECK", raising=False)
    s = desc.apply(Scene())
    t0 = time.perf_counter()
    s.commit(devices=(0,))
    return s, time.perf_counter() - t0


def _mixed_scene():
    """Two meshes above the GPU threshold, one far below it, a sphere, a synthetic cube map: every section of the arena."""
    desc = G.scene_icosphere_room(6)
    kind, pos, nrm, mat = desc.meshes[0]
    squashed = (pos * np.array([1.0, 0.05, 3.0], np.float32) + np.array([0.5, 2.0, -1.0], np.float32)).astype(np.float32)
    p2, n2 = G.icosphere(2, 0.8, (-2.0, 1.5, 0.0))
    desc.meshes += [(kind, squashed, nrm, mat), (kind, p2, n2, mat)]
    desc.spheres.append(((1.0, 2.0, -1.0), 0.5, 0))
    desc.env_f32 = G.synthetic_env_faces(8)
    return desc


@pytest.mark.parametrize("which,recheck", [("L7", False), ("L8", False), ("mixed", False), ("mixed", True)])
def test_gpu_packed_arena_equals_host_packed(which, recheck, monkeypatch):
    """Leaf planes, leaf vertices, cull boxes (bottom-up union), step records, candidate spheres (double arithmetic), padded
    triangle arrays, global nodes with their recheck flags: made on the device, equal to the host packer's bytes."""
    desc = _mixed_scene() if which == "mixed" else G.scene_icosphere_room(int(which[1]))
    g, tg = _commit_pack(desc, monkeypatch, "gpu", recheck)
    h, th = _commit_pack(desc, monkeypatch, "host", recheck)
    assert g.packed_on_gpu() and not h.packed_on_gpu()
    assert g.gpu_built_meshes() == h.gpu_built_meshes() >= 1
    n = g.device_bytes()
    a, b = g.export_blob()[-n:], h.export_blob()[-n:]
    if not np.array_equal(a, b):
        bad = np.flatnonzero(a != b)
        raise AssertionError("%d of %d arena bytes differ, first at offset %d" % (bad.size, n, bad[0]))
    print("\n%s: arena %.1f MB; commit %.3f s with the GPU packer (kd %.3f s + pack %.3f s), %.3f s with the host packer (kd %.3f s + pack %.3f s)"
          % (which, n / 1e6, tg, g.kd_seconds(), g.build_seconds() - g.kd_seconds(), th, h.kd_seconds(), h.build_seconds() - h.kd_seconds()))


def test_gpu_packed_scene_survives_recommit_second_slot_and_export(monkeypatch):
    """A GPU-packed scene keeps no host copy of its arena: a second commit, a second device slot and an export each have to
    fetch it from the device first; renders stay bit-identical throughout and equal the oracle's."""
    desc = _mixed_scene()
    s, _ = _commit_pack(desc, monkeypatch, "gpu")
    assert s.packed_on_gpu()
    o = desc.apply(OracleScene())
    o.commit()
    cfg = make_config(80, 60, spp=2, cam_depth=4, light_depth=2, use_envmap=True)
    ref = o.render(cfg)[0]
    first = s.render(cfg)[0]
    assert np.array_equal(first.view(np.uint32), ref.view(np.uint32))
    s.commit(devices=(0,))                     # packed already: the arena comes back from the device, then goes up again
    assert np.array_equal(s.render(cfg)[0].view(np.uint32), ref.view(np.uint32))
    s.commit(devices=(0, 0))                   # two slots: the second one needs the host copy
    two = s.render(cfg)[0]
    assert np.allclose(two, ref, rtol=2e-6, atol=1e-7)  # (two slots split the samples: fp32 summation order)
    t = Scene()
    t.import_blob(s.export_blob())
    t.commit(devices=(0,))
    assert np.array_equal(t.render(cfg)[0].view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(t.kd_export(1)["refs"], s.kd_export(1)["refs"])
