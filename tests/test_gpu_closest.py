"""GPU parity: kd-tree closest hit (mipt_trace_closest through the C-ABI) vs the CPU oracle.
Bar (BASELINE.json north_star): primitive id bit-exact, t within 1e-5 relative (here: bit-exact)."""
import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene
from oracle.binding import OracleScene

pytestmark = pytest.mark.gpu


def _pair(desc):
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    o = desc.apply(OracleScene())
    o.commit()
    return s, o


def _rays(desc, n):
    o1, d1 = G.camera_rays(desc.camera, 256, 256, n // 2)
    o2, d2 = G.incoherent_rays(n - n // 2, (-3, -3.5, -4.5), (3, 1.5, 0.5))
    return np.concatenate([o1, o2]), np.concatenate([d1, d2])


def _check(desc, n):
    s, o = _pair(desc)
    ro, rd = _rays(desc, n)
    gp, gt, st = s.trace_closest(ro, rd, counters=True)
    op, ot, ctr = o.trace_closest(ro, rd)
    mism = np.flatnonzero(gp != op)
    assert mism.size == 0, f"{mism.size} prim mismatches, first {mism[:5]}: gpu {gp[mism[:5]]} oracle {op[mism[:5]]}"
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32)), "t differs bitwise"
    for k in ("nodes", "pops", "refs", "tris", "hits"):
        assert st["ctr_" + k] == ctr[k], (k, st["ctr_" + k], ctr[k])
    assert (gp >= 0).mean() > 0.5
    return s, o


def test_default_scene(cube_obj):
    _check(G.scene_default(cube_obj), 8192)


def test_icosphere_l4():
    _check(G.scene_icosphere_room(4), 8192)


def test_icosphere_l6():
    _check(G.scene_icosphere_room(6), 65536)


def test_axis_aligned_and_degenerate_rays():
    desc = G.scene_icosphere_room(4)
    s, o = _pair(desc)
    ro = np.array([[0, -1, 3], [0, -1, 3], [0, 4, -2], [-4, -1, -2], [0, -1, -2], [0, -1, -2], [0, 0, 0]], np.float32)
    rd = np.array([[0, 0, -1], [0, 0, 1], [0, -1, 0], [1, 0, 0], [0, 1, 0], [1, 1, 1], [0, 0, 0]], np.float32)
    gp, gt, _ = s.trace_closest(ro, rd)
    op, ot, _ = o.trace_closest(ro, rd)
    assert np.array_equal(gp, op)
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32))


def test_empty_batch():
    s, _ = _pair(G.scene_icosphere_room(2))
    p, t, _ = s.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert p.shape == (0,) and t.shape == (0,)
