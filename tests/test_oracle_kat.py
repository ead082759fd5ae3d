"""Known-answer tests that tie the CPU oracle to the reference's formulas (closed forms worked out by hand from
the reference source; file:line in each test).  These are the pins of an otherwise 'parity unpinned' oracle."""
import math

import numpy as np
import pytest

from oracle import binding as B

EPS = np.float32(1e-5)


def f32(x):
    return np.float32(x)


def test_normalize_adds_epsilon_to_length():
    # math.cuh:28  v * (1 / (|v| + 1e-5)): a (3,0,4) vector has length 5 -> components 3/(5+1e-5), 0, 4/(5+1e-5)
    out = B.normalize([3, 0, 4])
    k = f32(1) / (f32(5) + EPS)
    assert out[0] == f32(3) * k and out[1] == 0 and out[2] == f32(4) * k
    assert np.linalg.norm(out.astype(np.float64)) < 1.0  # Q1: not exactly unit


def test_philox4x32_10_published_vectors():
    # Random123 kat_vectors for philox4x32-10
    assert list(B.philox4x32_10([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(B.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(B.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rng_uniform_in_half_open_unit_interval_and_keyed():
    u = np.array([B.rng_uniform(7, p, s, d) for p in range(4) for s in range(4) for d in range(9)], np.float32)
    assert (u > 0).all() and (u <= 1).all()
    assert len(np.unique(u)) == len(u)
    # draw d of (seed, pixel, sample) = word d&3 of philox(ctr=(d>>2, sample, pixel, 0), key=(seed, 'mipt'))
    w = B.philox4x32_10([1, 5, 3, 0], [7, 0x6d697074])
    assert B.rng_uniform(7, 3, 5, 6) == float(f32((int(w[2]) >> 8) + 1) * f32(1.0 / 16777216.0))


def test_plane_intersect():
    # plane.cuh:18-30: t = n.(p - o) / n.d ; reject |n.d| < eps and t < eps
    assert B.plane_intersect([0, -5, 0], [0, 1, 0], [0, 0, 0], [0, -1, 0]) == 5.0
    assert B.plane_intersect([0, -5, 0], [0, 1, 0], [0, 0, 0], [0, 1, 0]) == -1.0       # behind
    assert B.plane_intersect([0, -5, 0], [0, 1, 0], [0, 0, 0], [1, 0, 0]) == -1.0       # parallel
    assert B.plane_intersect([0, -5, 0], [0, 1, 0], [0, 0, 0], [1, 5e-6, 0]) == -1.0    # |n.d| < eps
    assert B.plane_intersect([0, 0, 0], [0, 1, 0], [0, 5e-6, 0], [0, -1, 0]) == -1.0    # t < eps
    d = np.float32(1 / math.sqrt(2))
    assert B.plane_intersect([0, -5, 0], [0, 1, 0], [0, 0, 0], [d, -d, 0]) == float(f32(-5) / (-d))


def test_sphere_intersect():
    # sphere.cuh:19-37: nearer positive root, inside -> far root, a = d.d kept
    assert B.sphere_intersect([0, 0, -5], 1, [0, 0, 0], [0, 0, -1]) == 4.0
    assert B.sphere_intersect([0, 0, 0], 2, [0, 0, 0], [0, 0, -1]) == 2.0     # origin inside: t2 < 0 -> t1
    assert B.sphere_intersect([0, 0, 5], 1, [0, 0, 0], [0, 0, -1]) == -1.0    # behind
    assert B.sphere_intersect([0, 3, -5], 1, [0, 0, 0], [0, 0, -1]) == -1.0   # miss (discr < 0)
    assert B.sphere_intersect([0, 0, -5], 1, [0, 0, 0], [0, 0, -2]) == 2.0    # un-normalised d: t scales by 1/|d|


def test_triangle_intersect_quirks():
    tri = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    # mesh.cuh:36-90: plane hit then three edge tests with the un-normalised N = (0,0,1)
    assert B.triangle_intersect(tri, [0.25, 0.25, 1], [0, 0, -1]) == 1.0
    assert B.triangle_intersect(tri, [0.25, 0.25, -1], [0, 0, 1]) == 1.0          # two-sided
    assert B.triangle_intersect(tri, [2, 2, 1], [0, 0, -1]) == -1.0               # outside -> -1
    assert B.triangle_intersect(tri, [0.25, 0.25, 1], [0, 0, 1]) == -1.0          # behind: raw negative t (Q4)
    assert B.triangle_intersect(tri, [0.25, 0.25, 0], [0, 0, -1]) == 0.0          # t == 0 is returned (caller needs t > 0)
    assert B.triangle_intersect(tri, [0.25, 0.25, 1], [1, 0, -5e-6]) == -1.0      # |N.d| < 1e-5 -> parallel (Q3)
    # Q3: absolute epsilon on the UN-normalised normal: a tiny triangle (|N| = 2e-6) is invisible even head-on
    tiny = [0, 0, 0, 1e-3, 0, 0, 0, 2e-3, 0]
    assert B.triangle_intersect(tiny, [2e-4, 2e-4, 1], [0, 0, -1]) == -1.0
    # edge-on hit exactly on an edge is accepted (dot(N, C) == 0 is not < 0)
    assert B.triangle_intersect(tri, [0.5, 0, 1], [0, 0, -1]) == 1.0


def test_box_contains_asymmetric_epsilon():
    # mesh.cuh:131-139: minX has no slack, every other bound has +1e-5 (Q5)
    box = [0, 1, 0, 1, 0, 1]
    assert not B.box_contains(box, [-5e-6, 0.5, 0.5])
    assert B.box_contains(box, [0.5, -5e-6, 0.5])
    assert B.box_contains(box, [0.5, 0.5, -5e-6])
    assert B.box_contains(box, [1 + 5e-6, 0.5, 0.5])
    assert B.box_contains(box, [0.5, 1 + 5e-6, 1 + 5e-6])
    assert not B.box_contains(box, [0.5, 1 + 3e-5, 0.5])


def test_box_intersect():
    # mesh.cuh:148-170: smallest positive plane parameter whose point is inside the box
    box = [-1, 1, -1, 1, -1, 1]
    assert B.box_intersect(box, [0, 0, 3], [0, 0, -1]) == 2.0          # enters through z = 1
    assert B.box_intersect(box, [0, 0, 0], [0, 0, -1]) == 1.0          # origin inside: EXIT distance
    assert B.box_intersect(box, [0, 0, 3], [0, 0, 1]) == -1.0          # pointing away
    assert B.box_intersect(box, [3, 3, 3], [0, 0, -1]) == -1.0         # passes beside the box
    d = B.normalize([1, 1, 1])
    assert B.box_intersect(box, [-3, -3, -3], d) == pytest.approx(2 / float(d[0]), rel=1e-6)


def test_camera_ray_formula():
    # camera.cuh:9-20 with runtime W,H; fov 90 deg, eye at z=2 looking at the origin: right=(2,0,0), up=(0,2,0) up to Q1
    s = B.OracleScene()
    s.set_camera([0, 0, 2], [0, 0, 0], [0, 1, 0], math.pi / 2)
    o, d = s.camera_ray(50, 50, 100, 100)   # image centre -> straight ahead
    assert np.allclose(o, [0, 0, 2]) and np.allclose(d, [0, 0, -1], atol=2e-5)
    o, d = s.camera_ray(100, 50, 100, 100)  # right edge: dir = right - (0,0,2) normalised
    assert np.allclose(d, np.array([2, 0, -2]) / math.sqrt(8), atol=3e-5)
    o, d = s.camera_ray(50, 100, 100, 100)  # larger Y is up (row r <-> Y = r)
    assert d[1] > 0.7


def test_fixed_sincos_accuracy():
    import ctypes as C
    lib = B.load()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(1e-6, 2 * math.pi, 20001):
        lib.orc_sincos(C.c_float(float(np.float32(x))), C.byref(s), C.byref(c))
        xf = float(np.float32(x))
        worst = max(worst, abs(s.value - math.sin(xf)), abs(c.value - math.cos(xf)))
    assert worst < 1.5e-7


def test_srt_matrix_matches_reference_conventions():
    # math.cuh:127-158 row-vector convention: translation in the last ROW; rotation about y by +90 deg maps the row
    # vector (1,0,0) to (0,0,-1)
    m, it = B.srt_matrix([1, 1, 1], [0, math.pi / 2, 0], [5, 6, 7])
    assert np.allclose(m[3, :3], [5, 6, 7])
    assert np.allclose(np.array([1, 0, 0, 0], np.float32) @ m, [0, 0, -1, 0], atol=1e-6)
    # math.cu:24-26 lays cross(b,c)/det, ... out as ROWS, which for a pure rotation reproduces the matrix itself rather
    # than its inverse; .transpose() then yields R^T, so normals of a rotated mesh turn the opposite way (a reference
    # quirk the oracle keeps; invisible in the default scene, whose only rotation is by pi).
    assert np.allclose(it[:3, :3], m[:3, :3].T, atol=5e-5)
