"""CPU: the product's host side (kd builder, packer, OBJ reader, C-ABI surface) against the oracle -- no GPU needed
(mipt_scene_commit with zero devices packs on the host only; no compute entry point is called here)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from cuda_path_tracer_amd import _lib, generators as G
from cuda_path_tracer_amd.scene import Scene, make_config
from oracle.binding import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.setrecursionlimit(100000)


def _walk(k, i, out):
    if k["left"][i] < 0:
        out.append((tuple(k["boxes"][i]), tuple(k["refs"][k["first"][i]:k["first"][i] + k["count"][i]])))
    else:
        out.append((tuple(k["boxes"][i]), int(k["axis"][i])))
        _walk(k, k["right"][i], out)
        _walk(k, k["left"][i], out)


def _pair(desc):
    s = desc.apply(Scene())
    s.commit(devices=())
    o = desc.apply(OracleScene())
    o.commit()
    return s, o


@pytest.mark.parametrize("level", [2, 4, 6])
def test_kd_tree_identical_to_oracle(level):
    s, o = _pair(G.scene_icosphere_room(level))
    a, b = s.kd_export(0), o.kd_export(0)
    wa, wb = [], []
    _walk(a, 0, wa)
    _walk(b, 0, wb)
    assert wa == wb                      # same boxes (bitwise), axes, leaf membership and leaf order
    assert len(a["refs"]) == len(b["refs"])
    # packed layout: children adjacent (right = left + 1) at even indices, padding slot 1 unused
    inner = a["left"] >= 0
    assert (a["right"][inner] == a["left"][inner] + 1).all() and (a["left"][inner] % 2 == 0).all()
    assert len(a["axis"]) == len(b["axis"]) + 1


def test_multi_mesh_and_obj_path(cube_obj):
    s, o = _pair(G.scene_default(cube_obj))
    assert s.mesh_count() == o.mesh_count() == 3
    for m in range(3):
        ps, ns = s.mesh_triangles(m)
        po, no = o.mesh_triangles(m)
        assert np.array_equal(ps, po) and np.array_equal(ns, no)   # readObjIntoMesh incl. y/z swap + SRT, bitwise
        wa, wb = [], []
        _walk(s.kd_export(m), 0, wa)
        _walk(o.kd_export(m), 0, wb)
        assert wa == wb
    # mesh 2 is rotated by pi about y: the cube lands on x in [-3,-1]
    p2, _ = s.mesh_triangles(2)
    assert p2[..., 0].min() == pytest.approx(-3, abs=1e-4) and p2[..., 0].max() == pytest.approx(-1, abs=1e-4)
    assert s.device_bytes() > 0 and s.build_seconds() >= 0


def test_flat_mesh_degenerate_split_becomes_leaf():
    # a planar, axis-aligned mesh has zero width on one axis: the reference divides 0/0 there (mesh.cu:76-78);
    # both builders fall back to a leaf instead (documented deviation) and still agree with each other
    n = 12
    xs = np.linspace(0, 1, n + 1, dtype=np.float32)
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (xs[i], xs[j], 0), (xs[i + 1], xs[j], 0), (xs[i + 1], xs[j + 1], 0), (xs[i], xs[j + 1], 0)
            tris += [(a, b, c), (a, c, d)]
    pos = np.array(tris, np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0], 3, 1))
    desc = G.SceneDesc(materials=[((1, 1, 1), (0, 0, 0))], meshes=[("arrays", pos, nrm, 0)])
    s, o = _pair(desc)
    wa, wb = [], []
    _walk(s.kd_export(0), 0, wa)
    _walk(o.kd_export(0), 0, wb)
    assert wa == wb


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mipt.h")).read()
    declared = set(re.findall(r"\b(mipt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.load().mipt_version()


def test_cabi_struct_sizes_match_header():
    # mipt_render_config: 12 x int32; mipt_stats: 7 x int64 + 3 x double + 6 x int64 (ctr_*) + 1 + 6 x int64 (work_*)
    assert C.sizeof(_lib.RenderConfig) == 48 and C.sizeof(_lib.Stats) == 25 * 8
    hdr = open(os.path.join(ROOT, "include", "mipt.h")).read()
    body = hdr[hdr.index("typedef struct mipt_stats"):hdr.index("} mipt_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n.strip() for decl in re.findall(r"(?:int64_t|double)\s+([^;]+);", body) for n in decl.split(",")]
    assert names == [n for n, _ in _lib.Stats._fields_]


def test_export_import_round_trip(cube_obj):
    """mipt_scene_export / _import: the packed scene of one process, byte for byte, in another scene object -- the same
    arena, the same kd-tree and triangles through the introspection calls, no rebuild; edits are refused."""
    desc = G.scene_default(cube_obj)
    pos, nrm = G.icosphere(3, 1.5, (0.0, -2.0, -1.0))
    desc.meshes.append(("arrays", pos, nrm, 0))
    desc.spheres.append(((1.0, 2.0, -1.0), 0.5, 1))
    desc.env_f32 = G.synthetic_env_faces(4)
    a = desc.apply(Scene())
    a.commit(devices=())
    blob = a.export_blob()
    b = Scene()
    b.import_blob(blob)
    b.commit(devices=())
    assert np.array_equal(b.export_blob(), blob)
    assert b.mesh_count() == a.mesh_count() == 4 and b.device_bytes() == a.device_bytes()
    for m in range(4):
        ka, kb = a.kd_export(m), b.kd_export(m)
        for k in ka:
            assert np.array_equal(ka[k], kb[k]), (m, k)
        for x, y in zip(a.mesh_triangles(m), b.mesh_triangles(m)):
            assert np.array_equal(x, y)
    with pytest.raises(_lib.MiptError, match="imported scene"):
        b.add_material((1, 1, 1), (0, 0, 0))
    with pytest.raises(_lib.MiptError, match="imported scene"):
        b.add_mesh(pos, nrm, 0)
    with pytest.raises(_lib.MiptError, match="blob"):
        Scene().import_blob(blob[:-16])
    bad = blob.copy()
    bad[0] ^= 0xFF
    with pytest.raises(_lib.MiptError, match="blob"):
        Scene().import_blob(bad)
    # a corrupt header must be refused before anything is read through it (offsets: BlobHeader / BlobMesh in scene.cpp):
    # table counts that do not fit their sections, mesh ranges that leave the arena, totals that do not add up
    def patched(offset, value, dtype):
        out = blob.copy()
        out[offset:offset + np.dtype(dtype).itemsize] = np.array([value], dtype).view(np.uint8)
        return out
    # header: magic, struct_bytes, arena_bytes (24 B); 17 section offsets (136 B); 3 totals (24 B); 9 int32; camera; time = 280 B
    for off, val, dt in ((192, 1 << 27, np.int32),      # n_planes
                         (184, 1 << 30, np.int32),      # n_materials
                         (168, 1 << 28, np.int64),      # total_nodes
                         (176, (1 << 31) - 2, np.int64),  # total_refs
                         (24 + 8 * 12, blob.size, np.uint64),  # layout.planes beyond the arena
                         (280 + 0, 1 << 40, np.int64),   # mesh 0: n_tris
                         (280 + 16, 3, np.int64),        # mesh 0: node_base
                         (280 + 80 + 32, 1 << 33, np.int64)):  # mesh 1: n_nodes
        with pytest.raises(_lib.MiptError, match="blob"):
            Scene().import_blob(patched(off, val, dt))


def test_error_behaviour_without_device():
    s = Scene()
    with pytest.raises(_lib.MiptError, match="unknown material"):
        s.add_plane((0, 0, 0), (0, 1, 0), 3)
    m = s.add_material((0.5, 0.5, 0.5), (0, 0, 0))
    with pytest.raises(_lib.MiptError, match="at least one triangle"):
        s.add_mesh(np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 3), np.float32), m)
    with pytest.raises(_lib.MiptError, match="cannot open OBJ"):
        s.add_mesh_obj("/nonexistent/cube.obj", (0, 0, 0), (0, 0, 0), (1, 1, 1), m)
    with pytest.raises(_lib.MiptError, match="cannot open HDR"):
        s.set_envmap_dir("/nonexistent/")
    with pytest.raises(_lib.MiptError, match="not committed"):
        s.trace_closest(np.zeros((1, 3)), np.ones((1, 3)))
    pos, nrm = G.cube_mesh()
    s.add_mesh(pos, nrm, m)
    s.commit(devices=())
    # the product path fails loudly when no HIP device backs the scene: there is no CPU fallback
    with pytest.raises(_lib.MiptError, match="HIP device required"):
        s.trace_closest(np.zeros((1, 3)), np.ones((1, 3)))
    with pytest.raises(_lib.MiptError, match="HIP device required"):
        s.render(make_config(8, 8, 1))


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "cuda_path_tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "oracle/" not in text and "import oracle" not in text \
                    and "from oracle" not in text, os.path.join(dirpath, f)


def test_product_holds_no_library_kernels():
    """Queue binning, the kd build's key sort, scans and compactions are the package's own wave64 kernels: nothing under
    csrc/ includes rocPRIM, hipCUB, rocThrust or Thrust."""
    csrc = os.path.join(ROOT, "cuda_path_tracer_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".cpp", ".hip")):
            text = open(os.path.join(csrc, f), errors="replace").read()
            for inc in re.findall(r"#include\s*[<\"]([^>\"]+)[>\"]", text):
                assert not any(lib in inc.lower() for lib in ("rocprim", "hipcub", "thrust", "cub/")), (f, inc)
