"""Deterministic procedural inputs: meshes, the five BASELINE scenes and ray batches.

The reference's `resources/objects/*.obj` are absent from its repository (swallowed by its own
.gitignore), so the default scene's cube is synthesised: vertices (+-1,+-1,+-1), 12 triangles,
per-face normals.  Scene descriptions are backend-neutral: `SceneDesc.apply(scene)` drives any object
with the Scene method names of cuda_path_tracer_amd.scene (the product) or oracle.binding (the checker).
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

SEED = 0x5EED


# --------------------------------------------------------------------------------------- meshes
def cube_indexed():
    """Unit cube (+-1): 8 positions, 6 face normals, 12 triangles as (pos idx, normal idx) triples."""
    P = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32)
    faces = [  # (normal, quad corners as indices into P, counter-clockwise seen from outside)
        ((1, 0, 0), (4, 6, 7, 5)), ((-1, 0, 0), (0, 1, 3, 2)), ((0, 1, 0), (2, 3, 7, 6)),
        ((0, -1, 0), (0, 4, 5, 1)), ((0, 0, 1), (1, 5, 7, 3)), ((0, 0, -1), (0, 2, 6, 4)),
    ]
    N = np.array([f[0] for f in faces], dtype=np.float32)
    tris = []
    for ni, (_, q) in enumerate(faces):
        tris.append(((q[0], ni), (q[1], ni), (q[2], ni)))
        tris.append(((q[0], ni), (q[2], ni), (q[3], ni)))
    return P, N, tris


def write_cube_obj(path):
    """cube.obj stand-in in the `f p//n` form read by readObjIntoMesh (mesh.cu:269-291)."""
    P, N, tris = cube_indexed()
    with open(path, "w") as f:
        for p in P:
            f.write("v %g %g %g\n" % tuple(p))
        for n in N:
            f.write("vn %g %g %g\n" % tuple(n))
        for t in tris:
            f.write("f " + " ".join("%d//%d" % (pi + 1, ni + 1) for pi, ni in t) + "\n")
    return path


def write_icosphere_obj(path, level, radius=1.0, center=(0.0, 0.0, 0.0), form="p//n", eol="\n"):
    """Indexed icosphere as an OBJ file (`v`, `vn`, `f` in the `p//n` or `p/t/n` form).  Coordinates are written with 9
    significant digits, so the fp32 values of `icosphere()` come back bit for bit.  readObjIntoMesh swaps y and z
    (math.cuh:55): the mesh a scene gets from this file is `icosphere(...)` with the y and z columns exchanged."""
    V, F = icosphere_indexed(level)
    P = (V * radius + np.asarray(center, np.float64)).astype(np.float32)
    N = V.astype(np.float32)
    idx = F + 1
    with open(path, "w", newline="") as f:
        f.write("# icosphere level %d: %d vertices, %d faces%s" % (level, len(V), len(F), eol))
        np.savetxt(f, P, fmt="v %.9g %.9g %.9g", newline=eol)
        np.savetxt(f, N, fmt="vn %.9g %.9g %.9g", newline=eol)
        if form == "p/t/n":
            f.write("vt 0.5 0.5" + eol)
            np.savetxt(f, np.stack([idx[:, 0], idx[:, 0], idx[:, 1], idx[:, 1], idx[:, 2], idx[:, 2]], 1), fmt="f %d/1/%d %d/1/%d %d/1/%d", newline=eol)
        else:
            np.savetxt(f, np.stack([idx[:, 0], idx[:, 0], idx[:, 1], idx[:, 1], idx[:, 2], idx[:, 2]], 1), fmt="f %d//%d %d//%d %d//%d", newline=eol)
    return path


def cube_mesh(center=(0, 0, 0), half=1.0):
    P, N, tris = cube_indexed()
    pos = np.array([[P[pi] for pi, _ in t] for t in tris], dtype=np.float32) * np.float32(half) + np.asarray(center, np.float32)
    nrm = np.array([[N[ni] for _, ni in t] for t in tris], dtype=np.float32)
    return pos, nrm


def icosphere_indexed(level):
    """Unit icosphere as (V float64 (n_verts, 3), F int64 (20*4**level, 3)); 10*4**level + 2 vertices."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    V = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
                  [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        nv = V.shape[0]
        e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * nv + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // nv, uniq % nv
        mid = V[a] + V[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        V = np.concatenate([V, mid], axis=0)
        nf = F.shape[0]
        m01, m12, m20 = nv + inv[:nf], nv + inv[nf:2 * nf], nv + inv[2 * nf:]
        F = np.concatenate([
            np.stack([F[:, 0], m01, m20], 1), np.stack([F[:, 1], m12, m01], 1),
            np.stack([F[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    return V, F


def icosphere(level, radius=1.0, center=(0.0, 0.0, 0.0)):
    """Icosphere with 20*4**level triangles; smooth normals = normalised (position - center).
    Returns de-indexed (pos, nrm) float32 arrays of shape (n_tris, 3, 3)."""
    V, F = icosphere_indexed(level)
    nrm = V[F].astype(np.float32)
    pos = (V[F] * radius + np.asarray(center, np.float64)).astype(np.float32)
    return np.ascontiguousarray(pos), np.ascontiguousarray(nrm)


# --------------------------------------------------------------------------------------- scenes
@dataclass
class SceneDesc:
    camera: tuple = None                       # (eye, lookat, vup, fov_rad)
    materials: list = field(default_factory=list)   # (diffuse, mirror)
    lights: list = field(default_factory=list)      # (location, power)
    planes: list = field(default_factory=list)      # (point, normal, material)
    spheres: list = field(default_factory=list)     # (center, radius, material)
    meshes: list = field(default_factory=list)      # ("arrays", pos, nrm, material) | ("obj", path, position, rotate, scale, material)
    env_dir: str = None
    env_f32: object = None

    def apply(self, scene):
        if self.camera is not None:
            scene.set_camera(*self.camera)
        for m in self.materials:
            scene.add_material(*m)
        for l in self.lights:
            scene.add_light(*l)
        for p in self.planes:
            scene.add_plane(*p)
        for s in self.spheres:
            scene.add_sphere(*s)
        for m in self.meshes:
            if m[0] == "arrays":
                scene.add_mesh(m[1], m[2], m[3])
            else:
                scene.add_mesh_obj(*m[1:])
        if self.env_dir is not None:
            scene.set_envmap_dir(self.env_dir)
        if self.env_f32 is not None:
            scene.set_envmap_f32(self.env_f32)
        return scene

    def n_triangles(self):
        n = 0
        for m in self.meshes:
            n += m[1].shape[0] if m[0] == "arrays" else 12
        return n


def _box_room(desc):
    """Camera, light and the six grey planes of Scene::build (kernel.cu:53-96)."""
    desc.camera = ((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), np.float32(100.0) * np.float32(math.pi) / np.float32(180.0))
    desc.lights.append(((0.0, -4.0, -4.5), (1000.0, 1000.0, 1000.0)))
    grey = len(desc.materials)
    desc.materials.append(((0.9, 0.9, 0.9), (0.0, 0.0, 0.0)))
    for point, normal in (((0, -5, 0), (0, 1, 0)), ((0, 0, 5), (0, 0, 1)), ((0, 0, -5), (0, 0, 1)),
                          ((5, 0, 0), (1, 0, 0)), ((-5, 0, 0), (1, 0, 0)), ((0, 5, 0), (0, 1, 0))):
        desc.planes.append((point, normal, grey))
    return desc


def scene_default(cube_obj_path):
    """C1/C3: the reference's default scene (kernel.cu:53-137) with the synthesised cube.obj."""
    d = _box_room(SceneDesc())
    specs = [((2.0, -4.0, -4.0), (0.0, 0.0, 0.0), (0.8392, 0.0, 0.4392)),
             ((0.0, -4.0, -2.5), (0.0, 0.0, 0.0), (0.6078, 0.3098, 0.5882)),
             ((-2.0, -4.0, -4.0), (0.0, float(np.float32(math.pi)), 0.0), (0.0, 0.2196, 0.6588))]
    for pos, rot, col in specs:
        mat = len(d.materials)
        d.materials.append((col, (0.0, 0.0, 0.0)))
        d.meshes.append(("obj", cube_obj_path, pos, rot, (1.0, 1.0, 1.0), mat))
    return d


def scene_icosphere_room(level=6):
    """C2: the box room + one icosphere (level 6 = 81,920 triangles), radius 2.5 at (0,-1,-2), diffuse 0.8."""
    d = _box_room(SceneDesc())
    mat = len(d.materials)
    d.materials.append(((0.8, 0.8, 0.8), (0.0, 0.0, 0.0)))
    pos, nrm = icosphere(level, 2.5, (0.0, -1.0, -2.0))
    d.meshes.append(("arrays", pos, nrm, mat))
    return d


def scene_env_sphere(level=8, env_dir=None, env_f32=None):
    """C4: open scene -- floor plane y=-5, mirror-ish icosphere (level 8 = 1,310,720 triangles) radius 2.5 at the
    origin, cube-map environment."""
    d = SceneDesc()
    d.camera = ((0.0, 0.0, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), np.float32(60.0) * np.float32(math.pi) / np.float32(180.0))
    d.lights.append(((0.0, 8.0, 4.0), (2000.0, 2000.0, 2000.0)))
    d.materials.append(((0.9, 0.9, 0.9), (0.0, 0.0, 0.0)))
    d.planes.append(((0, -5, 0), (0, 1, 0), 0))
    d.materials.append(((0.2, 0.2, 0.2), (0.7, 0.7, 0.7)))
    pos, nrm = icosphere(level, 2.5, (0.0, 0.0, 0.0))
    d.meshes.append(("arrays", pos, nrm, 1))
    d.env_dir = env_dir
    d.env_f32 = env_f32
    return d


def scene_multi_sphere(level=8):
    """C5: eight overlapping icospheres (8 x level 8 = 10,485,760 triangles), radius 2.4 at (+-2.2)^3, box room."""
    d = _box_room(SceneDesc())
    base, nrm = icosphere(level, 2.4, (0.0, 0.0, 0.0))
    k = 0
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                mat = len(d.materials)
                d.materials.append(((0.3 + 0.08 * k, 0.7 - 0.05 * k, 0.5), (0.0, 0.0, 0.0)))
                off = np.array([2.2 * sx, 2.2 * sy, 2.2 * sz], np.float32)
                d.meshes.append(("arrays", base + off, nrm, mat))
                k += 1
    return d


def synthetic_env_faces(size=8, seed=SEED):
    """Six small fp32 cube faces with distinct, position-dependent values (for lookup goldens)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    return rng.random((6, size, size, 3), dtype=np.float32) * np.float32(4.0)


# --------------------------------------------------------------------------------------- rays
def camera_rays(camera, width, height, n=None, seed=SEED):
    """Coherent batch: un-normalised primary-ray directions of Camera::getRay (camera.cuh:17-20) through
    jittered pixel positions, in pixel order (the library normalises like the Ray constructor)."""
    eye, lookat, vup, fov = [np.asarray(c, np.float64) for c in camera[:3]] + [float(camera[3])]
    w = eye - lookat
    f = np.linalg.norm(w)
    right = np.cross(vup, w)
    right = right / np.linalg.norm(right) * f * math.tan(fov / 2)
    up = np.cross(w, right)
    up = up / np.linalg.norm(up) * f * math.tan(fov / 2)
    npix = width * height
    n = npix if n is None else n
    idx = np.arange(n, dtype=np.int64) % npix
    rng = np.random.Generator(np.random.Philox(key=seed))
    X = (idx % width) + rng.random(n)
    Y = (idx // width) + rng.random(n)
    d = lookat + right * (2 * X / width - 1)[:, None] + up * (2 * Y / height - 1)[:, None] - eye
    o = np.broadcast_to(eye, d.shape)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def incoherent_rays(n, aabb_lo, aabb_hi, radius=6.0, seed=SEED + 1):
    """Incoherent batch: origins uniform on a sphere of `radius`, aimed at uniform points of an AABB."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = v * radius
    lo, hi = np.asarray(aabb_lo, np.float64), np.asarray(aabb_hi, np.float64)
    tgt = lo + rng.random((n, 3)) * (hi - lo)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(tgt - o, np.float32)


def reference_env_dir():
    """The six shipped 1024^2 Radiance faces, if the reference checkout is present (never on the GPU box)."""
    p = "/root/reference/cuda-path-tracer/resources/hdr-env-cube/"
    return p if os.path.isdir(p) else None
