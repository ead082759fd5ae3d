"""Host-side mirror of the reference's scene-construction and render surface, on top of the C-ABI.

Reference surface mirrored (paths relative to /root/reference/cuda-path-tracer/):
  Camera::set(eye, lookat, vup, fov)                      pathTracing/camera.cuh:9
  Material(diffuseAlbedo, mirrorAlbedo)                   pathTracing/material.cuh:9
  Light(location, power)                                  pathTracing/light.cuh:15
  Plane(point, normal, Material*)                         shapes/plane.cuh:14
  Sphere(center, radius, Material*)                       shapes/sphere.cuh:15
  MeshObject(objPath, position, rotate, scale, Material*) shapes/mesh.cuh:322
  EnvMap(folder)                                          HDREnvMap/envMap.cuh:34
  Scene::build() + render<<<>>> + cudaMemcpy D2H          kernel.cu:52-142, 310-346, 397, 416
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FLAG_COUNTERS, FLAG_WORK_COUNTERS, RenderConfig, Stats, check


def _f3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_config(width, height, spp, cam_depth=6, light_depth=0, use_envmap=False, seed=0x5EED, sample_begin=0,
                sample_end=None, sample_stride=1, samples_per_pass=0, flags=0):
    return RenderConfig(int(width), int(height), int(spp), int(cam_depth), int(light_depth), int(use_envmap),
                        int(seed) & 0xFFFFFFFF, int(sample_begin), int(spp if sample_end is None else sample_end),
                        int(sample_stride), int(samples_per_pass), int(flags))


def load_hdr_rgbe(path):
    """HDRLoader::load (HDRLoader.cu:29): one Radiance file -> packed RGBE texels, uint8 array (h, w, 4), file row order."""
    lib = _lib.load()
    w, h = C.c_int32(), C.c_int32()
    check(lib.mipt_hdr_load_rgbe(str(path).encode(), C.byref(w), C.byref(h), None))
    tex = np.empty((h.value, w.value), dtype=np.uint32)
    check(lib.mipt_hdr_load_rgbe(str(path).encode(), C.byref(w), C.byref(h), _ptr(tex)))
    return tex.view(np.uint8).reshape(h.value, w.value, 4)


class Scene:
    """One `Scene scene;` of the reference (kernel.cu:383), backed by libmipt."""

    def __init__(self):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.mipt_scene_create(C.byref(h)))
        self._h = h
        self.devices = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mipt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- construction ---------------------------------------------------------------------
    def set_camera(self, eye, lookat, vup, fov_rad):
        check(self._lib.mipt_scene_set_camera(self._h, _f3(eye), _f3(lookat), _f3(vup), float(fov_rad)))

    def add_material(self, diffuse, mirror):
        return check(self._lib.mipt_scene_add_material(self._h, _f3(diffuse), _f3(mirror)))

    def add_light(self, location, power):
        return check(self._lib.mipt_scene_add_light(self._h, _f3(location), _f3(power)))

    def add_plane(self, point, normal, material):
        return check(self._lib.mipt_scene_add_plane(self._h, _f3(point), _f3(normal), int(material)))

    def add_sphere(self, center, radius, material):
        return check(self._lib.mipt_scene_add_sphere(self._h, _f3(center), float(radius), int(material)))

    def add_mesh(self, pos, nrm, material):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 9)
        if pos.shape != nrm.shape:
            raise ValueError("pos and nrm must both be (n_tris, 3, 3)")
        return check(self._lib.mipt_scene_add_mesh(self._h, _ptr(pos), _ptr(nrm), pos.shape[0], int(material)))

    def add_mesh_obj(self, path, position, rotate, scale, material):
        return check(self._lib.mipt_scene_add_mesh_obj(self._h, str(path).encode(), _f3(position), _f3(rotate),
                                                       _f3(scale), int(material)))

    def set_envmap_dir(self, folder):
        folder = str(folder)
        if not folder.endswith("/"):
            folder += "/"
        check(self._lib.mipt_scene_set_envmap_rgbe_dir(self._h, folder.encode()))

    def set_envmap_f32(self, faces6):
        faces6 = np.ascontiguousarray(faces6, dtype=np.float32)
        if faces6.ndim != 4 or faces6.shape[0] != 6 or faces6.shape[3] != 3:
            raise ValueError("faces6 must be (6, h, w, 3)")
        check(self._lib.mipt_scene_set_envmap_f32(self._h, _ptr(faces6), faces6.shape[2], faces6.shape[1]))

    def commit(self, devices=(0,)):
        """kd build + pack + upload.  devices=() packs on the host only (no GPU needed)."""
        devices = list(devices)
        arr = (C.c_int * max(1, len(devices)))(*devices) if devices else None
        check(self._lib.mipt_scene_commit(self._h, arr, len(devices)))
        self.devices = devices

    def export_blob(self):
        """The packed scene (kd-trees, leaf records, tables, cube map) as one uint8 array: build on one rank, broadcast,
        `import_blob` + `commit` on the others (no second kd build)."""
        n = check(self._lib.mipt_scene_export_size(self._h))
        blob = np.empty(n, dtype=np.uint8)
        check(self._lib.mipt_scene_export(self._h, _ptr(blob), n))
        return blob

    def import_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.mipt_scene_import(self._h, _ptr(blob), blob.size))
        self.devices = []

    def set_caller_stream(self, stream_handle, dev_slot=0):
        """hipStream_t (as an integer; 0 = the default stream) whose queued work the *_device calls wait for on entry."""
        check(self._lib.mipt_scene_set_caller_stream(self._h, int(dev_slot), C.c_void_p(int(stream_handle) or None)))

    # ---- rendering --------------------------------------------------------------------------
    def render(self, cfg):
        """Returns (image float32 (H, W, 3), stats dict)."""
        img = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
        st = Stats()
        check(self._lib.mipt_render(self._h, C.byref(cfg), _ptr(img), C.byref(st)))
        return img, st.as_dict()

    def render_device(self, cfg, accum_ptr, dev_slot=0):
        """Adds this call's samples into a device buffer (W*H*3 fp32) given by its address."""
        st = Stats()
        check(self._lib.mipt_render_device(self._h, int(dev_slot), C.byref(cfg), C.c_void_p(int(accum_ptr)), C.byref(st)))
        return st.as_dict()

    def trace_closest(self, ray_o, ray_d, counters=False, work_counters=False):
        """Closest hit of Ray(o, d) for host arrays (n, 3).  Returns (prim int32 (n,), t float32 (n,), stats)."""
        ray_o = np.ascontiguousarray(ray_o, dtype=np.float32).reshape(-1, 3)
        ray_d = np.ascontiguousarray(ray_d, dtype=np.float32).reshape(-1, 3)
        if ray_o.shape != ray_d.shape:
            raise ValueError("ray_o and ray_d must both be (n, 3)")
        n = ray_o.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        st = Stats()
        check(self._lib.mipt_trace_closest(self._h, _ptr(ray_o), _ptr(ray_d), n, _ptr(prim), _ptr(t), C.byref(st),
                                           FLAG_COUNTERS if counters else (FLAG_WORK_COUNTERS if work_counters else 0)))
        return prim, t, st.as_dict()

    def trace_closest_device(self, o_ptr, d_ptr, n, prim_ptr, t_ptr, counters=False, dev_slot=0, work_counters=False):
        st = Stats()
        check(self._lib.mipt_trace_closest_device(self._h, int(dev_slot), C.c_void_p(int(o_ptr)), C.c_void_p(int(d_ptr)),
                                                  int(n), C.c_void_p(int(prim_ptr)), C.c_void_p(int(t_ptr)),
                                                  C.byref(st),
                                                  FLAG_COUNTERS if counters else (FLAG_WORK_COUNTERS if work_counters else 0)))
        return st.as_dict()

    def trace_closest_device_noalloc(self, o_ptr, d_ptr, n, prim_ptr, t_ptr, dev_slot=0):
        """Same results from the library's fixed scratch array: no allocation, at most 2**20 rays per call."""
        check(self._lib.mipt_trace_closest_device_noalloc(self._h, int(dev_slot), C.c_void_p(int(o_ptr)), C.c_void_p(int(d_ptr)),
                                                          int(n), C.c_void_p(int(prim_ptr)), C.c_void_p(int(t_ptr))))

    def queue_bin_device(self, keys_ptr, vals_ptr, count_ptr, capacity, key_bits, out_ptr, dev_slot=0):
        """The renderer's queue-binning stage (csrc/queue_sort.h) on caller-provided device buffers; the length is the
        int32 at count_ptr (device memory)."""
        check(self._lib.mipt_queue_bin_device(self._h, int(dev_slot), C.c_void_p(int(keys_ptr)), C.c_void_p(int(vals_ptr)),
                                              C.c_void_p(int(count_ptr)), int(capacity), int(key_bits), C.c_void_p(int(out_ptr))))

    def env_lookup(self, dirs):
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        rgb = np.empty_like(dirs)
        check(self._lib.mipt_env_lookup(self._h, _ptr(dirs), dirs.shape[0], _ptr(rgb)))
        return rgb

    # ---- introspection ----------------------------------------------------------------------
    def mesh_count(self):
        return check(self._lib.mipt_mesh_count(self._h))

    def kd_export(self, mesh):
        nn, nr, nt = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.mipt_kd_sizes(self._h, int(mesh), C.byref(nn), C.byref(nr), C.byref(nt)))
        out = {
            "boxes": np.empty((nn.value, 6), np.float32), "axis": np.empty(nn.value, np.int32),
            "left": np.empty(nn.value, np.int32), "right": np.empty(nn.value, np.int32),
            "first": np.empty(nn.value, np.int32), "count": np.empty(nn.value, np.int32),
            "refs": np.empty(nr.value, np.int32),
        }
        check(self._lib.mipt_kd_export(self._h, int(mesh), *[_ptr(out[k]) for k in
                                                             ("boxes", "axis", "left", "right", "first", "count", "refs")]))
        out["n_tris"] = nt.value
        return out

    def mesh_triangles(self, mesh):
        nt = C.c_int64()
        check(self._lib.mipt_kd_sizes(self._h, int(mesh), None, None, C.byref(nt)))
        pos = np.empty((nt.value, 3, 3), np.float32)
        nrm = np.empty((nt.value, 3, 3), np.float32)
        check(self._lib.mipt_mesh_triangles(self._h, int(mesh), _ptr(pos), _ptr(nrm)))
        return pos, nrm

    def gpu_built_meshes(self):
        """Meshes whose kd-tree the last commit built on the GPU (the others: host threads; same tree either way)."""
        return check(self._lib.mipt_scene_gpu_built_meshes(self._h))

    def packed_on_gpu(self):
        """True when the last commit made the arena's big sections on the GPU (csrc/pack_gpu.hip)."""
        return bool(check(self._lib.mipt_scene_packed_on_gpu(self._h)))

    def device_bytes(self):
        return int(self._lib.mipt_scene_device_bytes(self._h))

    def build_seconds(self):
        return float(self._lib.mipt_scene_build_seconds(self._h))

    def kd_seconds(self):
        return float(self._lib.mipt_scene_kd_seconds(self._h))
