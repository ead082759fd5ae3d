"""ctypes binding of libmipt.so (include/mipt.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPT_LIB") or os.path.join(_HERE, "libmipt.so")  # MIPT_LIB: a developer build (scripts/build_phase_stats.sh)


class RenderConfig(C.Structure):
    """mipt_render_config (replaces constants/constants.cuh:6-9)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("cam_depth", C.c_int32),
        ("light_depth", C.c_int32), ("use_envmap", C.c_int32), ("seed", C.c_uint32),
        ("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("sample_stride", C.c_int32),
        ("samples_per_pass", C.c_int32), ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    """mipt_stats."""
    _fields_ = [
        ("rays_total", C.c_int64), ("rays_camera", C.c_int64), ("rays_shadow", C.c_int64), ("rays_light", C.c_int64),
        ("samples", C.c_int64), ("extend_launches", C.c_int64), ("rays_extended", C.c_int64),
        ("ms_total", C.c_double), ("ms_extend", C.c_double), ("ms_other", C.c_double),
        ("ctr_nodes", C.c_int64), ("ctr_pops", C.c_int64), ("ctr_refs", C.c_int64), ("ctr_tris", C.c_int64),
        ("ctr_hits", C.c_int64), ("ctr_culled", C.c_int64), ("nonfinite_samples", C.c_int64),
        ("work_steps", C.c_int64), ("work_pops", C.c_int64), ("work_sphere_pairs", C.c_int64), ("work_planes", C.c_int64),
        ("work_verts", C.c_int64), ("work_leaves", C.c_int64), ("shade_launches", C.c_int64), ("ms_shade", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


FLAG_COUNTERS = 1
FLAG_DROP_NONFINITE = 2
FLAG_WORK_COUNTERS = 4

_F3 = C.POINTER(C.c_float)
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/mipt.h declares
SIGNATURES = {
    "mipt_scene_create": (C.c_int, [C.POINTER(_VP)]),
    "mipt_scene_destroy": (None, [_VP]),
    "mipt_scene_set_camera": (C.c_int, [_VP, _F3, _F3, _F3, C.c_float]),
    "mipt_scene_add_material": (C.c_int, [_VP, _F3, _F3]),
    "mipt_scene_add_light": (C.c_int, [_VP, _F3, _F3]),
    "mipt_scene_add_plane": (C.c_int, [_VP, _F3, _F3, C.c_int]),
    "mipt_scene_add_sphere": (C.c_int, [_VP, _F3, C.c_float, C.c_int]),
    "mipt_scene_add_mesh": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int]),
    "mipt_scene_add_mesh_obj": (C.c_int, [_VP, C.c_char_p, _F3, _F3, _F3, C.c_int]),
    "mipt_scene_set_envmap_rgbe_dir": (C.c_int, [_VP, C.c_char_p]),
    "mipt_scene_set_envmap_f32": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "mipt_scene_commit": (C.c_int, [_VP, C.POINTER(C.c_int), C.c_int]),
    "mipt_scene_export_size": (C.c_int64, [_VP]),
    "mipt_scene_export": (C.c_int, [_VP, _VP, C.c_int64]),
    "mipt_scene_import": (C.c_int, [_VP, _VP, C.c_int64]),
    "mipt_scene_set_caller_stream": (C.c_int, [_VP, C.c_int, _VP]),
    "mipt_render": (C.c_int, [_VP, C.POINTER(RenderConfig), _VP, C.POINTER(Stats)]),
    "mipt_render_device": (C.c_int, [_VP, C.c_int, C.POINTER(RenderConfig), _VP, C.POINTER(Stats)]),
    "mipt_trace_closest": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, C.POINTER(Stats), C.c_int]),
    "mipt_trace_closest_device": (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int64, _VP, _VP, C.POINTER(Stats), C.c_int]),
    "mipt_trace_closest_device_noalloc": (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int64, _VP, _VP]),
    "mipt_queue_bin_device": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, C.c_int64, C.c_int, _VP]),
    "mipt_env_lookup": (C.c_int, [_VP, _VP, C.c_int64, _VP]),
    "mipt_scene_gpu_built_meshes": (C.c_int, [_VP]),
    "mipt_scene_packed_on_gpu": (C.c_int, [_VP]),
    "mipt_hdr_load_rgbe": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _VP]),
    "mipt_mesh_count": (C.c_int, [_VP]),
    "mipt_kd_sizes": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mipt_kd_export": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mipt_mesh_triangles": (C.c_int, [_VP, C.c_int, _VP, _VP]),
    "mipt_scene_device_bytes": (C.c_int64, [_VP]),
    "mipt_scene_build_seconds": (C.c_double, [_VP]),
    "mipt_scene_kd_seconds": (C.c_double, [_VP]),
    "mipt_last_error": (C.c_char_p, []),
    "mipt_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libmipt.so; raises if it has not been built (`__graft_entry__.build()` / `make -C csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MiptError(RuntimeError):
    pass


def check(rc):
    if rc < 0:
        raise MiptError(f"libmipt error {rc}: {load().mipt_last_error().decode(errors='replace')}")
    return rc
