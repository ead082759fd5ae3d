"""GPU: mipt_trace_closest_device_noalloc (the entry point that allocates nothing: fixed device scratch array) against
mipt_trace_closest_device on the same device buffers, bit for bit, on two calls in a row -- the second with a smaller
input, so stale contents of the shared scratch (cursors, hit records of the first call) would show."""
import ctypes as C

import numpy as np
import pytest

from cuda_path_tracer_amd import _lib, generators as G
from cuda_path_tracer_amd.scene import Scene

pytestmark = pytest.mark.gpu
MAX_RAYS = 1 << 20


def _noalloc(lib, scene, d_o, d_d, n, d_p, d_t, slot=0):
    fn = lib.mipt_trace_closest_device_noalloc
    return fn(scene._h, slot, C.c_void_p(d_o.data_ptr()), C.c_void_p(d_d.data_ptr()), n, C.c_void_p(d_p.data_ptr()),
              C.c_void_p(d_t.data_ptr()))


def test_noalloc_entry_point_matches_the_allocating_one():
    import torch
    lib = _lib.load()
    desc = G.scene_icosphere_room(5)
    s = desc.apply(Scene())
    s.commit(devices=(0,))
    for n in (200_000, 70_001):  # second call smaller: same scratch, fewer rays
        ro1, rd1 = G.camera_rays(desc.camera, 512, 512, n // 2, seed=n)
        ro2, rd2 = G.incoherent_rays(n - n // 2, (-3, -3.5, -4.5), (3, 1.5, 0.5), seed=n + 1)
        d_o = torch.from_numpy(np.concatenate([ro1, ro2])).cuda()
        d_d = torch.from_numpy(np.concatenate([rd1, rd2])).cuda()
        ref_p = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        ref_t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
        got_p, got_t = ref_p.clone(), ref_t.clone()
        torch.cuda.synchronize()
        s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, ref_p.data_ptr(), ref_t.data_ptr())
        # "allocates nothing": free device memory as the DRIVER sees it (hipMemGetInfo: a hipMalloc / hipFree inside libmipt
        # would move it; torch's own counter only sees torch's caching allocator) is the same before and after the call
        before = torch.cuda.memory_allocated()
        free_before, _ = torch.cuda.mem_get_info()
        _lib.check(_noalloc(lib, s, d_o, d_d, n, got_p, got_t))
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, (free_before, free_after)
        assert torch.cuda.memory_allocated() == before
        assert torch.equal(got_p, ref_p)
        assert torch.equal(got_t.view(torch.int32), ref_t.view(torch.int32))
        assert (ref_p >= 0).float().mean().item() > 0.3
    # sizes beyond the fixed scratch are refused with a clear error, n == 0 is a no-op
    big = torch.zeros(3, dtype=torch.float32, device="cuda")
    rc = _noalloc(lib, s, big, big, MAX_RAYS + 1, got_p, got_t)
    assert rc < 0 and b"at most 1048576" in lib.mipt_last_error()
    assert _noalloc(lib, s, big, big, 0, got_p, got_t) == 0
