"""GPU: the hand-written queue-binning stage (csrc/queue_sort.h) that orders shadow / light ray queues by direction cell in
front of the closest-hit kernel, driven through the C-ABI (mipt_queue_bin_device) on explicit keys.

Contract checked here (the order of a queue changes no result, so the stage promises coherence, not a total order): the
output is a permutation of the input values; the top 14 key bits decrease at most n / 262144 times (slices of a bucket that
one crowded key range made huge); on uniform keys neighbours are in order up to a few low bits.  Lengths are read from
device memory; edge cases: empty queue, one entry, all keys equal (the wave-uniform path and the oversized-bin path), keys in
one narrow range, runs of the largest key, a length that is not the capacity."""
import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene

pytestmark = pytest.mark.gpu
CHUNK = 1 << 18  # kSortChunk


@pytest.fixture(scope="module")
def scene():
    import torch  # before libmipt initialises HIP: torch brings its own copy of the runtime and wants to be first
    torch.cuda.init()
    s = G.scene_icosphere_room(2).apply(Scene())
    s.commit(devices=(0,))
    yield s
    s.close()


def _bin(scene, keys, n, key_bits, capacity=None):
    import torch
    capacity = capacity or max(1, len(keys))
    d_keys = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint32).view(np.int32)).cuda()
    vals = np.arange(len(keys), dtype=np.int32) * 3 + 7
    d_vals = torch.from_numpy(vals).cuda()
    d_count = torch.tensor([n], dtype=torch.int32, device="cuda")
    d_out = torch.full((max(1, len(keys)),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    scene.queue_bin_device(d_keys.data_ptr(), d_vals.data_ptr(), d_count.data_ptr(), capacity, key_bits, d_out.data_ptr())
    return vals, d_out.cpu().numpy()


def _check(keys, vals, out, n, key_bits, uniform=False):
    assert np.array_equal(out[n:], np.full(len(out) - n, -1, np.int32)), "wrote past the queue's length"
    got = out[:n]
    assert np.array_equal(np.sort(got), np.sort(vals[:n])), "not a permutation of the input"
    if n == 0:
        return
    k = keys[(got - 7) // 3].astype(np.int64)
    fine = k >> (key_bits - 14)
    drops = int(np.count_nonzero(np.diff(fine) < 0))
    assert drops <= n // CHUNK, (drops, n // CHUNK)
    if uniform and n >= 8000:  # buckets span ~16-40 fine bins: ordered down to a few bits above the last one
        assert np.diff(k).min() > -(1 << (key_bits - 19)), int(np.diff(k).min())
    return k


@pytest.mark.parametrize("n", [0, 1, 63, 64, 8191, 8192, 8193, 100_003, 3_000_001])
def test_random_keys(scene, n):
    rng = np.random.default_rng(n + 1)
    keys = rng.integers(0, 1 << 24, size=max(n, 1), dtype=np.uint32)
    vals, out = _bin(scene, keys, n, 24)
    _check(keys, vals, out, n, 24, uniform=True)


def test_length_comes_from_the_device_not_the_capacity(scene):
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 24, size=500_000, dtype=np.uint32)
    for n in (500_000, 123_457, 2):  # shrinking lengths over the same buffers: nothing of an earlier, longer call may show
        vals, out = _bin(scene, keys, n, 24, capacity=500_000)
        _check(keys, vals, out, n, 24)


def test_skewed_and_degenerate_keys(scene):
    rng = np.random.default_rng(9)
    n = 1_200_000
    cases = {
        "all equal": np.full(n, 0x00abcdef, np.uint32),
        "all largest (unbinned entries)": np.full(n, (1 << 24) - 1, np.uint32),
        "one huge bucket": (np.uint32(0x2a) << np.uint32(14)) | rng.integers(0, 1 << 14, size=n, dtype=np.uint32),
        "runs of the largest key between binned entries": np.where((np.arange(n) // 1536) % 2 == 0, rng.integers(0, 1 << 24, size=n, dtype=np.uint32),
                                                                   np.uint32((1 << 24) - 1)).astype(np.uint32),
        "two cells": np.where(rng.random(n) < 0.5, np.uint32(5), np.uint32(0xfffff0)).astype(np.uint32),
        "sorted already": np.sort(rng.integers(0, 1 << 24, size=n, dtype=np.uint32)),
        "reversed": np.sort(rng.integers(0, 1 << 24, size=n, dtype=np.uint32))[::-1].copy(),
    }
    for name, keys in cases.items():
        vals, out = _bin(scene, keys, n, 24)
        k = _check(keys, vals, out, n, 24)
        if name == "one huge bucket":  # 16 crowded fine bins: a bucket each, ordered down to the last bit
            assert np.all(np.diff(k) >= 0), name


def test_wider_keys_light_index_above_the_cell(scene):
    """First light rays are keyed (light << 24 | cell): key_bits = 24 + bits of the light index."""
    rng = np.random.default_rng(11)
    n = 700_000
    for light_bits in (1, 3, 8):
        keys = (rng.integers(0, 1 << light_bits, size=n, dtype=np.uint32) << np.uint32(24)) | rng.integers(0, 1 << 24, size=n, dtype=np.uint32)
        vals, out = _bin(scene, keys, n, 24 + light_bits)
        _check(keys, vals, out, n, 24 + light_bits, uniform=True)
