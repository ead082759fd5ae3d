"""CPU, world_size 2 over gloo: the multi-GPU sample partition + all-reduce path (cuda_path_tracer_amd.parallel)
driven with the CPU oracle as the renderer.  The all-reduced image must equal the single-rank image up to fp32
summation order, and every rank must end up with the same framebuffer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd.parallel import count_samples, rank_samples, render_distributed
    from cuda_path_tracer_amd.scene import make_config
    from oracle.binding import OracleScene

    o = G.scene_icosphere_room(2).apply(OracleScene(threads=2))
    o.commit()
    W, H, SPP = 40, 30, 6

    def render_into(b, e, s, accum):
        img, _ = o.render(make_config(W, H, spp=SPP, cam_depth=4, sample_begin=b, sample_end=e, sample_stride=s))
        accum += torch.from_numpy(img.reshape(-1))

    accum = torch.zeros(W * H * 3, dtype=torch.float32)
    render_distributed(render_into, accum, 0, SPP, rank, world, all_reduce=dist.all_reduce)
    b, e, s = rank_samples(0, SPP, rank, world)
    assert count_samples(b, e, s) == SPP // world
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_partition_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)
    sys.path.insert(0, ROOT)
    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd.scene import make_config
    from oracle.binding import OracleScene
    o = G.scene_icosphere_room(2).apply(OracleScene(threads=2))
    o.commit()
    full, _ = o.render(make_config(40, 30, spp=6, cam_depth=4))
    assert np.allclose(a, full.reshape(-1), rtol=2e-6, atol=1e-7)
    assert full.mean() > 0.01


def test_rank_samples_cover_range_exactly_once():
    from cuda_path_tracer_amd.parallel import count_samples, rank_samples
    for world in (1, 2, 3, 4, 8):
        for begin, end in ((0, 0), (0, 1), (0, 7), (5, 64), (0, 256)):
            seen = []
            for r in range(world):
                b, e, s = rank_samples(begin, end, r, world)
                seen += list(range(b, e, s))
                assert count_samples(b, e, s) == len(range(b, e, s))
            assert sorted(seen) == list(range(begin, end))
    with pytest.raises(ValueError):
        rank_samples(0, 4, 2, 2)
