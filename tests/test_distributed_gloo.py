"""CPU, world_size 2 over gloo: the functions bench.py runs at N > 1 (cuda_path_tracer_amd.parallel: share_scene,
render_distributed in steps, one all-reduce) with the CPU oracle as the renderer and the product's host-side packer
as the scene builder.  The all-reduced image must equal the single-rank image up to fp32 summation order, every rank
must end up with the same framebuffer, and the scene built on rank 0 must arrive byte for byte on rank 1."""
import subprocess
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
    from cuda_path_tracer_amd.parallel import count_samples, rank_samples, render_distributed, share_scene
    from cuda_path_tracer_amd.scene import Scene, make_config
    from oracle.binding import OracleScene

    # build once on rank 0 (the product's host packer; no GPU: zero devices), broadcast, import on rank 1
    built = []

    def build():
        sc = G.scene_icosphere_room(3).apply(Scene())
        sc.commit(devices=())
        built.append(1)
        return sc

    scene = share_scene(build, Scene, rank, world, device=None)
    scene.commit(devices=())  # packed already: nothing is rebuilt
    assert len(built) == (1 if rank == 0 else 0)
    np.save(os.path.join(out_dir, "blob%d.npy" % rank), scene.export_blob())
    assert scene.kd_export(0)["n_tris"] == 20 * 4 ** 3

    o = G.scene_icosphere_room(2).apply(OracleScene(threads=2))
    o.commit()
    W, H, SPP = 40, 30, 6
    calls = []

    def render_into(b, e, s, accum):
        calls.append((b, e, s))
        img, ctr = o.render(make_config(W, H, spp=SPP, cam_depth=4, sample_begin=b, sample_end=e, sample_stride=s))
        accum += torch.from_numpy(img.reshape(-1))
        return ctr

    accum = torch.zeros(W * H * 3, dtype=torch.float32)
    # as bench.py: steps of `samples_per_step` sample indices over all ranks, ONE all-reduce at the end
    stats = render_distributed(render_into, accum, 0, SPP, rank, world, all_reduce=dist.all_reduce, samples_per_step=2 * world)
    assert calls == [(0 + rank, 4, world), (4 + rank, 6, world)] and len(stats) == 2
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
    assert np.array_equal(np.load(tmp_path / "blob0.npy"), np.load(tmp_path / "blob1.npy"))
    sys.path.insert(0, ROOT)
    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd.scene import make_config
    from oracle.binding import OracleScene
    o = G.scene_icosphere_room(2).apply(OracleScene(threads=2))
    o.commit()
    full, _ = o.render(make_config(40, 30, spp=6, cam_depth=4))
    assert np.allclose(a, full.reshape(-1), rtol=2e-6, atol=1e-7)
    assert full.mean() > 0.01


def _worker_build_fails(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuda_path_tracer_amd.parallel import share_scene
    from cuda_path_tracer_amd.scene import Scene

    def build():
        raise ValueError("no such asset")

    try:
        share_scene(build, Scene, rank, world, device=None)
        what = "returned"
    except ValueError as e:
        what = "ValueError: %s" % e
    except RuntimeError as e:
        what = "RuntimeError: %s" % e
    open(os.path.join(out_dir, "fail%d.txt" % rank), "w").write(what)
    dist.barrier()
    dist.destroy_process_group()


def test_failed_build_on_rank0_raises_on_every_rank(tmp_path):
    """share_scene: when build() raises on rank 0 the peers must not wait in the broadcast for ever -- the size travels
    as -1 and every rank raises."""
    mp.spawn(_worker_build_fails, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "fail0.txt").read() == "ValueError: no such asset"
    assert open(tmp_path / "fail1.txt").read().startswith("RuntimeError: rank 0 failed to build the scene")


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


def test_bench_refuses_to_mislabel_the_gpu_count():
    """`python bench.py --gpus N` starts N ranks itself or fails: with fewer than N visible devices (none here) it exits
    non-zero before anything runs, and under a launcher whose WORLD_SIZE disagrees with --gpus it refuses as well --
    it never prints an n_gpus it did not use."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "visible device" in r.stderr and "n_gpus" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "n_gpus" not in r.stdout
