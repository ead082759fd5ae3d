"""GPU: the N > 1 path of bench.py end to end on the hardware a test box has -- two rank processes sharing ONE GPU over
gloo (`--share-gpu --backend gloo`; RCCL refuses two ranks on one device).  What runs is everything the 8-GPU scaling run
runs except the RCCL transport: the rank launcher, `share_scene` (rank 0 builds, the arena is broadcast and imported),
`render_distributed` (interleaved samples, one all-reduce), the strong-scaling leg, the JSON line.  The frame of the two
ranks together must be the frame one rank renders from the same sample indices, up to fp32 summation order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp-per-step", "4", "--width", "640",
                        "--height", "360", "--level", "4", "--no-pmc", "--no-closest", "--no-cpu-baseline", *extra],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints, once
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_render_the_single_rank_frame():
    two = _bench("--gpus", "2", "--backend", "gloo", "--share-gpu")
    assert two["n_gpus"] == 2 and "rehearsal" in two and two["scaling"] == "weak" and two["strong"]["scaling"] == "strong"
    assert two["config"]["scene_build"].startswith("rank 0 builds")
    assert two["nonfinite_samples"] == 0 and two["value"] > 0 and two["strong"]["value"] > 0
    # weak scaling: 2 ranks x 2 steps x 4 spp = the 16 sample indices one rank renders in 4 steps of 4
    one = _bench("--gpus", "1", "--steps", "4")
    assert one["n_gpus"] == 1 and "rehearsal" not in one and "strong" not in one
    assert np.isclose(two["image_mean"], one["image_mean"], rtol=2e-6), (two["image_mean"], one["image_mean"])
    assert two["rays_per_sample"] > 0


def test_in_process_rccl_reduce_on_a_one_device_communicator(monkeypatch):
    """mipt_render's device-side combine (api.hip: dlopen(librccl) -> ncclCommInitAll -> ncclReduce to slot 0, the replacement
    of the D2H copy of kernel.cu:416 at G > 1) forced on for the single device of the test box (MIPT_FORCE_RCCL=1): RCCL is
    opened, a communicator is made and the reduce runs on the library's stream; the image is the one the plain path gives."""
    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd.scene import Scene, make_config
    s = G.scene_icosphere_room(4).apply(Scene())
    s.commit(devices=(0,))
    cfg = make_config(160, 96, spp=3, cam_depth=4, light_depth=2)
    plain, _ = s.render(cfg)
    monkeypatch.setenv("MIPT_FORCE_RCCL", "1")
    via_rccl, st = s.render(cfg)
    again, _ = s.render(cfg)  # the communicator is kept between calls
    assert np.array_equal(plain.view(np.uint32), via_rccl.view(np.uint32)) and np.array_equal(plain.view(np.uint32), again.view(np.uint32))
    assert st["samples"] == 160 * 96 * 3
