"""Multi-GPU rendering, one process per GPU.  The reference is single-GPU (`cudaSetDevice(0)`,
kernel.cu:377); what replaces its one D2H copy (kernel.cu:416) at N > 1 is:

  * sample partition  every rank renders the FULL frame with the samples s = begin + rank, begin + rank + world, ...
                      (`rank_samples`); the RNG is keyed on (seed, pixel, sample index), so the sample set -- and the
                      image up to fp32 summation order -- does not depend on the number of ranks;
  * one collective    the fp32 accumulation framebuffers are summed by ONE all-reduce (RCCL over xGMI when the
                      process group's backend is "nccl") at the end of `render_distributed`;
  * one kd build      rank 0 builds and packs the scene, the packed arena travels by broadcast and the other ranks
                      import it (`share_scene`) -- instead of `world` contended host builds of the same trees.

`bench.py` and the gloo tests drive exactly these functions; the renderer is passed in as a callable so that the CPU
test can put the oracle behind the same partition / collective logic."""
import numpy as np


def rank_samples(sample_begin, sample_end, rank, world):
    """(begin, end, stride) of the samples rank `rank` of `world` renders out of [sample_begin, sample_end)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return sample_begin + rank, sample_end, world


def count_samples(begin, end, stride):
    return 0 if end <= begin else (end - begin + stride - 1) // stride


def render_distributed(render_into, accum, sample_begin, sample_end, rank, world, all_reduce=None, samples_per_step=0):
    """Renders this rank's share of the sample indices [sample_begin, sample_end) into `accum` and sums `accum` over ranks.

    render_into(begin, end, stride, accum) must ADD sum_s trace_s/spp for s = begin, begin + stride, ... < end into
    `accum` and return a stats dict (or None).  samples_per_step > 0 cuts the range into steps of that many sample
    INDICES (all ranks together), one render_into call each; 0 renders it in one call.  all_reduce(accum) sums the
    framebuffer over ranks in place (torch.distributed.all_reduce); None is only allowed for world == 1.
    Returns the list of the steps' stats."""
    if world > 1 and all_reduce is None:
        raise ValueError("all_reduce is required when world > 1")
    step = int(samples_per_step) if samples_per_step and samples_per_step > 0 else max(1, sample_end - sample_begin)
    stats = []
    for lo in range(sample_begin, sample_end, step):
        b, e, s = rank_samples(lo, min(sample_end, lo + step), rank, world)
        stats.append(render_into(b, e, s, accum))
    if world > 1:
        all_reduce(accum)
    return stats


def broadcast_blob(blob, rank, world, device=None):
    """Broadcasts a uint8 numpy array from rank 0 over the default torch.distributed process group (RCCL needs the
    buffer on the GPU: pass this rank's torch device; gloo takes host tensors: device=None).  Returns the array on
    every rank.  Rank 0 may pass an Exception instead of an array (its build failed): the size travels as -1 and EVERY
    rank raises, instead of the peers waiting in the second broadcast for ever."""
    if world == 1:
        if isinstance(blob, BaseException):
            raise blob
        return blob
    import torch
    import torch.distributed as dist
    failed = rank == 0 and isinstance(blob, BaseException)
    size = torch.tensor([(-1 if failed else blob.size) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=0)
    if int(size.item()) < 0:
        if failed:
            raise blob
        raise RuntimeError("rank 0 failed to build the scene; nothing to import on rank %d" % rank)
    if rank == 0:
        buf = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8))
        buf = buf.to(device) if device is not None else buf.clone()
    else:
        buf = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=0)
    return buf.cpu().numpy()


def share_scene(build, make_empty, rank, world, device=None):
    """Build once, ship everywhere.  `build()` -> a scene that has been described and packed on the host
    (`commit(devices=())`), called on rank 0 only; `make_empty()` -> an empty scene of the same class.  Returns this
    rank's scene, packed but not yet uploaded: the caller commits it on its own device (which for an imported scene
    only uploads).  If `build()` raises on rank 0, every rank raises (see broadcast_blob)."""
    if world == 1:
        return build()
    if rank == 0:
        try:
            scene = build()
            blob = scene.export_blob()
        except Exception as e:  # tell the peers before re-raising
            broadcast_blob(e, rank, world, device)
            raise
        broadcast_blob(blob, rank, world, device)
        return scene
    scene = make_empty()
    scene.import_blob(broadcast_blob(None, rank, world, device))
    return scene
