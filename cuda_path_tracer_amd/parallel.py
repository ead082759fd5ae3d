"""Multi-GPU sample partitioning: one process per GPU, every rank renders the FULL frame with
the samples s = begin + rank, begin + rank + world, ... (interleaved), the scene is replicated, and the fp32
accumulation framebuffer is combined with ONE all-reduce (RCCL over xGMI when the backend is "nccl").

The RNG is keyed on (seed, pixel, sample index), so the sample set -- and therefore the image up to fp32
summation order -- does not depend on the number of ranks."""


def rank_samples(sample_begin, sample_end, rank, world):
    """(begin, end, stride) of the samples rank `rank` of `world` renders out of [sample_begin, sample_end)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return sample_begin + rank, sample_end, world


def count_samples(begin, end, stride):
    return 0 if end <= begin else (end - begin + stride - 1) // stride


def render_distributed(render_into, accum, sample_begin, sample_end, rank, world, all_reduce=None):
    """render_into(begin, end, stride, accum) must ADD sum_s trace_s/spp for its samples into `accum`.
    all_reduce(accum) sums the framebuffer over ranks in place (torch.distributed.all_reduce); None = single rank."""
    b, e, s = rank_samples(sample_begin, sample_end, rank, world)
    render_into(b, e, s, accum)
    if world > 1:
        if all_reduce is None:
            raise ValueError("all_reduce is required when world > 1")
        all_reduce(accum)
    return accum
