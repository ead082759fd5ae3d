"""Comparison with the two renders of the reference's README (tests/golden/reference_readme_renders.npz, made by
tests/golden/make_readme_fixture.py).  An image goes through the TGA sink's conversion (SaveTGAFile, kernel.cu:349-371:
clamp(v * 255.5) to 8 bits, array row H-1 on top) and is reduced to the fixture's 64 x 64 grid of block means.

Blocks are put in three classes:
  (i)   wall / floor / ceiling blocks: no pixel of a cube in or next to them, and at most SAT_FEW of their pixels saturated
        in the reference's picture (at 200 samples a stray clipped firefly sits in most blocks; "no pixel at all" leaves 35 blocks);
  (ii)  blocks a cube touches (the author's cube.obj is absent from the reference's repository: ours is a reconstruction;
        their strongly coloured pixels are also where the 256-colour palette of the published PNGs is coarsest);
  (iii) blocks with saturated pixels (the 255 clamp makes their means depend on the noise of each pixel).
North-star's 1 % bar is held on class (i).  What closed the 2.6-3.7 % gap of round 1: the point on a ray evaluated with
one fused multiply-add per component, as nvcc's default contraction does (oracle.cpp ray_at, DESIGN.md 2, deviation 7)."""
import os

import numpy as np

from cuda_path_tracer_amd.image_io import tga_bytes

FIXTURE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_readme_renders.npz"))
CONFIGS = {"simple_d6_200spp": dict(cam_depth=6, light_depth=0), "bidir_3_3_200spp": dict(cam_depth=3, light_depth=3)}
GRID = 64
SAT_FEW = 0.05  # up to 12 of a block's 256 pixels


def displayed(image, height, width):
    """The picture as SaveTGAFile writes it: uint8 RGB, row 0 = top."""
    img = np.asarray(image, np.float32).reshape(height, width, 3)
    return np.frombuffer(tga_bytes(img)[18:], np.uint8).reshape(height, width, 3)[:, :, ::-1]


def block_stats(image, height, width):
    px = displayed(image, height, width).astype(np.float64)
    bh, bw = height // GRID, width // GRID
    blocks = px.reshape(GRID, bh, GRID, bw, 3).mean(axis=(1, 3))
    sat = (px == 255).any(axis=2).reshape(GRID, bh, GRID, bw).mean(axis=(1, 3))
    return blocks, sat, px.mean(axis=(0, 1))


def cube_blocks(prim_top_down, n_tris):
    """(GRID, GRID) bool: blocks with a mesh-triangle pixel, grown by one block.  prim_top_down: (H, W) closest-hit
    primitive of the pixel-centre camera rays, row 0 = top."""
    h, w = prim_top_down.shape
    m = (prim_top_down < n_tris).reshape(GRID, h // GRID, GRID, w // GRID).any(axis=(1, 3))
    g = m.copy()
    g[1:] |= m[:-1]; g[:-1] |= m[1:]; g[:, 1:] |= m[:, :-1]; g[:, :-1] |= m[:, 1:]
    g[1:, 1:] |= m[:-1, :-1]; g[:-1, :-1] |= m[1:, 1:]; g[1:, :-1] |= m[:-1, 1:]; g[:-1, 1:] |= m[1:, :-1]
    return g


def cube_mask_of(scene, camera, width, n_tris=36):
    """cube_blocks() from the closest hits of the pixel-centre camera rays (Camera::getRay, camera.cuh:17-20) of a
    width x width frame, traced by `scene` (the oracle or the product)."""
    import math
    eye, lookat, vup = [np.asarray(c, np.float64) for c in camera[:3]]
    w = eye - lookat
    f = np.linalg.norm(w)
    right = np.cross(vup, w)
    right = right / np.linalg.norm(right) * f * math.tan(float(camera[3]) / 2)
    up = np.cross(w, right)
    up = up / np.linalg.norm(up) * f * math.tan(float(camera[3]) / 2)
    idx = np.arange(width * width)
    d = lookat + right * (2 * ((idx % width) + 0.5) / width - 1)[:, None] + up * (2 * ((idx // width) + 0.5) / width - 1)[:, None] - eye
    prim = scene.trace_closest(np.broadcast_to(eye, d.shape).astype(np.float32), d.astype(np.float32))[0]
    return cube_blocks(prim.reshape(width, width)[::-1], n_tris)  # array row H-1 is the top row of the picture


def _coarse_rms(diff, ref, mask, group):
    """rms over the (GRID/group)^2 super-blocks of the mean difference of their masked blocks / mean reference level."""
    n = GRID // group
    m = mask.reshape(n, group, n, group)
    cnt = m.sum(axis=(1, 3))
    ok = cnt >= group * group // 4
    d = (diff * mask[..., None]).reshape(n, group, n, group, 3).sum(axis=(1, 3)) / np.maximum(cnt, 1)[..., None]
    r = (ref * mask[..., None]).reshape(n, group, n, group, 3).sum(axis=(1, 3)) / np.maximum(cnt, 1)[..., None]
    return float(np.sqrt((d[ok] ** 2).mean()) / r[ok].mean())


def compare(key, image, height, width, cube_mask=None):
    """-> dict: corr (min over channels of the block-mean correlation), rel_rms (all blocks, 64-pixel-equivalent super-blocks),
    mean_dev (largest relative difference of the channel means) and, with cube_mask, per class: the number of blocks, the
    largest relative difference of the class's channel sums and the super-block rms."""
    blocks, sat, mean = block_stats(image, height, width)
    ref_b, ref_s, ref_m = FIXTURE[key + "_blocks"].astype(np.float64), FIXTURE[key + "_sat"], FIXTURE[key + "_mean"].astype(np.float64)
    diff = blocks - ref_b
    out = {"corr": min(float(np.corrcoef(blocks[..., c].ravel(), ref_b[..., c].ravel())[0, 1]) for c in range(3)),
           "rel_rms": _coarse_rms(diff, ref_b, np.ones((GRID, GRID), bool), 4),
           "mean_dev": float(np.abs(mean / ref_m - 1.0).max())}
    if cube_mask is not None:
        saturated = ref_s > SAT_FEW  # (chosen by the REFERENCE picture alone: a fixed set of blocks, whatever the render under test)
        classes = {"walls": ~cube_mask & ~saturated, "cubes": cube_mask, "saturated": saturated & ~cube_mask}
        for name, m in classes.items():
            n = int(m.sum())
            dev = float(np.abs(blocks[m].sum(axis=0) / ref_b[m].sum(axis=0) - 1.0).max()) if n else 0.0
            out[name] = {"blocks": n, "sum_dev": dev, "rel_rms": _coarse_rms(diff, ref_b, m, 4) if n else 0.0}
    return out
