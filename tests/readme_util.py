"""Comparison with the two renders of the reference's README (tests/golden/reference_readme_renders.npz, made by
tests/golden/make_readme_fixture.py): an image goes through the TGA sink's conversion (SaveTGAFile, kernel.cu:349-371:
clamp(v * 255.5) to 8 bits, array row H-1 on top) and is reduced to the same 16x16 block means."""
import os

import numpy as np

from cuda_path_tracer_amd.image_io import tga_bytes

FIXTURE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_readme_renders.npz"))
CONFIGS = {"simple_d6_200spp": dict(cam_depth=6, light_depth=0), "bidir_3_3_200spp": dict(cam_depth=3, light_depth=3)}


def block_stats(image, height, width):
    img = np.asarray(image, np.float32).reshape(height, width, 3)
    px = np.frombuffer(tga_bytes(img)[18:], np.uint8).reshape(height, width, 3)[:, :, ::-1].astype(np.float64)  # as displayed, RGB
    return px.reshape(16, height // 16, 16, width // 16, 3).mean(axis=(1, 3)), px.mean(axis=(0, 1))


def compare(key, image, height, width):
    """-> (min correlation of the block means over R,G,B; rms of the block-mean differences / reference mean;
    largest relative difference of the channel means)"""
    blocks, mean = block_stats(image, height, width)
    ref_b, ref_m = FIXTURE[key + "_blocks"].astype(np.float64), FIXTURE[key + "_mean"].astype(np.float64)
    corr = min(float(np.corrcoef(blocks[..., c].ravel(), ref_b[..., c].ravel())[0, 1]) for c in range(3))
    rel_rms = float(np.sqrt(((blocks - ref_b) ** 2).mean()) / ref_b.mean())
    return corr, rel_rms, float(np.abs(mean / ref_m - 1.0).max())
