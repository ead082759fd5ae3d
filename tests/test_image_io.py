"""CPU: TGA sink restates SaveTGAFile (kernel.cu:349-371) byte for byte; PFM round-trips fp32 exactly."""
import numpy as np

from cuda_path_tracer_amd.image_io import read_pfm, tga_bytes, write_pfm


def test_tga_header_order_and_clamp():
    img = np.zeros((2, 3, 3), np.float32)
    img[0, 0] = [1.0, 0.5, 0.0]        # array row 0 is written LAST (Y = H-1 first, :362)
    img[1, 2] = [2.0, -1.0, 0.25]      # clamps: min(v*255.5, 255.5) -> 255, max(.., 0) -> 0
    img[1, 0] = [np.nan, np.inf, 1 / 255.5]
    b = tga_bytes(img)
    assert list(b[:18]) == [0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 2, 0, 24, 32]
    px = np.frombuffer(b[18:], np.uint8).reshape(2, 3, 3)
    assert list(px[1, 0]) == [0, 127, 255]          # B, G, R of array row 0: int(0.5*255.5) = 127
    assert list(px[0, 2]) == [63, 0, 255]           # int(0.25*255.5) = 63
    assert list(px[0, 0]) == [1, 255, 255]          # NaN -> fminf(NaN, 255.5) = 255.5 -> 255, inf -> 255, 1/255.5*255.5 -> 1
    assert len(b) == 18 + 2 * 3 * 3


def test_pfm_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.random((5, 7, 3), dtype=np.float32) * 10
    write_pfm(tmp_path / "a.pfm", img)
    assert np.array_equal(read_pfm(tmp_path / "a.pfm"), img)
