"""Pin of the oracle (and of the HIP path) against the only OUTPUTS the reference repository holds for this path: the two
1024x1024 renders of its README (img/simple_path_tracing_200samples_6camera.png: simple path tracing, depth 6, 200
samples; img/bidirectional_path_tracing_200samples_3light_3cam.png: bidirectional 3+3, 200 samples; default scene of
kernel.cu:53-137).  They are tone-clamped 8-bit pictures, reduced by their author to a 256-colour palette (PNG mode P),
made with cuRAND and the author's cube.obj (absent from the repository) -- so statistics are kept, not pixels:
  <key>_blocks  (64, 64, 3) float32  means of the 16x16-pixel blocks, per channel, row 0 = top of the picture
  <key>_sat     (64, 64)    float32  fraction of each block's pixels with a saturated (255) channel
  <key>_mean    (3,)        float32  channel means of the whole picture
Run in the build container (reads /root/reference/img as data, PIL): python tests/golden/make_readme_fixture.py"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/img/"
FILES = {"simple_d6_200spp": "simple_path_tracing_200samples_6camera.png",
         "bidir_3_3_200spp": "bidirectional_path_tracing_200samples_3light_3cam.png"}
out = {}
for key, f in FILES.items():
    a = np.asarray(Image.open(SRC + f).convert("RGB")).astype(np.float64)  # row 0 = top of the picture
    assert a.shape == (1024, 1024, 3)
    out[key + "_blocks"] = a.reshape(64, 16, 64, 16, 3).mean(axis=(1, 3)).astype(np.float32)
    out[key + "_sat"] = (a == 255).any(axis=2).reshape(64, 16, 64, 16).mean(axis=(1, 3)).astype(np.float32)
    out[key + "_mean"] = a.mean(axis=(0, 1)).astype(np.float32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_readme_renders.npz"), **out)
print({k: v.shape for k, v in out.items()})
