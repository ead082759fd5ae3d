"""Test helper: write small Radiance .hdr files (new-style RLE scanlines, or flat old-style pixels)."""
import numpy as np


def write_hdr(path, rgbe, rle=True):
    """rgbe: uint8 array (h, w, 4).  New RLE needs 8 <= w <= 32767 (HDRLoader.cu:122)."""
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(("-Y %d +X %d\n" % (h, w)).encode())
        for y in range(h):
            row = rgbe[y]
            if not rle:
                f.write(row.tobytes())
                continue
            f.write(bytes([2, 2, (w >> 8) & 255, w & 255]))
            for c in range(4):
                comp = row[:, c]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and comp[i + run] == comp[i]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(comp[i])]))
                        i += run
                    else:
                        j = i
                        lit = []
                        while j < w and len(lit) < 128:
                            if j + 2 < w and comp[j] == comp[j + 1] == comp[j + 2]:
                                break
                            lit.append(int(comp[j]))
                            j += 1
                        if not lit:
                            lit = [int(comp[i])]
                            j = i + 1
                        f.write(bytes([len(lit)] + lit))
                        i = j


def decode_rgbe(rgbe):
    """convertComponent (HDRLoader.cu:99-104): val/256 * 2^(E-128), in float32."""
    e = rgbe[..., 3].astype(np.int32) - 128
    scale = np.ldexp(np.float32(1.0), e).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) / np.float32(256.0)) * scale[..., None]


def random_rgbe(h, w, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    a[..., 3] = rng.integers(120, 136, (h, w), dtype=np.uint8)
    a[:, : w // 3, :] = a[:, :1, :]  # long runs so the RLE path emits run codes
    return a
