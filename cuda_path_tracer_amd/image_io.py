"""Image sinks.  write_tga restates SaveTGAFile (kernel.cu:349-371) byte for byte: 18-byte header
(type 2, 24 bpp, descriptor 32 = top-left origin), pixels as B,G,R with int(max(min(v*255.5, 255.5), 0)), and array row
H-1 written first.  write_pfm keeps the fp32 data for lossless comparisons."""
import numpy as np


def tga_bytes(image):
    img = np.asarray(image, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("image must be (H, W, 3)")
    h, w, _ = img.shape
    if w >= 65536 or h >= 65536:
        raise ValueError("TGA dimensions are 16-bit")
    header = bytes([0, 0, 2] + [0] * 9 + [w % 256, w // 256, h % 256, h // 256, 24, 32])
    with np.errstate(invalid="ignore"):
        # fmaxf(fminf(v*255.5, 255.5), 0) (:364): C's fminf/fmaxf return the non-NaN operand, so a NaN pixel saturates
        v = np.fmax(np.fmin(img * np.float32(255.5), np.float32(255.5)), np.float32(0.0))
    v = v.astype(np.int32).astype(np.uint8)                            # (int) truncation
    return header + v[::-1, :, ::-1].tobytes()                        # rows H-1..0, channels B,G,R


def write_tga(path, image):
    with open(path, "wb") as f:
        f.write(tga_bytes(image))


def write_pfm(path, image):
    img = np.asarray(image, dtype=np.float32)
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))   # little-endian, rows bottom-to-top = array row 0 first (row r <-> Y = r)
        f.write(np.ascontiguousarray(img).tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3).astype(np.float32)
