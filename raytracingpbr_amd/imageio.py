"""On-disk formats on either side of the path.

The reference reads environment maps with ``ti.tools.imread`` (src/ibl.py:15) and writes
frames with ``ti.tools.imwrite`` (src/main.py:55, bunny_sdf_glass.py:449).  Both go through
stb_image in Taichi; what the hot path relies on is:
  * imread decodes ANY file — including Radiance ``.hdr`` — to **8-bit**: stb converts HDR
    floats to LDR as ``int(255 * v**(1/2.2) + 0.5)`` clamped to [0,255], and returns the array
    as (W, H, C) with the origin at the bottom-left (``swapaxes(0,1)[:, ::-1]``);
  * imwrite takes a (W, H, C) field in [0,1], applies the inverse transpose/flip and writes
    8-bit (``clip(x,0,1)*255 + 0.5``).
This module reproduces that contract without Taichi: a Radiance RGBE reader (flat and
new-style RLE scanlines), the stb HDR->LDR conversion, and PNG/JPEG I/O through Pillow.
"""
import re

import numpy as np


# ---------------------------------------------------------------- Radiance .hdr (RGBE)
def read_hdr(path):
    """Decode a Radiance RGBE file to float32 (H, W, 3), row 0 = top (file order)."""
    with open(path, "rb") as f:
        data = f.read()
    if not (data.startswith(b"#?RADIANCE") or data.startswith(b"#?RGBE")):
        raise ValueError("not a Radiance HDR file")
    pos = data.index(b"\n\n") + 2                       # header ends with an empty line
    header = data[:pos].decode("ascii", "replace")
    if "FORMAT=32-bit_rle_rgbe" not in header:
        raise ValueError("unsupported HDR format (need 32-bit_rle_rgbe)")
    end = data.index(b"\n", pos)
    m = re.match(rb"-Y (\d+) \+X (\d+)", data[pos:end])
    if not m:
        raise ValueError("unsupported HDR orientation (need -Y H +X W)")
    H, W = int(m.group(1)), int(m.group(2))
    p = end + 1
    rgbe = np.empty((H, W, 4), np.uint8)
    buf = np.frombuffer(data, np.uint8)
    for y in range(H):
        if W < 8 or W > 0x7FFF or not (buf[p] == 2 and buf[p + 1] == 2 and (buf[p + 2] & 0x80) == 0):
            # flat scanline (also the fallback stb uses)
            rgbe[y] = buf[p:p + 4 * W].reshape(W, 4)
            p += 4 * W
            continue
        if (int(buf[p + 2]) << 8 | int(buf[p + 3])) != W:
            raise ValueError("corrupt RLE scanline width")
        p += 4
        for c in range(4):
            x = 0
            while x < W:
                n = int(buf[p]); p += 1
                if n > 128:                              # run
                    n -= 128
                    rgbe[y, x:x + n, c] = buf[p]; p += 1
                else:                                    # literal
                    rgbe[y, x:x + n, c] = buf[p:p + n]; p += n
                x += n
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)   # stb: ldexp(1, e - (128+8))
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def write_hdr(path, img, rle=True):
    """Encode float32 (H, W, 3) as Radiance RGBE (used by the tests to make fixtures)."""
    img = np.asarray(img, np.float32)
    H, W, _ = img.shape
    mx = img.max(axis=2)
    rgbe = np.zeros((H, W, 4), np.uint8)
    nz = mx > 1e-32
    m, e = np.frexp(mx[nz])
    s = (m * 256.0 / mx[nz]).astype(np.float32)
    rgbe[nz, :3] = (img[nz] * s[:, None]).astype(np.uint8)
    rgbe[nz, 3] = (e + 128).astype(np.uint8)
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H} +X {W}\n".encode())
    for y in range(H):
        if not rle or W < 8 or W > 0x7FFF:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, W >> 8, W & 255])
        for c in range(4):
            row = rgbe[y, :, c]
            x = 0
            while x < W:
                run = 1
                while x + run < W and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[x])]); x += run
                else:
                    n = 1
                    while x + n < W and n < 128 and not (x + n + 2 < W and row[x + n] == row[x + n + 1] == row[x + n + 2]):
                        n += 1
                    out += bytes([n]) + row[x:x + n].tobytes(); x += n
    with open(path, "wb") as f:
        f.write(bytes(out))


def hdr_to_ldr_stb(img):
    """stb_image's stbi__hdr_to_ldr with its defaults (gamma 2.2, scale 1):
    z = pow(v, 1/2.2) * 255 + 0.5, clamped to [0,255], truncated to uint8."""
    z = np.power(np.maximum(img.astype(np.float32), 0.0), np.float32(1.0 / 2.2)) * np.float32(255.0) + np.float32(0.5)
    return np.clip(z, 0, 255).astype(np.uint8)


def _to_field(img_hwc):
    """(H, W, C) top-down image -> (W, H, C) field with the origin bottom-left."""
    return np.ascontiguousarray(np.swapaxes(img_hwc, 0, 1)[:, ::-1, :])


def _to_image(field_whc):
    return np.ascontiguousarray(np.swapaxes(field_whc, 0, 1)[::-1])


def imread(path):
    """ti.tools.imread equivalent: uint8 (W, H, 3), [x][y] with y = 0 at the bottom; .hdr files
    are tone-mapped to 8 bit exactly like stb does (this is why the scripts re-apply ^2.2)."""
    if str(path).lower().endswith(".hdr"):
        ldr = hdr_to_ldr_stb(read_hdr(path))
    else:
        from PIL import Image
        ldr = np.asarray(Image.open(path).convert("RGB"), np.uint8)
    return _to_field(ldr)


def imwrite(image_pixels, path):
    """ti.tools.imwrite equivalent for a (W, H, 3) float field in [0,1] (or uint8)."""
    from PIL import Image
    a = np.asarray(image_pixels)
    if a.dtype != np.uint8:
        a = (np.clip(np.nan_to_num(a, nan=0.0), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(_to_image(a)).save(path)
