"""Environment map ("IBL") side of the path.

reference: src/ibl.py:12-40 — ``Image(path)`` loads an equirectangular picture with
``ti.tools.imread`` (uint8, shape (W,H,3), origin bottom-left — SURVEY.md D3), divides by 255
and ``process(exposure, gamma)`` re-linearises it as ``(c*exposure)**gamma``; lookups are
nearest-texel.  The HDR assets of the reference are absent from the checkout
(.MISSING_LARGE_BLOBS), so ``synthetic_env`` provides a deterministic procedural stand-in
of the same shape and dtype.
"""
import numpy as np

BUNNY_WEIGHTS_FILE = "bunny_weights.npy"


def _hash_u32(x):
    x = x.astype(np.uint64)
    x = (x ^ (x >> 16)) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> 15)) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> 16)
    return x


def synthetic_env(width=3072, height=1536, seed=0):
    """Deterministic procedural equirect map, uint8 (W,H,3), [x][y] with y=0 at the bottom.

    Vertical sky/ground gradient + one bright "sun" disc (clipped at 255) + coarse value noise
    from a fixed integer hash.  Only integer and exactly-rounded arithmetic is used, so the
    texels are identical on every machine."""
    x = np.arange(width, dtype=np.int64)[:, None]
    y = np.arange(height, dtype=np.int64)[None, :]
    v = (2 * y + 1) * 1024 // (2 * height)                      # 0..1023 bottom -> top
    up = np.clip(v - 512, 0, 512)                               # sky part
    dn = np.clip(512 - v, 0, 512)                               # ground part
    r = 96 + (up * 40) // 512 + (dn * 20) // 512
    g = 112 + (up * 60) // 512 - (dn * 24) // 512
    b = 128 + (up * 120) // 512 - (dn * 64) // 512
    r = np.broadcast_to(r, (width, height)).copy()
    g = np.broadcast_to(g, (width, height)).copy()
    b = np.broadcast_to(b, (width, height)).copy()
    # value noise on a 64x32 cell grid, +-16 levels
    cell = (x * 64 // width) * 4096 + (y * 32 // height) + (seed * 7919 + 1) * 65537
    n = (_hash_u32(np.broadcast_to(cell, (width, height)).copy()) & np.uint64(31)).astype(np.int64) - 16
    r += n
    g += n
    b += n
    # sun: ellipse in texture space centred at (0.68 W, 0.72 H)
    cx, cy = (68 * width) // 100, (72 * height) // 100
    rx, ry = max(width // 96, 1), max(height // 48, 1)
    inside = ((x - cx) * (x - cx)) * (ry * ry) + ((y - cy) * (y - cy)) * (rx * rx) < (rx * rx) * (ry * ry)
    inside = np.broadcast_to(inside, (width, height))
    for ch in (r, g, b):
        ch[inside] = 255
    img = np.stack([r, g, b], axis=2)
    return np.clip(img, 0, 255).astype(np.uint8)


def preprocess(texels_u8, exposure, gamma):
    """Image.__init__ + Image.process (src/ibl.py:14-23, postprocessor.adjust :17-21):
    ((c/255) * exposure) ** gamma in float32 — what rtpbr_set_env does for RGB8 input."""
    c = texels_u8.astype(np.float32) / np.float32(255.0)
    return np.power(c * np.float32(exposure), np.float32(gamma)).astype(np.float32)


def load_bunny_weights():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", BUNNY_WEIGHTS_FILE)).astype(np.float32)
