"""Generates tests/golden/cornell_taichi_png_blockmeans.npy (K3 statistical fixture).

Input: the reference's committed result image others/cornell_box_taichi.png (512x512 RGB,
display space).  Output: its 16x16 block means as float32 in [0,1], shape (16,16,3), image
orientation (row 0 = top).  This is derived DATA of a result artefact, not source code.
Run in the dev container (needs /root/reference):  python tools/make_png_fixture.py
"""
import os
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/others/cornell_box_taichi.png"
img = np.asarray(Image.open(src).convert("RGB"), dtype=np.float32) / 255.0
assert img.shape == (512, 512, 3), img.shape
bm = img.reshape(16, 32, 16, 32, 3).mean(axis=(1, 3)).astype(np.float32)
out = os.path.join(ROOT, "tests", "golden", "cornell_taichi_png_blockmeans.npy")
np.save(out, bm)
print(out, bm.shape, bm.mean())

# full-resolution copy (uint8, image orientation) for the per-pixel GPU comparison
# (tests/test_gpu_refimage.py): result DATA of the reference, 512x512x3 bytes
out8 = os.path.join(ROOT, "tests", "golden", "cornell_taichi_png_u8.npz")
np.savez_compressed(out8, rgb=np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8))
print(out8, os.path.getsize(out8))
