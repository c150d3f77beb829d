"""Generates tests/golden/bunny_glass_jpg_silhouette.npz: the silhouette of the glass bunny in the reference's committed
result image others/sdf_bunny_glass.jpg (1920x1080, rendered by examples/bunny/bunny_sdf_glass.py with a real HDR
environment that is not part of the repository).

The bunny is in focus and full of sharp refracted detail, the background is blurred by the thin lens: the RMS of the
high-pass residual (luminance minus its sigma = 1 px Gaussian) over a 21 x 21 window separates the two cleanly (the mask
is stable for thresholds 0.006 .. 0.008).  Opening / closing / largest component / hole filling give the silhouette,
which is stored at 480 x 270 as packed bits in the renderer's [x][y] (y up) convention.  The window dilates the outline by
about 2 px at that size; the test erodes accordingly.  Derived DATA of a result artefact, not source code.
Run in the dev container (needs /root/reference, Pillow, SciPy):  python tools/make_jpg_fixture.py
"""
import os

import numpy as np
from PIL import Image
from scipy import ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/others/sdf_bunny_glass.jpg"
im = np.asarray(Image.open(src).convert("RGB"), np.float32) / 255.0
assert im.shape == (1080, 1920, 3), im.shape
lum = im @ np.array([0.299, 0.587, 0.114], np.float32)
hp = lum - ndi.gaussian_filter(lum, 1.0)
e = np.sqrt(ndi.uniform_filter(hp * hp, 21))
areas = {}
for thr in (0.006, 0.007, 0.008):
    m = ndi.binary_closing(ndi.binary_opening(e > thr, iterations=3), iterations=10)
    lab, n = ndi.label(m)
    sizes = ndi.sum(m, lab, range(1, n + 1))
    areas[thr] = ndi.binary_fill_holes(lab == 1 + int(np.argmax(sizes)))
full = areas[0.007]
assert abs(int(areas[0.006].sum()) - int(areas[0.008].sum())) < 0.02 * full.sum()      # threshold-insensitive
small = np.asarray(Image.fromarray((full * 255).astype(np.uint8)).resize((480, 270), Image.BILINEAR)) > 127
xy = np.ascontiguousarray(small[::-1].T)                                             # [x][y], y up
out = os.path.join(ROOT, "tests", "golden", "bunny_glass_jpg_silhouette.npz")
np.savez_compressed(out, bits=np.packbits(xy), shape=np.array(xy.shape), area_fullres=int(full.sum()))
print(out, xy.shape, int(xy.sum()), os.path.getsize(out))
