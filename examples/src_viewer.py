#!/usr/bin/env python3
"""Headless src/main.py:16-66: the persistent-ray library pipeline with the smoothed camera.

A scripted "user" moves the camera for a few frames; SmoothCamera.update eases the rendered pose towards it and
reports `moving`, and render(refreshing or moving) clears the accumulation exactly when the reference does
(src/camera.py:82-112, src/renderer.py:25-32).  One pathtrace() launch per frame, like SAMPLES_PER_FRAME = 1.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracingpbr_amd import Config, Renderer, src_scene                               # noqa: E402
from raytracingpbr_amd.camera import SmoothCamera, render_interactive_frame            # noqa: E402
from raytracingpbr_amd.ibl import synthetic_env                                        # noqa: E402
from raytracingpbr_amd.imageio import imread, imwrite                                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=2, default=[768, 432])
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--env", default=None)
ap.add_argument("--out", default="src_viewer.png")
a = ap.parse_args()
W, H = a.size
r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0))
r.set_env(imread(a.env) if a.env else synthetic_env(3072, 1536), exposure=1.4, gamma=2.2)   # src/ibl.py:32-33
smooth = SmoothCamera().init((0, -0.2, 4.0))                                           # src/main.py:16-18
target = (0.0, -0.2, 4.0)
refreshes = 0
for f in range(a.frames):
    if f == 30:
        target = (0.6, 0.1, 3.6)                     # the user steps aside: the pose eases over the next frames
    moving = smooth.update(1.0 / 60.0, target, (0, 0, 1), (0, 1, 0))
    refreshes += moving
    render_interactive_frame(r, smooth, refreshing=(f == 0))
imwrite(r.image_pixels, a.out)
print(f"{a.out}: {a.frames} frames, {refreshes} of them refreshed while the camera moved, "
      f"{int(r.image_buffer[..., 3].max())} samples accumulated since it came to rest")
