#!/usr/bin/env python3
"""Headless Cornell Box: what examples/cornell_box/cornell_box_v3/main.py:10-29 does, minus the window.

    python examples/cornell_box.py --variant v3 --size 1920 1080 --spp 256 --bounces 8 --out cornell.png

The reference loop calls render(camera_position, camera_lookat, camera_up, moving) once per displayed frame (one
sample per pixel each) and shows image_pixels; here `--spp` samples are taken in one call and the tone-mapped frame
is written with the PNG writer that mirrors ti.tools.imwrite (bottom-left origin).  Runs on the HIP library only.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracingpbr_amd import Config, Renderer, cornell_box            # noqa: E402
from raytracingpbr_amd.imageio import imwrite                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="v3", choices=["v1", "v2", "v3", "shortest"])
ap.add_argument("--size", type=int, nargs=2, default=[512, 512])
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--bounces", type=int, default=None, help="MAX_RAYTRACE (script default if omitted)")
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--out", default="cornell_box.png")
a = ap.parse_args()
W, H = a.size
preset = {"v1": Config.cornell_v1, "v2": Config.cornell_v2, "v3": Config.cornell_v3, "shortest": Config.cornell_shortest}[a.variant]
cfg = preset(W, H, a.seed) if a.bounces is None else preset(W, H, a.seed, a.bounces)
scene = cornell_box(a.variant, aspect=W / H) if a.variant != "shortest" else cornell_box("shortest")
r = Renderer(scene, cfg)
t0 = time.time()
r.render(refreshing=True, spp=a.spp)          # moving=True on the first call: clear, sample, tone map
r.sync()
dt = time.time() - t0
imwrite(r.image_pixels, a.out)
c = r.counters()
print(f"{a.out}: {W}x{H}, {a.spp} spp, {W * H * a.spp / dt / 1e6:.0f} Msamples/s wall, "
      f"{c.raycasts / max(c.samples, 1):.2f} raycasts/sample, {c.march_steps / max(c.raycasts, 1):.1f} steps/raycast")
