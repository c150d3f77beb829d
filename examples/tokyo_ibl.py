#!/usr/bin/env python3
"""Headless scene_demo/tokyo_ibl.py:441-462: sample() x spp, then render() (ACES -> gamma -> clamp), PNG out.

    python examples/tokyo_ibl.py --size 2880 1620 --spp 512 --env assets/Tokyo_BigSight_3k.hdr --out tokyo.png
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracingpbr_amd import Config, Renderer, src_scene              # noqa: E402
from raytracingpbr_amd.ibl import synthetic_env                        # noqa: E402
from raytracingpbr_amd.imageio import imread, imwrite                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=2, default=[960, 540])
ap.add_argument("--spp", type=int, default=128)
ap.add_argument("--env", default=None)
ap.add_argument("--out", default="tokyo_ibl.png")
a = ap.parse_args()
W, H = a.size
r = Renderer(src_scene(aspect=W / H, tokyo=True), Config.tokyo_ibl(W, H, 0))
r.set_env(imread(a.env) if a.env else synthetic_env(3072, 1536), exposure=1.8, gamma=2.2)   # tokyo_ibl.py:59-60
t0 = time.time()
r.sample(a.spp)
r.post_process()
r.sync()
imwrite(r.image_pixels, a.out)
print(f"{a.out}: {W}x{H}, {a.spp} spp in {time.time() - t0:.2f} s")
