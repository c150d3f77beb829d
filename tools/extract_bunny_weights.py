"""Extracts the 625 numeric weights of the neural bunny SDF as DATA.

Source of the numbers: examples/bunny/bunny_sdf_glass.py:157-201 of the reference (itself a
transcription of https://www.shadertoy.com/view/wtVyWK); only the numeric literals are
read, in order, and stored as float32 in raytracingpbr_amd/data/bunny_weights.npy:
  [0,64)    layer 0: 4 blocks x {wy[4], wz[4], wx[4], bias[4]}
  [64,336)  layer 1: 4 blocks x {4 mat4 row-major (64), bias[4]}
  [336,608) layer 2: same, the sine of this layer is divided by 1.4
  [608,625) output : 4 x vec4, then the constant -0.16
Run in the dev container (needs /root/reference): python tools/extract_bunny_weights.py
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/examples/bunny/bunny_sdf_glass.py"
lines = open(src).read().split("\n")[156:201]
text = "\n".join(lines)
groups = re.findall(r"(?:vec4|mat4)\(([^()]*)\)", text)
vals = []
for g in groups:
    vals += [float(x) for x in g.split(",")]
assert len(vals) == 64 + 272 + 272 + 16, len(vals)
m = re.search(r"\)\)\s*-\s*([0-9.]+)\s*$", lines[-1].strip())
vals.append(-float(m.group(1)))
w = np.array(vals, dtype=np.float32)
assert w.size == 625 and abs(w[-1] + 0.16) < 1e-7
out = os.path.join(ROOT, "raytracingpbr_amd", "data", "bunny_weights.npy")
np.save(out, w)
print(out, w.size, float(np.abs(w).max()))
