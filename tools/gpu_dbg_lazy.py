import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
W, H = 1920, 1080
r = Renderer(cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, 8))
r.set_option("primary_split", 2)
r.sample(16); c = r.counters()
fallbacks = c.deposits - W * H * 16
print("march lane-steps", c.march_steps, "fallback wave-steps", fallbacks, "=> per 56 lane-steps:", fallbacks / (c.march_steps / 56))
