"""Debug build (-DRT_DEBUG_LAZY): how often a wave-step falls back from nearest_boxes_lazy to the exact expression on the
headline frame.  RTPBR_HIP_LIB=<debug .so> python tools/gpu_dbg_lazy.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
W, H = 1920, 1080
r = Renderer(cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, 8))
r.set_option("primary_split", 2)
r.sample(16); c = r.counters()
t2 = r.counter("mlp_lane_evals")
print("march lane-steps", c.march_steps, "fallback wave-steps", t2, "=> per wave-step (~54 lanes):", t2 / (c.march_steps / 54))
