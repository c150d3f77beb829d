import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
W, H = 1920, 1080
# run as: RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE python tools/gpu_dbg_phase.py   (the instrumented build of the run-time kernels)
assert "RT_DEBUG_PHASE" in os.environ.get("RTPBR_JIT_EXTRA_FLAGS", ""), "set RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE"
r = Renderer(cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, 8))
r.set_option("jit", 2); r.set_option("jit_bake", 1)
r.sample(64); r.sync()
r.refresh(); r.sample(64); c = r.counters(); tr, tot, n = r.last_sample_ms()
B, D, A = r.counter("mlp_wave_evals"), c.sky_lookups, r.counter("mlp_lane_evals")
ml, ms = c.hits, c.deposits - W * H * 64      # marching lanes summed over wave-steps, wave-steps
print("march lane utilisation %.4f (lanes marching per wave-step / 64), wave-steps per sample %.4f" % (ml / max(ms, 1) / 64, ms / (W * H * 64)))
t = B + D + A
print("trace ms", tr, "phase shares (cycle sums >> 10): shade/refill B %.3f  dispatch %.3f  march A %.3f" % (B / t, D / t, A / t), B, D, A)
