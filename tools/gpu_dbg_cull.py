"""Measurement asked for by the round-2 review before any GPU work on it: with per-lane Lipschitz lower bounds in the POOL kernel's
march loop (secondary rays, 64 unrelated rays per wave), what fraction of the (wave-step, object) evaluations would vanish because
NO marching lane of the wave needs the object?  Instrumented build of the run-time kernels (results unchanged: the culled step is exact).
    RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_CULL python tools/gpu_dbg_cull.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "RT_DEBUG_CULL" in os.environ.get("RTPBR_JIT_EXTRA_FLAGS", ""), "set RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_CULL"
from raytracingpbr_amd import workloads, Renderer
for name, spp in (("c2", 32), ("c4", 16)):
    wl = workloads.get(name, spp=spp)
    r = Renderer(wl.scene, wl.cfg)
    wl.setup(r)
    r.set_option("jit", 2); r.set_option("jit_bake", 1)
    r.sample(spp); r.sync()
    ev, al = r.counter("dbg0"), r.counter("dbg1")
    print(f"{name}: {al / 1e6:.1f} M (wave-step, object) pairs in the pool kernel's march loop, {ev / 1e6:.1f} M needed by some lane: "
          f"{100 * (1 - ev / max(al, 1)):.1f} % would vanish under wave-level culling with per-lane bounds", flush=True)
    r.close()
