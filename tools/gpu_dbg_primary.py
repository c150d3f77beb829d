"""Measurement build (RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PRIMARY): how many objects a wave-step of the coherent primary-ray kernel evaluates
after wave-level culling — the gate for a one-object lean loop in primary_rays.   python tools/gpu_dbg_primary.py [c2|c4|c5] [spp]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Renderer, workloads
from raytracingpbr_amd.tiles import default_tile
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = workloads.get(name, 0, 0, spp)
r = Renderer(wl.scene, wl.cfg); wl.setup(r)
if wl.virtual_world > 1:
    tw, th = default_tile(wl.cfg.width, wl.cfg.height, wl.virtual_world); r.set_tiles(tw, th, 0, wl.virtual_world)
r.set_option("jit", 2); r.set_option("jit_bake", 2); r.set_option("primary_split", 2)
r.sample(spp); r.sync()
h = [r.counter("dbg" + "012345678"[i]) for i in range(9)]
tot = sum(h)
print(json.dumps({"workload": name, "spp": spp, "wave_steps": tot, "fraction_by_objects_evaluated(0..8)": [round(x / max(tot, 1), 4) for x in h],
                  "mean_objects": round(sum(i * x for i, x in enumerate(h)) / max(tot, 1), 3), "primary_ms": round(r.last_primary_ms()[0], 3)}))
r.close()
