"""Throughput of one workload in both flavours: python tools/gpu_fast_ab.py c2|c3|c4 [spp] [KEY=VALUE ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Renderer, workloads
from raytracingpbr_amd.tiles import default_tile
name = sys.argv[1]
rest = sys.argv[2:]
spp = int(rest[0]) if rest and "=" not in rest[0] else 256
opts = dict(kv.split("=") for kv in rest if "=" in kv)
wl = workloads.get(name, 0, 0, spp)
for prec in (0, 1):
    r = Renderer(wl.scene, wl.cfg)
    wl.setup(r)
    if wl.virtual_world > 1:
        tw, th = default_tile(wl.cfg.width, wl.cfg.height, wl.virtual_world)
        r.set_tiles(tw, th, 0, wl.virtual_world)
    r.set_option("jit", 1); r.set_option("jit_bake", 2); r.set_option("precision", prec)
    for k, v in opts.items():
        r.set_option(k, int(v))
    r.set_option("reserve_spp", spp)
    r.sample(spp); r.sync()
    best = 1e9
    for _ in range(3):
        r.refresh()
        t0 = time.perf_counter(); r.sample(spp); r.sync(); dt = time.perf_counter() - t0
        best = min(best, dt)
    c = r.counters()
    tr, tot, n = r.last_sample_ms()
    print(json.dumps({"workload": name, "precision": prec, "spp": spp, "Msamples_per_s": round(c.samples / best / 1e6, 1), "wall_ms": round(best * 1e3, 2),
                      "trace_ms": round(tr, 2), "primary_ms": round(r.last_primary_ms()[0], 2), "total_kernels_ms": round(tot, 2), "deposits": c.deposits, "samples": c.samples}), flush=True)
    r.close()
