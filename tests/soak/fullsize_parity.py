"""One-off: the headline frame (Cornell Box 1920x1080, 256 spp, 8 bounces) rendered by the HIP path and by
the CPU oracle, compared bit for bit (image_buffer and the work counters).  ~6 minutes of oracle time on
16 cores; the result is written to gpurun_out/fullsize_parity.json (copied to profiles/)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_backend import OracleRenderer, usable_cores
from raytracingpbr_amd import Config, Renderer, cornell_box
W, H, SPP, B = 1920, 1080, int(os.environ.get("SPP", "256")), 8
cfg = Config.cornell_v3(W, H, 0, B)
sc = cornell_box("v3", aspect=W / H)
g = Renderer(sc, cfg)
for k, v in json.loads(os.environ.get("OPTS", "{}")).items(): g.set_option(k, v)     # e.g. OPTS='{"jit": 2, "jit_bake": 1}' = bench.py's kernels
g.refresh(); t0 = time.time(); g.sample(SPP); g.sync(); tg = time.time() - t0
o = OracleRenderer(sc, cfg, threads=usable_cores()); t0 = time.time(); o.sample(SPP); to = time.time() - t0
a, b = g.image_buffer, o.image_buffer
cg, co = g.counters(), o.counters()
same = bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)))
ctr = lambda c: dict(samples=c.samples, raycasts=c.raycasts, march_steps=c.march_steps, hits=c.hits, sky_lookups=c.sky_lookups, deposits=c.deposits)
out = {"workload": f"Cornell Box v3 {W}x{H}, {SPP} spp, {B} bounces, seed 0", "image_buffer_bit_identical": same,
       "pixels_differing": int((a != b).any(axis=2).sum()), "counters_identical": ctr(cg) == ctr(co), "counters": ctr(cg),
       "hip_seconds": round(tg, 3), "oracle_seconds": round(to, 1), "oracle_threads": usable_cores(),
       "mean_radiance": [float(x) for x in (a[..., :3] / a[..., 3:4]).mean(axis=(0, 1))],
       "options": json.loads(os.environ.get("OPTS", "{}")), "run_time_instance_active": bool(g.counter("jit_active"))}
# further option sets (OPTS2='[{...}, {...}]'): each rendered by the HIP path again and compared with the first render bit for bit
for extra in json.loads(os.environ.get("OPTS2", "[]")):
    g2 = Renderer(sc, cfg)
    for k, v in extra.items(): g2.set_option(k, v)
    g2.refresh(); g2.sample(SPP); g2.sync()
    out.setdefault("further_option_sets", []).append({"options": extra, "image_buffer_bit_identical_to_first": bool(np.array_equal(
        np.ascontiguousarray(g2.image_buffer).view(np.uint32), np.ascontiguousarray(a).view(np.uint32))), "counters_identical": ctr(g2.counters()) == ctr(cg),
        "dense_launches": int(g2.counter("dense_launches"))})
    g2.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fullsize_parity.json"), "w"), indent=1)
print(json.dumps(out))
