"""C3 (glass bunny 1920x1080, 16 bounces) on the HIP path: throughput per option set and the network's slot utilisation
from the mlp_* counters (parity: tests/test_gpu_parity.py, tests/test_gpu_fullsize.py).   python tools/gpu_bunny.py [spp] ['{"mlp_lanes": 32}' ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracingpbr_amd import SHAPE, Config, Renderer, bunny
from raytracingpbr_amd.ibl import load_bunny_weights, synthetic_env
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [json.loads(a) for a in sys.argv[2:]] or [{}]
env = synthetic_env(3072, 1536, seed=0)
def mk(R, sc, cfg):
    r = R(sc, cfg); r.set_env(env, 1.8, 2.2); r.set_shape_data(SHAPE.BUNNY, load_bunny_weights()); return r
sc = bunny(aspect=16 / 9)
cfg = Config.bunny_glass(1920, 1080, 0, 16)
rs = []
for opts in variants:
    r = mk(Renderer, sc, cfg)
    for k, v in opts.items(): r.set_option(k, v)
    r.sample(1); r.sync(); rs.append(r)
best = [1e9] * len(variants)
for rep in range(2):
    for i, r in enumerate(rs):
        r.sample(spp); tr, tot, n = r.last_sample_ms(); best[i] = min(best[i], tot)
out = []
for opts, b, r in zip(variants, best, rs):
    c = r.counters(); we, le = r.counter("mlp_wave_evals"), r.counter("mlp_lane_evals")
    rec = dict(opts=opts, ms=round(b, 2), Msamples_per_s=round(1920 * 1080 * spp / b / 1e3, 1), mlp_wave_evals_per_sample=round(we / c.samples, 4),
               mlp_lane_evals_per_sample=round(le / c.samples, 3), mlp_lane_utilisation=round(le / max(we * 32, 1), 4),
               raycasts_per_sample=round(c.raycasts / c.samples, 3), steps_per_raycast=round(c.march_steps / c.raycasts, 2))
    out.append(rec); print(json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bunny.json"), "w"), indent=1)
