"""Interleaved A/B of option sets on the headline config (best of N, same process/box)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
cfg = Config.cornell_v3(1920, 1080, seed=0, max_raytrace=8)
sc = cornell_box("v3", aspect=1920 / 1080)
variants = json.loads(os.environ.get("VARIANTS", '[{}]'))
SPP = int(os.environ.get("SPP", "256"))
rs = []
for opts in variants:
    r = Renderer(sc, cfg)
    for k, v in opts.items(): r.set_option(k, v)
    r.sample(4); r.sync(); rs.append(r)
best = [1e9] * len(variants)
parts = [None] * len(variants)
for rep in range(int(os.environ.get("REPS", "3"))):
    for i, r in enumerate(rs):
        r.refresh(); r.sample(SPP); tr, tot, n = r.last_sample_ms()
        if tot < best[i]:
            best[i], parts[i] = tot, (round(r.last_primary_ms()[0], 2), round(tr, 2), n)
for opts, b, pt in zip(variants, best, parts):
    print(json.dumps(opts), f"best total ms={b:.2f} Msamples/s={1920 * 1080 * SPP / b / 1e3:.1f} (primary, trace, launches)={pt}", flush=True)
