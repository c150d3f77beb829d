import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
cfg = Config.cornell_v3(1920, 1080, seed=0, max_raytrace=1)     # primary raycast only
sc = cornell_box("v3", aspect=1920 / 1080)
for ps in (0, 1):
    g = Renderer(sc, cfg); g.set_option("primary_split", ps); g.sample(2); g.sync(); g.sample(64); tr, tot, n = g.last_sample_ms(); c = g.counters()
    print("ps", ps, "ms", tr, "steps/sample", c.march_steps / c.samples, "hits(dbg eval)", c.hits, "sky(dbg wsteps)", c.sky_lookups, "extra boxes evaluated per wave-step", (c.hits - 0) / max(c.sky_lookups, 1))
