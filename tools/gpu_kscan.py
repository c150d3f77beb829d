"""Per-sample cost of one launch as a function of spp per launch (1080p Cornell)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
W, H = 1920, 1080
r = Renderer(cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, 8))
for k, v in json.loads(os.environ.get("OPTS", "{}")).items(): r.set_option(k, v)
r.sample(256); r.sync()
for spp in (8, 16, 32, 64, 128, 256):
    best = None
    for _ in range(4):
        r.sample(spp); tr, tot, n = r.last_sample_ms(); pr, pn = r.last_primary_ms()
        if best is None or tot < best[0]: best = (tot, pr, tr, n)
    ns = W * H * spp
    print(spp, "total %.2f primary %.2f trace %.2f ms launches %d | ns/sample: primary %.4f trace %.4f | %.0f Msamples/s"
          % (*best, best[1] / ns * 1e6, best[2] / ns * 1e6, ns / best[0] / 1e3), flush=True)
