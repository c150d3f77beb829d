"""Timing of small workloads (launch-overhead regime): primary / trace / total per sample() call."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
for (w, h, spp) in ((256, 256, 16), (256, 256, 64), (512, 512, 16), (960, 540, 16), (1920, 1080, 4), (1920, 1080, 16)):
    for opts in ({"primary_split": 0}, {"primary_split": 1}):
        r = Renderer(cornell_box("v3", aspect=w / h), Config.cornell_v3(w, h, 0, 4))
        for k, v in opts.items(): r.set_option(k, v)
        r.sample(spp); r.sync()
        best = None
        for _ in range(5):
            r.sample(spp); tr, tot, n = r.last_sample_ms(); pr, pn = r.last_primary_ms()
            if best is None or tot < best[0]: best = (tot, pr, tr)
        print(w, h, spp, json.dumps(opts), "total %.3f primary %.3f trace %.3f ms  %.0f Msamples/s" % (*best, w * h * spp / best[0] / 1e3), flush=True)
        r.close()
