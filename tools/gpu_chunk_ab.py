import os, sys, json
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
from raytracingpbr_amd import Config, Renderer, cornell_box
for (w, h, spp) in ((1920, 1080, 16), (1920, 1080, 32)):
    rs = []
    vals = (64, 96, 128, 192, 256, 384, 512)
    for ch in vals:
        r = Renderer(cornell_box("v3", aspect=w / h), Config.cornell_v3(w, h, 0, 8))
        r.set_option("jit", 1); r.set_option("jit_bake", 1); r.set_option("chunk", ch)
        r.sample(spp); r.sync(); rs.append(r)
    best = [1e9] * len(vals)
    for rep in range(4):
        for i, r in enumerate(rs):
            r.refresh(); r.sample(spp); best[i] = min(best[i], r.last_sample_ms()[1])
    print(w, h, spp, {v: round(b, 3) for v, b in zip(vals, best)}, flush=True)
