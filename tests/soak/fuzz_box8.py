"""One-off soak: many random 8-box scenes, HIP (default options) vs oracle, bit-exact."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz import random_box8_case, run
from oracle_backend import OracleRenderer
from raytracingpbr_amd import Renderer
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    sc, cfg, env, n = random_box8_case(seed)
    o = run(OracleRenderer(sc, cfg), env, n, cfg.kernel_form == 1)
    for opts in ({}, {"primary_split": 2}):
        g = Renderer(sc, cfg)
        for k, v in opts.items(): g.set_option(k, v)
        g = run(g, env, n, cfg.kernel_form == 1)
        same = np.array_equal(np.ascontiguousarray(g.image_buffer).view(np.uint32), np.ascontiguousarray(o.image_buffer).view(np.uint32))
        cg, co = g.counters(), o.counters()
        same = same and (cg.raycasts, cg.march_steps, cg.hits) == (co.raycasts, co.march_steps, co.hits)
        if not same:
            bad += 1
            print("MISMATCH seed", seed, opts, flush=True)
        g.close()
print("seeds", lo, hi, "mismatches", bad)
