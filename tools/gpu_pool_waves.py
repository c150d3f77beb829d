"""Instrumented fused src/ launch (RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE=5): per-wave record of the POOL kernel — residency class, lifetime,
pixels owned, march iterations, passes, march steps — to see which light waves end last.   python tools/gpu_pool_waves.py W H [KEY=VALUE ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
W, H = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)
r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1))
r.set_env(synthetic_env(3072, 1536, seed=0), 1.4, 2.2)
r.set_option("jit", 1); r.set_option("jit_bake", 1)
for k, v in opts.items():
    r.set_option(k, int(v))
r.sample(64)
for _ in range(4):
    r.refresh(); r.sample(256)
r.sync()
tr, tot, n = r.last_sample_ms()
db = np.ascontiguousarray(r.diff_buffer).view(np.uint64).reshape(-1)[4096 * 8:(4096 + 8192) * 8].reshape(-1, 8)
db = db[db[:, 7] == 0x7654321]
cls = (db[:, 0] >> 32).astype(int); life = db[:, 1] / 1e6; heavy = (db[:, 2] & 1) == 1; own = db[:, 3]; iters = db[:, 4]; passes = db[:, 5]; steps = db[:, 6]
q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
out = {"kernel_ms": round(tr, 3), "waves": len(db), "heavy_waves": int(heavy.sum())}
lt = ~heavy
for c in sorted(set(cls[lt])):
    m = lt & (cls == c)
    out[f"light_class_{c}"] = {"waves": int(m.sum()), "life_Mcycles_pctl(0,10,50,90,99,100)": q(life[m]), "pixels_owned": q(own[m])[::5], "march_steps_pctl": q(steps[m] / 1e3),
                               "cycles_per_march_iter_median": round(float(np.median(life[m] * 1e6 / np.maximum(iters[m], 1)))), "corr(life, steps)": round(float(np.corrcoef(life[m], steps[m])[0, 1]), 2)}
print(json.dumps(out, indent=1))
r.close()
