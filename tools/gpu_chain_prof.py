"""Instrumented fused launch of the src/ form with the chain kernel (RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE): per-wave record of
the chain kernel — pixels, lifetime, march cycles, steps, iterations by form of the tracked march.   python tools/gpu_chain_prof.py W H [KEY=VALUE ...]"""
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
for _ in range(3):
    r.refresh(); r.sample(256)
r.sync()
tr, tot, n = r.last_sample_ms()
db = np.ascontiguousarray(r.diff_buffer).view(np.uint64).reshape(-1)[:2048 * 8].reshape(-1, 8)
db = db[db[:, 7] == 0x1234567]
k = db[:, 0] & 0xffffffff; life = db[:, 1] / 1e6; march = db[:, 2] / 1e6; tot_steps = db[:, 3]; lane0 = db[:, 0] >> 32
print(json.dumps({"kernel_ms_events": round(tr, 3), "chain_waves": len(db), "pixels": int(k.sum()), "waves_by_pixels": {int(x): int((k == x).sum()) for x in np.unique(k)},
                  "life_Mcycles_pctl(0,50,90,100)": [round(float(x), 1) for x in np.percentile(life, [0, 50, 90, 100])]}))
one = np.nonzero(k == 1)[0][:3]
for i in list(np.argsort(-life)[:5]) + list(one):
    f1, s1 = int(db[i, 4] & 0xffffffff), int(db[i, 4] >> 32); f2, s2 = int(db[i, 5] & 0xffffffff), int(db[i, 5] >> 32); f3, f4 = int(db[i, 6] & 0xffffffff), int(db[i, 6] >> 32)
    it = f1 + f2 + f3 + f4
    print(json.dumps({"wave_pixels": int(k[i]), "life_Mcycles": round(float(life[i]), 1), "march_Mcycles": round(float(march[i]), 1), "lane_steps_total": int(tot_steps[i]), "lane0_steps": int(lane0[i]),
                      "iterations": it, "lean1_calls/steps": [f1, s1], "lean2_calls/steps": [f2, s2], "tracked": f3, "full": f4,
                      "march_cycles_per_step_iteration": round(float(db[i, 2]) / max(1, s1 + s2 + f3 + f4))}))
r.close()
