"""Instrumented fused src/ launch (-DRT_DEBUG_PHASE=5): the pool kernel's waves grouped by the SIMD they ran on (HW_ID / XCC_ID in the per-wave record).
    python tools/gpu_pool_simd.py W H [KEY=VALUE ...]"""
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
cls = (db[:, 0] >> 32).astype(int); life = db[:, 1] / 1e6; heavy = (db[:, 2] & 1) == 1; own = db[:, 3]; steps = db[:, 6] / 1e3
hw = (db[:, 2] >> 8) & 0xffffffff; xcc = (db[:, 2] >> 40) & 15
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(int)
print(json.dumps({"kernel_ms": round(tr, 3), "waves": len(db), "heavy": int(heavy.sum()), "distinct_simds": int(len(np.unique(key))), "distinct_cus": int(len(np.unique(key // 4)))}))
# light waves by what shares their SIMD
by = {}
for k in np.unique(key):
    m = key == k
    nh = int(heavy[m].sum()); nl = int((~heavy[m]).sum())
    for i in np.nonzero(m & ~heavy)[0]:
        by.setdefault((nh, nl), []).append((life[i], steps[i], cls[i]))
for (nh, nl), v in sorted(by.items()):
    a = np.array(v)
    print(json.dumps({"simd_has_heavy": nh, "simd_light": nl, "light_waves": len(v), "life_pctl(0,50,90,100)": [round(float(x), 1) for x in np.percentile(a[:, 0], [0, 50, 90, 100])],
                      "steps_k_median": round(float(np.median(a[:, 1])), 1)}))
hl = life[heavy]
if len(hl):
    print(json.dumps({"heavy_life_pctl(0,50,90,100)": [round(float(x), 1) for x in np.percentile(hl, [0, 50, 90, 100])], "heavy_own_median": float(np.median(own[heavy]))}))
# per CU: number of pool waves
cnt = np.bincount(key // 4 - (key // 4).min())
print(json.dumps({"waves_per_cu_hist": {int(x): int((cnt == x).sum()) for x in np.unique(cnt)}}))
cs = np.bincount(key - key.min())
print(json.dumps({"waves_per_simd_hist": {int(x): int((cs == x).sum()) for x in np.unique(cs)}}))
# slowest 12 waves
for i in np.argsort(-life)[:12]:
    m = key == key[i]
    print(json.dumps({"life": round(float(life[i]), 1), "cls": int(cls[i]), "heavy": bool(heavy[i]), "own": int(own[i]), "steps_k": round(float(steps[i]), 1),
                      "simd_mates": [[int(cls[j]), bool(heavy[j]), round(float(life[j]), 1), round(float(steps[j]), 1)] for j in np.nonzero(m)[0] if j != i]}))
r.close()
