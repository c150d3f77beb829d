"""src/ form, fused launches: sweep of the pool grid (blocks per CU) and residency at small frames — contexts per wave against waves per SIMD.
    python tools/gpu_src_grid.py W H "K=V K=V" "K=V" ...   ("-" = defaults)  ->  one JSON line per option set (kernel ms of 6 launches)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
W, H = int(sys.argv[1]), int(sys.argv[2])
env = synthetic_env(3072, 1536, seed=0)
for s in sys.argv[3:]:
    opts = {} if s == "-" else dict(kv.split("=") for kv in s.split())
    r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1))
    r.set_env(env, 1.4, 2.2)
    r.set_option("jit", 1); r.set_option("jit_bake", 2)
    for k, v in opts.items():
        r.set_option(k, int(v))
    ms = []
    for i in range(int(os.environ.get('NL', 10))):
        r.refresh()
        r.sample(256)
        ms.append(round(r.last_sample_ms()[0], 2))
    print(json.dumps({"W": W, "H": H, "opts": opts, "kernel_ms": ms, "median_last6": sorted(ms[-6:])[2:4], "G_median": round(W * H * 256 / (sum(sorted(ms[-6:])[2:4]) / 2) / 1e6, 3)}), flush=True)
    r.close()
