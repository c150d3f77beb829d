"""src/ form, fused launches in a row: how the self-tuned schedule (cost plan, age-weighted shares) settles.
    python tools/gpu_src_conv.py W H [N] [KEY=VALUE ...]  ->  kernel ms of each of N launches of 256 bounce-steps"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
W, H = int(sys.argv[1]), int(sys.argv[2])
rest = sys.argv[3:]
N = int(rest[0]) if rest and "=" not in rest[0] else 8
opts = dict(kv.split("=") for kv in rest if "=" in kv)
r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1))
r.set_env(synthetic_env(3072, 1536, seed=0), 1.4, 2.2)
r.set_option("jit", 1); r.set_option("jit_bake", 2)
for k, v in opts.items():
    r.set_option(k, int(v))
ms = []
for i in range(N):
    r.refresh()
    r.sample(256)
    ms.append(round(r.last_sample_ms()[0], 2))
print(json.dumps({"W": W, "H": H, "opts": opts, "kernel_ms_per_launch": ms, "G_bounce_steps_per_s_last": round(W * H * 256 / ms[-1] / 1e6, 3)}))
r.close()
