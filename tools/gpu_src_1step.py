"""src/ form called the way the reference calls it: ONE bounce-step per launch (src/renderer.py:29-30), N launches.
    python tools/gpu_src_1step.py W H [N] [KEY=VALUE ...]   ->  ms per launch, G bounce-steps/s"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
W, H = int(sys.argv[1]), int(sys.argv[2])
rest = sys.argv[3:]
N = int(rest[0]) if rest and "=" not in rest[0] else 256
opts = dict(kv.split("=") for kv in rest if "=" in kv)
cfg = Config.src(W, H, 0, 1)
if "MAX_RAYMARCH" in os.environ:       # timing experiment only (changes the image): how much of a launch is its longest raycast?
    cfg.max_raymarch = int(os.environ["MAX_RAYMARCH"])
r = Renderer(src_scene(aspect=W / H), cfg)
r.set_env(synthetic_env(3072, 1536, seed=0), 1.4, 2.2)
r.set_option("jit", 1); r.set_option("jit_bake", 1)
for k, v in opts.items():
    r.set_option(k, int(v))
for _ in range(128):
    r.sample(1)
r.sync()
t0 = time.perf_counter()
ker = 0.0
for i in range(N):
    r.sample(1)
    if i % 32 == 31:
        ker += r.last_sample_ms()[0]
r.sync()
dt = time.perf_counter() - t0
print(json.dumps({"W": W, "H": H, "launches": N, "opts": opts, "ms_per_launch_wall": round(dt / N * 1e3, 4), "kernel_ms_sampled": round(ker / (N // 32), 4),
                  "G_bounce_steps_per_s": round(W * H * N / dt / 1e9, 3)}))
r.close()
