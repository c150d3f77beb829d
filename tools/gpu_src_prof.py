"""One `src/`-form (persistent-ray) run for profiling: python tools/gpu_src_prof.py W H scheduler [steps] [KEY=VALUE ...]
Prints one JSON line: HIP-event kernel time, bounce-steps/s, march-steps/s, path statistics."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env

W, H, sched = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
steps = int(rest[0]) if rest and "=" not in rest[0] else 256
opts = dict(kv.split("=") for kv in rest if "=" in kv)
env = synthetic_env(3072, 1536, seed=0)
r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1))
r.set_env(env, 1.4, 2.2)
r.set_option("scheduler", sched)
r.set_option("jit", int(opts.pop("jit", 1)))
r.set_option("jit_bake", int(opts.pop("jit_bake", 1)))
for k, v in opts.items():
    r.set_option(k, int(v))
r.sample(64)
r.sync()
r.sample(steps)
tr, tot, n = r.last_sample_ms()
c = r.counters()
print(json.dumps({"W": W, "H": H, "scheduler": sched, "steps": steps, "launches": n, "kernel_ms": round(tr, 3),
                  "G_bounce_steps_per_s": round(c.samples / tr / 1e6, 3), "G_march_steps_per_s": round(c.march_steps / tr / 1e6, 2),
                  "raycasts_per_step": round(c.raycasts / c.samples, 4), "march_per_raycast": round(c.march_steps / max(c.raycasts, 1), 3),
                  "hits_per_step": round(c.hits / c.samples, 4), "deposits_per_step": round(c.deposits / c.samples, 4),
                  "jit": r.counter("jit_active")}), flush=True)
if "RT_DEBUG_PHASE" in os.environ.get("RTPBR_JIT_EXTRA_FLAGS", ""):
    d = [r.counter("dbg%d" % i) for i in range(8)]
    if sched == 0:
        print(json.dumps({"max_march_steps_of_one_pixel": d[0], "mean_over_waves_of_max_lane": round(d[1] / (W * H / 64), 1),
                          "mean_march_steps_per_pixel": round(c.march_steps / (W * H), 1)}), flush=True)
        raise SystemExit
    waves = max(1, round(W * H / 128 / 4 + 0.5)) * 4
    print(json.dumps({"phase_Mcycles": {"B_and_dispatch": d[0] >> 10, "march": d[2] >> 10, "wave_life_sum": d[3] >> 10},
                      "passes": d[4], "slots_shaded_per_pass": round((d[5] & ((1 << 40) - 1)) / max(d[4], 1), 2),
                      "objects_evaluated_per_sparse_iter": round((d[5] >> 40) / max(d[1] & 0xffffffff, 1), 2), "march_iters": d[6], "sparse_iters": d[1] & 0xffffffff, "cycles_per_sparse_iter": round(((d[1] >> 32) << 10) / max(d[1] & 0xffffffff, 1)),
                      "cycles_per_dense_iter": round(((d[2] - (d[1] >> 32)) << 10) / max(d[6] - (d[1] & 0xffffffff), 1)),
                      "lanes_per_march_iter": round(d[7] / max(d[6], 1), 2),
                      "cycles_per_pass": round((d[0] << 10) / max(d[4], 1)), "cycles_per_march_iter": round((d[2] << 10) / max(d[6], 1))}), flush=True)
r.close()
