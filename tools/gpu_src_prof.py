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
for _ in range(int(os.environ.get("WARM", "2")) if steps >= 64 else 0):     # the self-tuned schedule (cost plan, age weights) settles in two launches
    r.refresh()
    r.sample(steps)
r.sync()
r.sample(steps)
tr, tot, n = r.last_sample_ms()
c = r.counters()
print(json.dumps({"W": W, "H": H, "scheduler": sched, "steps": steps, "launches": n, "kernel_ms": round(tr, 3),
                  "G_bounce_steps_per_s": round(c.samples / tr / 1e6, 3), "G_march_steps_per_s": round(c.march_steps / tr / 1e6, 2),
                  "raycasts_per_step": round(c.raycasts / c.samples, 4), "march_per_raycast": round(c.march_steps / max(c.raycasts, 1), 3),
                  "hits_per_step": round(c.hits / c.samples, 4), "deposits_per_step": round(c.deposits / c.samples, 4),
                  "jit": r.counter("jit_active"), "plan_heavy": r.counter("plan_heavy"), "plan_total": r.counter("plan_total")}), flush=True)
if "RT_DEBUG_PHASE" in os.environ.get("RTPBR_JIT_EXTRA_FLAGS", ""):
    d = [r.counter("dbg" + "0123456789abcdefghijklmnopqrstuv"[i]) for i in range(32)]
    if sched == 0:
        print(json.dumps({"max_march_steps_of_one_pixel": d[0], "mean_over_waves_of_max_lane": round(d[1] / (W * H / 64), 1),
                          "mean_march_steps_per_pixel": round(c.march_steps / (W * H), 1)}), flush=True)
        raise SystemExit
    waves = max(1, round(W * H / 128 / 4 + 0.5)) * 4
    print(json.dumps({"phase_Mcycles": {"B_and_dispatch": d[0] >> 10, "march": d[2] >> 10, "wave_life_sum": d[3] >> 10},
                      "passes": d[4], "slots_shaded_per_pass": round(d[5] / max(d[4], 1), 2),
                      "march_iters": d[6], "fast_iters": d[15], "tracked_iters": d[1] & 0xffffffff, "full2_iters": d[12],
                      "ok_lanes_per_tracked_iter": round(d[14] / max(d[1] & 0xffffffff, 1), 2),
                      "waiting_lanes_per_tracked_iter": round(d[13] / max(d[1] & 0xffffffff, 1), 2),
                      "cycles_per_iter_in_tracked_loops": round(((d[1] >> 32) << 10) / max((d[1] & 0xffffffff) + d[12], 1)),
                      "cycles_per_plain_iter": round(((d[2] - (d[1] >> 32)) << 10) / max(d[6] - (d[1] & 0xffffffff) - d[12], 1)),
                      "heavy_waves": d[10], "heavy_wave_life_Mcycles_max_mean": [round(d[8] / 1e6, 1), round((d[9] << 10) / max(d[10], 1) / 1e6, 1)],
                      "light_wave_life_Mcycles_max": round(d[11] / 1e6, 1),
                      "lanes_per_march_iter": round(d[7] / max(d[6], 1), 2),
                      "cycles_per_pass": round((d[0] << 10) / max(d[4], 1)), "cycles_per_march_iter": round((d[2] << 10) / max(d[6], 1))}), flush=True)
    if 'RT_DEBUG_PHASE=3' in os.environ.get('RTPBR_JIT_EXTRA_FLAGS', ''):
        print(json.dumps({'B_pass_Mcycles': {'unpack_and_shade': d[28], 'fresh_item_loads': d[29], 'roulette_deposit_regen_writeback': d[30], 'B_pass_whole (without dispatch)': d[0], 'dispatch_swap': d[31], 'march': d[2]}}), flush=True)
    elif 'RT_DEBUG_PHASE=4' in os.environ.get('RTPBR_JIT_EXTRA_FLAGS', ''):
        print(json.dumps({'wave0': {'lean1_exits(max_it,raycast_ended,bound_failed)': d[16:19], 'iterations_by_form(lean1,lean2,tracked,full)': d[20:24],
                                    'steps_by_form': d[24:28]}}), flush=True)
    elif 'RT_DEBUG_PHASE=2' in os.environ.get('RTPBR_JIT_EXTRA_FLAGS', ''):
        print(json.dumps({'light_wave_life_hist_16Mcycle_bins': d[16:32]}), flush=True)
    elif sched == 1 and d[16]:
        print(json.dumps({"wave0": {"life_Mcycles": round(d[16] / 1e6, 1), "B_Mcycles": round(d[17] / 1e6, 1), "march_Mcycles": round(d[18] / 1e6, 1), "march_iters": d[19],
                                    "tracked_iters": d[20], "full2_iters": d[21], "passes": d[22], "tracked_ok_lanesteps": d[23], "lanesteps": d[24], "fast_steps": d[25] & 0xffffffff, "fast_calls": d[25] >> 32, "fast_Mcycles": round(d[30] / 1e6, 1), "cycles_per_tracked_step": round(d[26] / max(d[20], 1)), "cycles_per_full2_step": round(d[27] / max(d[21], 1)),
                                    "per_iter": {"stepping": round(d[24] / max(d[19], 1), 1), "flagged_waiting": round(d[28] / max(d[19], 1), 1), "done_waiting": round(d[29] / max(d[19], 1), 1),
                                                 "parked_ready": round((d[31] & 0xffffffff) / max(d[19], 1), 1), "parked_to_shade": round((d[31] >> 32) / max(d[19], 1), 1)}}}), flush=True)
r.close()
