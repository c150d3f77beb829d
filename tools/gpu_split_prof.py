"""Instrumented one-step launches of the src/ form's wavefront split (RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE): raycast-length
histogram and wave lifetimes of the march kernel.   python tools/gpu_split_prof.py W H [KEY=VALUE ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
W, H = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)
r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1))
r.set_env(synthetic_env(3072, 1536, seed=0), 1.4, 2.2)
r.set_option("jit", 1); r.set_option("jit_bake", 1)
for k, v in opts.items():
    r.set_option(k, int(v))
for _ in range(160):
    r.sample(1)
for i in range(4):
    r.sample(1)
    tr, tot, n = r.last_sample_ms()
    d = [r.counter("dbg" + "0123456789abcdefghijklmnopqrstuv"[i]) for i in range(32)]
    c = r.counters()
    print(json.dumps({"kernel_ms": round(tr, 4), "raycasts": c.raycasts, "march_steps": c.march_steps,
                      "raycasts_by_length(<=16,32,64,128,256,<cap,cap)": d[0:7], "steps_by_length": d[8:15],
                      "longest_wave_kcycles": d[15] >> 10, "iterations_of_busiest_wave": d[7], "plan_heavy": r.counter("plan_heavy"),
                      "raycasts>128_by_list_position(<1k,2k,4k,..)": [x & 0xffffffff for x in d[16:32]], "raycasts>256_by_list_position": [x >> 32 for x in d[16:32]]}), flush=True)
import numpy as np
db = np.ascontiguousarray(r.diff_buffer).view(np.uint64).reshape(-1)[:8192 * 16].reshape(-1, 16)
db = db[db[:, 0] > 0]
t0 = db[:, 0].min()
seq = (db[:, 1] - t0) / 1e3; end = (db[:, 2] - t0) / 1e3; start = (db[:, 0] - t0) / 1e3
it = db[:, 3] & 0xffffffff; its = db[:, 3] >> 32
q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
print(json.dumps({"waves": len(db), "kcycles_pctl(0,10,50,90,99,100)": {"start": q(start), "sequence_done": q(seq), "end": q(end), "tail": q(end - seq)},
                  "iterations": q(it), "iterations_until_sequence_done": q(its),
                  "waves_with_tail>50kcycles": int(((end - seq) > 50).sum()), "tail_kcycles_per_iteration_of_those": round(float(((end - seq)[(end - seq) > 50]).sum() / max(1, (it - its)[(end - seq) > 50].sum())), 3)}))
life = (db[:, 2] - db[:, 0]) / 1e3
top = np.argsort(-life)[:12]
print(json.dumps({"life_kcycles_pctl": q(life), "bulk_kcycles_pctl(start..sequence_done)": q(seq - start),
                  "slowest_waves": [{"life": round(float(life[i]), 1), "bulk": round(float(seq[i] - start[i]), 1), "iters": int(it[i]), "iters_bulk": int(its[i]),
                                     "kcycles_per_bulk_iter": round(float((seq[i] - start[i]) / max(1, its[i])), 2), "kcycles_per_tail_iter": round(float((end[i] - seq[i]) / max(1, it[i] - its[i])), 2),
                                     "fast_calls": int(db[i, 4] & 0xffffffff), "fast_steps": int(db[i, 4] >> 32), "fast2_calls": int(db[i, 7] & 0xffffffff), "fast2_steps": int(db[i, 7] >> 32), "full2": int(db[i, 5] & 0xffff), "op": int((db[i, 5] >> 16) & 0xffff), "tracked": int(db[i, 5] >> 32), "plain": int(db[i, 6] & 0xffffffff), "tail_lanesteps": int(db[i, 6] >> 32)} for i in top]}))
tot = lambda a: int(a.sum())
print(json.dumps({"all_waves": {"iterations": tot(it), "bulk_iterations": tot(its), "lean1_calls": tot(db[:, 4] & 0xffffffff), "lean1_steps": tot(db[:, 4] >> 32), "lean2_calls": tot(db[:, 7] & 0xffffffff),
                                "lean2_steps": tot(db[:, 7] >> 32), "full": tot(db[:, 5] & 0xffff), "op": tot((db[:, 5] >> 16) & 0xffff), "tracked": tot(db[:, 5] >> 32), "plain": tot(db[:, 6] & 0xffffffff),
                                "tail_lanesteps": tot(db[:, 6] >> 32), "tail_kcycles_mean": round(float((end - seq).mean()), 1), "bulk_kcycles_mean": round(float((seq - start).mean()), 1)}}))
fc = db[:, 8:15] & ((1 << 40) - 1); fn = db[:, 8:15] >> 40
print(json.dumps({"tail_by_form(plain,lean1,lean2,tracked,full,op,lanes)": {"calls": [int(x) for x in fn.sum(axis=0)], "Mcycles": [round(float(x) / 1e6, 2) for x in fc.sum(axis=0)],
                                                                      "kcycles_per_call": [round(float(c) / max(1, int(n)) / 1e3, 3) for c, n in zip(fc.sum(axis=0), fn.sum(axis=0))]},
                  "tail_Mcycles_total": round(float((end - seq).sum()) / 1e3, 2)}))
r.close()
