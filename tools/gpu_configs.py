"""Time every BASELINE.json config on ONE GPU (the multi-GPU ones as rank 0 of G virtual ranks,
i.e. the per-GPU share of the tile-partitioned frame).  Throughput only; parity is in tests/."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracingpbr_amd import SHAPE, Config, Renderer, bunny, cornell_box, src_scene
from raytracingpbr_amd.ibl import load_bunny_weights, synthetic_env
from raytracingpbr_amd.tiles import default_tile

LIST = "--list" in sys.argv          # print the config names (one per line) and exit
only = set(a for a in sys.argv[1:] if a != "--list")
res = {}

def run(name, sc, cfg, spp, env=None, envexp=1.8, tiles=None, chunk=None, warm=1, warm_launches=1, one_step=False):
    if LIST: print(name); return
    if only and name not in only: return
    r = Renderer(sc, cfg)
    if env is not None: r.set_env(env, envexp, 2.2)
    if any(o.type == SHAPE.BUNNY for o in sc.objects): r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())
    if tiles: r.set_tiles(*tiles)
    # what bench.py times unless OPTS says otherwise: the run-time instance of the scene, everything baked
    for k, v in json.loads(os.environ.get("OPTS", '{"jit": 1, "jit_bake": 2}')).items(): r.set_option(k, v)
    for i in range(warm_launches):      # (src/ form: the self-tuned schedule — cost plan, age weights — settles in three launches)
        if i: r.refresh()
        r.sample(warm)
    r.sync()
    if one_step:                        # the way the reference calls it: `spp` launches of ONE bounce-step (src/renderer.py:29-30)
        t0 = time.time(); tr_tot = 0.0
        for i in range(spp):
            r.sample(1)
            if i % 32 == 31: tr_tot += r.last_sample_ms()[0] * 32
        r.sync(); wall = time.time() - t0; c = r.counters()
        out = {"launches": spp, "ms_per_launch_wall": round(wall / spp * 1e3, 4), "kernel_ms_per_launch_sampled": round(tr_tot / spp, 4),
               "Mbounce_steps_per_s": round(cfg.width * cfg.height * spp / wall / 1e6, 1)}
        res[name] = out; print(name, json.dumps(out), flush=True); r.close(); return
    chunk = chunk or spp
    tr_tot, tot_tot, samples, c = 0.0, 0.0, 0, None
    t0 = time.time()
    for _ in range(spp // chunk):
        r.sample(chunk); tr, tot, n = r.last_sample_ms(); c = r.counters()
        tr_tot += tr; tot_tot += tot; samples += c.samples
    wall = time.time() - t0
    out = {"samples": samples, "trace_ms": round(tr_tot, 2), "total_ms": round(tot_tot, 2), "wall_s": round(wall, 3),
           "Msamples_per_s": round(samples / tot_tot / 1e3, 2), "raycasts_per_sample": round(c.raycasts / c.samples, 3),
           "march_steps_per_raycast": round(c.march_steps / max(c.raycasts, 1), 2), "sky_per_sample": round(c.sky_lookups / c.samples, 3)}
    res[name] = out
    print(name, json.dumps(out), flush=True)
    r.close()

env3k = synthetic_env(3072, 1536, seed=0)
run("C1_cornell_256x256_16spp_4b", cornell_box("v3"), Config.cornell_v3(256, 256, 0, 4), 16)
run("C2_cornell_1080p_256spp_8b", cornell_box("v3", aspect=16 / 9), Config.cornell_v3(1920, 1080, 0, 8), 256)
spp3 = int(os.environ.get("C3_SPP", "1024"))
run(f"C3_bunny_glass_1080p_{spp3}spp_16b", bunny(aspect=16 / 9), Config.bunny_glass(1920, 1080, 0, 16), spp3, env=env3k, warm=1)
tw, th = default_tile(3840, 2160, 4)
run("C4_tokyo_ibl_4k_512spp_rank0of4", src_scene(aspect=16 / 9, tokyo=True), Config.tokyo_ibl(3840, 2160, 0, 512), 512, env=env3k, tiles=(tw, th, 0, 4), chunk=256)
tw, th = default_tile(7680, 4320, 8)
spp5 = int(os.environ.get("C5_SPP", "4096"))
run(f"C5_cornell_8k_{spp5}spp_rank0of8", cornell_box("v3", aspect=16 / 9), Config.cornell_v3(7680, 4320, 0, 8), spp5, tiles=(tw, th, 0, 8), chunk=256)
# (a first launch has no cost plan yet, the age-weighted shares settle in three: four warm launches, the profiled one is the fifth)
run("src_768x432_persistent_256steps", src_scene(aspect=768 / 432), Config.src(768, 432, 0, 1), 256, env=env3k, envexp=1.4, warm=256, warm_launches=4)
run("src_1080p_persistent_256steps", src_scene(aspect=16 / 9), Config.src(1920, 1080, 0, 1), 256, env=env3k, envexp=1.4, warm=256, warm_launches=4)
run("src_4k_persistent_256steps", src_scene(aspect=16 / 9), Config.src(3840, 2160, 0, 1), 256, env=env3k, envexp=1.4, warm=256, warm_launches=4)
run("src_1080p_one_step_per_launch", src_scene(aspect=16 / 9), Config.src(1920, 1080, 0, 1), 256, env=env3k, envexp=1.4, warm=128, one_step=True)
run("src_768x432_one_step_per_launch", src_scene(aspect=768 / 432), Config.src(768, 432, 0, 1), 256, env=env3k, envexp=1.4, warm=128, one_step=True)
if not LIST:
    path = os.path.join(ROOT, "gpurun_out", "configs.json")
    if only and os.path.exists(path):            # per-config invocations accumulate into one file
        old = json.load(open(path)); old.update(res); res = old
    json.dump(res, open(path, "w"), indent=1)
