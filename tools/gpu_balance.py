"""Load balance of the tile partition on ONE GPU: for G = 1, 2, 4, 8 virtual ranks, render each rank's share of a config
in turn (same tiles the real N-GPU run deals: tile t -> rank t % G) and record its device time.  max/mean bounds the
tile scaling an N-GPU run can reach: speed-up(G) <= T(1) / max_rank_time(G); the gather (P/G x 16 B per rank over xGMI)
adds < 1 ms.  Writes gpurun_out/balance.json (copied to profiles/<tag>_load_balance.json).

    python tools/gpu_balance.py [c2 c4 c5]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box, src_scene
from raytracingpbr_amd.ibl import synthetic_env
from raytracingpbr_amd.tiles import default_tile

which = set(sys.argv[1:]) or {"c2", "c4", "c5"}
env3k = None
out = {}


def config(name):
    global env3k
    if name == "c2":
        return "Cornell 1920x1080, 256 spp, 8 bounces", cornell_box("v3", 16 / 9), Config.cornell_v3(1920, 1080, 0, 8), 256, None
    if name == "c5":
        return "Cornell 7680x4320, 256 spp (one progressive step of 4096), 8 bounces", cornell_box("v3", 16 / 9), Config.cornell_v3(7680, 4320, 0, 8), 256, None
    if env3k is None:
        env3k = synthetic_env(3072, 1536, seed=0)
    return "Tokyo IBL 3840x2160, 512 spp, 3k env", src_scene(16 / 9, tokyo=True), Config.tokyo_ibl(3840, 2160, 0, 512), 512, env3k


for name in ("c2", "c4", "c5"):
    if name not in which:
        continue
    desc, sc, cfg, spp, env = config(name)
    rec = {"workload": desc, "options": json.loads(os.environ.get("OPTS", "{}")), "ranks": {}}
    for G in (1, 2, 4, 8):
        tw, th = default_tile(cfg.width, cfg.height, G)
        times = []
        for rank in range(G):
            r = Renderer(sc, cfg)
            if env is not None:
                r.set_env(env, 1.8, 2.2)
            if G > 1:
                r.set_tiles(tw, th, rank, G)
            for k, v in json.loads(os.environ.get("OPTS", "{}")).items():      # e.g. OPTS='{"jit": 1, "jit_bake": 1}' = bench.py's kernels
                r.set_option(k, v)
            r.set_option("reserve_spp", spp)
            r.sample(1); r.sync()
            best = 1e9
            for _ in range(2):
                r.refresh(); r.sample(spp)
                best = min(best, r.last_sample_ms()[1])
            times.append(best)
            r.close()
        mean = sum(times) / G
        rec["ranks"][str(G)] = {"tile": [tw, th], "ms_per_rank": [round(t, 2) for t in times], "max_ms": round(max(times), 2),
                                "mean_ms": round(mean, 2), "max_over_mean": round(max(times) / mean, 4)}
    t1 = rec["ranks"]["1"]["max_ms"]
    for G in ("2", "4", "8"):
        rec["ranks"][G]["speedup_bound"] = round(t1 / rec["ranks"][G]["max_ms"], 3)
        rec["ranks"][G]["efficiency_bound"] = round(t1 / rec["ranks"][G]["max_ms"] / int(G), 4)
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "balance.json"), "w"), indent=1)
