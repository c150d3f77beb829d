import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, src_scene
from raytracingpbr_amd.ibl import synthetic_env
env = synthetic_env(3072, 1536, seed=0)
for (W, H) in ((768, 432), (1920, 1080), (3840, 2160)):
    for sched in (0, 1):
        r = Renderer(src_scene(aspect=W / H), Config.src(W, H, 0, 1)); r.set_env(env, 1.4, 2.2); r.set_option("scheduler", sched)
        r.sample(8); r.sync()
        r.sample(128); tr, tot, n = r.last_sample_ms(); c = r.counters()
        print(f"{W}x{H} sched={sched} launches={n} ms={tr:.2f} Msteps/s={c.samples / tr / 1e3:.1f} lane-steps/s={c.march_steps / tr / 1e6:.2f}G", flush=True)
        r.close()
