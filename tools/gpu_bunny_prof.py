import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import SHAPE, Config, Renderer, bunny
from raytracingpbr_amd.ibl import load_bunny_weights, synthetic_env
env = synthetic_env(3072, 1536, seed=0)
sc = bunny(aspect=16 / 9); cfg = Config.bunny_glass(1920, 1080, 0, 16)
r = Renderer(sc, cfg); r.set_env(env, 1.8, 2.2); r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())
r.sample(2); r.sync(); r.sample(16); r.sync(); print(r.last_sample_ms(), r.counters())
