"""Parity cases shared by the golden-fixture generator, the CPU oracle tests and the GPU
parity tests.  One case per behavioural variant of SURVEY.md Appendix B (every script of
the reference that renders through the hot path), at sizes the oracle finishes in seconds.
"""
import numpy as np

from raytracingpbr_amd import SHAPE, Config, bunny, cornell_box, src_scene
from raytracingpbr_amd.ibl import load_bunny_weights, synthetic_env


class Case:
    def __init__(self, name, scene, cfg, n, env=None, env_exposure=1.0, env_gamma=1.0, rounds=1):
        self.name, self.scene, self.cfg, self.n = name, scene, cfg, n
        self.env, self.env_exposure, self.env_gamma = env, env_exposure, env_gamma
        self.rounds = rounds          # number of sample(n) calls (persistent form: exercises state carry-over)

    def setup(self, r):
        if self.env is not None:
            r.set_env(self.env, self.env_exposure, self.env_gamma)
        if any(o.type == SHAPE.BUNNY for o in self.scene.objects):
            r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())
        return r

    def run(self, r):
        self.setup(r)
        for _ in range(self.rounds):
            r.sample(self.n)
        r.post_process()
        return r


class AdaptiveCase(Case):
    def run(self, r):
        self.setup(r)
        r.refresh()
        for _ in range(self.rounds):
            r.sample(self.n)
            r.post_process()
        return r


def _env(w=192, h=96):
    return synthetic_env(w, h, seed=0)


def all_cases():
    c = []
    # C1: BASELINE.json configs[0] — Cornell 256x256, 16 spp, 4 bounces (the reference's CPU-runnable case)
    c.append(Case("c1_cornell_v3_256_16spp_4b", cornell_box("v3"), Config.cornell_v3(256, 256, 0, 4), 16))
    c.append(Case("cornell_v3_8b_wide", cornell_box("v3", aspect=96 / 54), Config.cornell_v3(96, 54, 3, 8), 8))
    c.append(Case("cornell_v2", cornell_box("v2"), Config.cornell_v2(64, 64, 1, 3), 8))
    c.append(Case("cornell_v1_128b", cornell_box("v1"), Config.cornell_v1(48, 48, 2, 128), 4))
    c.append(Case("cornell_shortest", cornell_box("shortest"), Config.cornell_shortest(64, 64, 4, 3), 8))
    c.append(Case("scene_demo_gradient", src_scene(aspect=96 / 54, tokyo=True), Config.scene_demo(96, 54, 5, 128), 8))
    c.append(Case("tokyo_ibl_env", src_scene(aspect=96 / 54, tokyo=True), Config.tokyo_ibl(96, 54, 6, 64), 8,
                  env=_env(), env_exposure=1.8, env_gamma=2.2))
    c.append(Case("src_persistent", src_scene(aspect=96 / 54), Config.src(96, 54, 7, steps_per_launch=1), 24,
                  env=_env(), env_exposure=1.4, env_gamma=2.2, rounds=2))
    c.append(Case("src_persistent_4steps_blackbg", src_scene(aspect=64 / 36),
                  Config.src(64, 36, 8, steps_per_launch=4).copy(primary_miss=1), 8,
                  env=_env(), env_exposure=1.4, env_gamma=2.2))
    # self-adaptive sampling (ADAPTIVE_SAMPLING=True, src/config.py:14): refresh, then rounds of
    # (pathtrace x n, post_process); pixels whose display value stopped changing drop out
    c.append(AdaptiveCase("src_adaptive_sampling", src_scene(aspect=64 / 36),
                          Config.src(64, 36, 11, steps_per_launch=2).copy(adaptive_sampling=1, noise_threshold=0.2), 6,
                          env=_env(), env_exposure=1.4, env_gamma=2.2, rounds=8))
    c.append(Case("bunny_glass", bunny(aspect=64 / 36), Config.bunny_glass(64, 36, 9, 16, frame=0).copy(max_raymarch=512), 2,
                  env=_env(), env_exposure=1.8, env_gamma=2.2))
    c.append(Case("bunny_chrome_frame30", bunny(aspect=48 / 27, chrome=True),
                  Config.bunny_sdf(48, 27, 10, 8, frame=30), 2, env=_env(), env_exposure=1.8, env_gamma=2.2))
    # the per-frame animation WITH its vertical bob (bunny_sdf_glass.py:216, bunny_sdf_v2.py:216: p.z += 0.1 sin t; frame != 0)
    c.append(Case("bunny_glass_frame17", bunny(aspect=48 / 27), Config.bunny_glass(48, 27, 12, 8, frame=17).copy(max_raymarch=512), 2,
                  env=_env(), env_exposure=1.8, env_gamma=2.2))
    c.append(Case("bunny_chrome_v2_frame45", bunny(aspect=48 / 27, chrome=True, v2=True),
                  Config.bunny_sdf(48, 27, 13, 8, frame=45, v2=True), 2, env=_env(), env_exposure=1.8, env_gamma=2.2))
    return c


def case_by_name(name):
    for c in all_cases():
        if c.name == name:
            return c
    raise KeyError(name)


PROBES = 64


def fingerprint(r):
    """What a golden fixture stores: 8x8-block means of image_buffer, 64 probe pixels at full
    precision (bit patterns), the display image's block means, and the work counters."""
    ib = r.image_buffer
    W, H = ib.shape[:2]
    bw, bh = max(W // 8, 1), max(H // 8, 1)
    blocks = ib[:bw * 8, :bh * 8].astype(np.float64).reshape(8, bw, 8, bh, 4).mean(axis=(1, 3)).astype(np.float32)
    rng = np.random.default_rng(12345)
    px = rng.integers(0, W, PROBES)
    py = rng.integers(0, H, PROBES)
    probes = ib[px, py].view(np.uint32)
    ip = np.nan_to_num(r.image_pixels, nan=-1.0)
    pblocks = ip[:bw * 8, :bh * 8].astype(np.float64).reshape(8, bw, 8, bh, 3).mean(axis=(1, 3)).astype(np.float32)
    c = r.counters()
    ctr = np.array([c.samples, c.raycasts, c.march_steps, c.hits, c.sky_lookups, c.deposits], dtype=np.uint64)
    return {"blocks": blocks, "probes": probes, "pixel_blocks": pblocks, "counters": ctr,
            "checksum": np.array([np.bitwise_xor.reduce(ib.view(np.uint32).reshape(-1).astype(np.uint64))], dtype=np.uint64)}
