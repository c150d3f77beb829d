"""Random scenes / variant knobs for fuzzing the HIP path against the oracle."""
import numpy as np

from raytracingpbr_amd import SHAPE, Camera, Config, Material, SDFObject, Scene, Transform
from raytracingpbr_amd.ibl import synthetic_env


def random_case(seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 80)), int(rng.integers(9, 60))
    n = int(rng.integers(1, 12))
    shapes = [SHAPE.SPHERE, SHAPE.BOX, SHAPE.CYLINDER, SHAPE.CONE, SHAPE.PLANE, SHAPE.NONE]
    if seed % 3 == 0:
        shapes = [SHAPE.BOX]                       # exercises the all-boxes specialisation (n != 8 mostly)
    objs = []
    for i in range(n):
        t = shapes[int(rng.integers(0, len(shapes)))]
        pos = rng.uniform(-3, 3, 3)
        rot = rng.uniform(-180, 180, 3) if rng.random() < 0.6 else (0, 0, 0)
        sc = rng.uniform(0.2, 1.5, 3)
        if t == SHAPE.PLANE:
            pos, sc = (0, 0, 0), (0, float(rng.uniform(-3, -1)), 0)
            rot = (0, 0, 0)
        kind = rng.integers(0, 4)
        if kind == 0:
            m = Material(rng.uniform(0.1, 1, 3), (1, 1, 1), 1.0, 0.0, 0.0, float(rng.uniform(1.0, 2.0)))        # diffuse
        elif kind == 1:
            m = Material(rng.uniform(0.5, 1, 3), (1, 1, 1), float(rng.uniform(0, 0.4)), 1.0, 0.0, float(rng.uniform(0.4, 3)))  # metal
        elif kind == 2:
            m = Material(rng.uniform(0.8, 1, 3), (1, 1, 1), float(rng.uniform(0, 0.2)), 0.0, 1.0, float(rng.uniform(1.1, 2.0)))  # glass
        else:
            m = Material((1, 1, 1), rng.uniform(1, 20, 3), 1.0, 0.0, 0.0, 1.0)                                    # light
        objs.append(SDFObject(t, Transform(pos, rot, sc), m))
    cam = Camera(rng.uniform(-1, 1, 3) + np.array([0, 0, 6]), rng.uniform(-0.5, 0.5, 3), (0, 1, 0),
                 float(rng.uniform(25, 60)), W / H, float(rng.uniform(0, 0.1)), float(rng.uniform(2, 7)))
    persistent = seed % 4 == 1
    if persistent:
        cfg = Config.src(W, H, seed, steps_per_launch=int(rng.integers(1, 4)))
        cfg.primary_miss = int(rng.integers(0, 2))
        # self-adaptive sampling (src/config.py:14,17; src/pathtracer.py:97-101; src/postprocessor.py:40-43) on a third of the
        # src/ scenes (its own generator: the scenes of earlier rounds keep their draws); run() then drives the launches the
        # way the reference's render() does — post_process() after every pathtrace() call
        arng = np.random.default_rng(7000 + seed)
        if seed % 3 == 2 or seed % 8 == 5:
            cfg.adaptive_sampling = 1
            cfg.noise_threshold = float(arng.choice([0.02, 0.08, 0.3]))
    else:
        cfg = Config.scene_demo(W, H, seed, int(rng.integers(1, 12)))
        cfg.march_kind = int(rng.integers(0, 2))
        cfg.omega0 = float(rng.choice([1.0, 1.6, 0.5]))
        cfg.omega_guard = int(rng.integers(0, 2))
        cfg.omega_fb_a, cfg.omega_fb_b = (1.0, 0.0) if rng.random() < 0.5 else (0.5, 0.5)
        cfg.hit_eps = float(rng.choice([cfg.hit_eps, 1e-3]))
        cfg.primary_miss = int(rng.integers(0, 3))
        cfg.surface_kind = int(rng.random() < 0.15)
        cfg.camera_kind = int(rng.random() < 0.2)
        cfg.below_horizon = int(rng.integers(0, 2))
        cfg.normal_space = int(rng.integers(0, 2))
        cfg.origin_mode = 0
        cfg.light_quality = float(rng.choice([128.0, 8.0]))
    cfg.nearest_init = int(rng.integers(0, 2))
    cfg.fresnel_kind = int(rng.integers(0, 2))
    cfg.fresnel_roughness_mix = int(rng.integers(0, 2))
    cfg.box_round = float(rng.choice([0.0, 0.01, 0.03]))
    cfg.sky_kind = int(rng.integers(0, 3))
    cfg.tonemap_order = int(rng.integers(0, 4))
    cfg.aces_truncated = int(rng.integers(0, 2))
    cfg.exposure = float(rng.uniform(0.5, 1.5))
    cfg.max_raymarch = int(rng.choice([64, 512]))
    env = synthetic_env(64, 32, seed=seed) if cfg.sky_kind == 1 else None
    return Scene(objs, bool(rng.random() < 0.2), cam, f"fuzz{seed}"), cfg, env, int(rng.integers(1, 7))


def run(r, env, n, persistent):
    if env is not None:
        r.set_env(env, 1.8, 2.2)
    if persistent and r.config.adaptive_sampling:
        # ADAPTIVE_SAMPLING: refresh() initialises the statistics, every launch is followed by post_process() (src/renderer.py:25-32);
        # launches of n, n + 1 and then single steps, so that the mask changes between launches of every size
        r.refresh()
        for k in (n, n + 1, 1, 1, 2, 1, 1):
            r.sample(k)
            r.post_process()
        return r
    r.sample(n)
    if persistent:
        r.sample(n + 1)
    r.post_process()
    return r


def random_box8_case(seed):
    """Eight boxes (the unrolled / signature / squared-distance code paths): overlapping slabs that tie
    exactly, glass boxes (rays inside a box's core), large rounding radii (rays inside the rounding
    shell), single-axis and general rotations, both nearest_init conventions."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(24, 72)), int(rng.integers(16, 48))
    style = seed % 4          # 0: Cornell-like room (fits the listed signature), 1: axis-aligned clutter, 2: single-axis, 3: general
    objs = []
    for i in range(8):
        if style == 0:
            pos = [(0, 0, -1), (0, 1, 0), (0, -1, 0), (-1, 0, 0), (1, 0, 0), (-0.3, -0.3, -0.2), (0.3, -0.5, 0.2), (0, 0.8, 0)][i]
            rot = [(0, 0, 0), (90, 0, 0), (90, 0, 0), (0, 90, 0), (0, 90, 0), (0, float(rng.uniform(-180, 180)), 0),
                   (0, float(rng.uniform(-180, 180)), 0), (90, 0, 0)][i]
            sc = [(1, 1, 0.2)] * 5 + [(0.25, 0.5, 0.25), (0.25, 0.25, 0.25), (0.2, 0.2, 0.01)]
            sc = sc[i]
        else:
            pos = np.round(rng.uniform(-2, 2, 3) * 2) / 2 if style == 1 else rng.uniform(-2, 2, 3)   # half-integer grid: exact ties
            if style == 1:
                rot = tuple(float(90 * rng.integers(-1, 3)) if rng.random() < 0.3 else 0.0 for _ in range(3))
            elif style == 2:
                ax = int(rng.integers(0, 3))
                rot = tuple(float(rng.uniform(-180, 180)) if k == ax else 0.0 for k in range(3))
            else:
                rot = tuple(rng.uniform(-180, 180, 3))
            sc = np.round(rng.uniform(0.2, 1.2, 3) * 4) / 4 if style == 1 else rng.uniform(0.1, 1.2, 3)
        kind = int(rng.integers(0, 4))
        if i == 7 or kind == 3:
            m = Material((1, 1, 1), rng.uniform(5, 40, 3), 1.0, 0.0, 0.0, 1.0)                                    # light
        elif kind == 0:
            m = Material(rng.uniform(0.2, 1, 3), (1, 1, 1), 1.0, 0.0, 0.0, 1.5)                                   # diffuse
        elif kind == 1:
            m = Material(rng.uniform(0.5, 1, 3), (1, 1, 1), float(rng.uniform(0, 0.3)), 1.0, 0.0, 1.5)            # metal
        else:
            m = Material(rng.uniform(0.8, 1, 3), (1, 1, 1), float(rng.uniform(0, 0.1)), 0.0, 1.0, float(rng.uniform(1.1, 1.8)))  # glass
        objs.append(SDFObject(SHAPE.BOX, Transform(pos, rot, sc), m))
    scale10 = style == 0                                       # the examples' x10 scene scale
    cam = Camera((float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), 35.0 if style == 0 else 6.0), (0, 0, 0), (0, 1, 0),
                 float(rng.uniform(30, 50)), W / H, float(rng.uniform(0, 0.05)), 4.0)
    if seed % 5 == 4:
        cfg = Config.src(W, H, seed, steps_per_launch=int(rng.integers(1, 4)))
        cfg.sky_kind = 0                                       # gradient sky: no environment map needed
    else:
        cfg = Config.cornell_v3(W, H, seed, int(rng.integers(2, 10)))
        cfg.march_kind = int(rng.integers(0, 2))
        cfg.omega0 = float(rng.choice([1.0, 1.6, 1.9]))       # 1.9: many over-relaxed steps end inside boxes
        cfg.min_dis = float(rng.choice([0.05, 0.005]))
        cfg.hit_eps = float(rng.choice([cfg.hit_eps, 1e-3]))
    cfg.nearest_init = int(rng.integers(0, 2))
    cfg.box_round = float(rng.choice([0.0, 0.01, 0.1, 0.3]))
    return Scene(objs, scale10, cam, f"box8_{seed}"), cfg, None, int(rng.integers(2, 6))
