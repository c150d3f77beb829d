"""The BASELINE.json workloads as data: scene, configuration, environment, unit of work and the algorithmic FLOP
model SURVEY.md section 8(d) prescribes for each.  Used by bench.py, tools/gpu_configs.py and the full-size tests, so that
every place that says "C3" renders the same thing.

Reference scripts the entries stand for: C1/C2/C5 examples/cornell_box/cornell_box_v3 (config.py:3-25, scene.py:6-27),
C3 examples/bunny/bunny_sdf_glass.py, C4 examples/scene_demo/tokyo_ibl.py, `src` the library pipeline src/main.py
(src/config.py:7-28, src/scene.py:11-33).  The environment maps the scripts load are not in the repository
(.MISSING_LARGE_BLOBS), so a deterministic synthetic equirect of the same size stands in (ibl.synthetic_env).
"""
from dataclasses import dataclass, field

from .config import Config
from .dataclass import SHAPE
from .ibl import load_bunny_weights, synthetic_env
from .scene import bunny, cornell_box, src_scene

# FLOPs per march step / per normal / per surface interaction / camera ray / roulette + bookkeeping / sky
F_CAMERA, F_SURFACE, F_RR, F_SKY = 110.0, 110.0, 25.0, 15.0
F_STEP = {"cornell": 338.0, "tokyo": 242.0}
F_NORMAL = {"world": 197.0, "local": 155.0}
F_MLP = 1450.0 + 48.0        # one evaluation of the bunny network (+ its 48 sines)
F_BUNNY_CHEAP = 50.0         # a march step outside the unit sphere: transform, rotation, length, compare


@dataclass
class Workload:
    name: str
    title: str                # goes into bench.py's config.workload
    scene: object
    cfg: Config
    spp: int                  # samples per pixel (complete-path form) or bounce-steps per pixel (src form) of one bench step
    unit: str = "Msamples/s"
    env: tuple = None         # (width, height, exposure, gamma)
    bunny: bool = False
    family: str = "cornell"   # FLOP model
    virtual_world: int = 1    # the config's GPU count (C4: 4, C5: 8): rank 0's share is what ONE GPU renders of it
    opts: dict = field(default_factory=dict)

    @property
    def short(self):
        return {"cornell": "Cornell Box", "bunny": "SDF glass bunny", "tokyo": "Tokyo IBL scene_demo", "src": "src/ persistent-ray pipeline"}[self.family]

    def setup(self, r):
        """everything a Renderer (HIP or oracle) needs beyond scene and config"""
        if self.env is not None:
            w, h, ex, ga = self.env
            r.set_env(synthetic_env(w, h, seed=0), ex, ga)
        if self.bunny:
            r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())

    def flop_per_unit(self, c, mlp_lane_evals=0):
        """algorithmic FLOPs per sample (per bounce-step in the src form) from the work counters of a run (SURVEY 8(d):
        F_camera + B (S F_step + F_normal + F_surface + F_rr) + P_sky F_sky, B and S measured)"""
        n = max(c.samples, 1)
        if self.family == "bunny":
            tot = (c.samples * F_CAMERA + c.march_steps * F_BUNNY_CHEAP + mlp_lane_evals * F_MLP
                   + c.raycasts * (F_SURFACE + F_RR) + c.hits * 40.0 + c.sky_lookups * F_SKY)
            return tot / n
        if self.family == "src":
            tot = (c.samples * F_RR + c.deposits * (F_CAMERA + 8.0) + c.march_steps * F_STEP["tokyo"]
                   + c.hits * (F_NORMAL["local"] + F_SURFACE) + c.sky_lookups * F_SKY)
            return tot / n
        fs = F_STEP[self.family]
        B = c.raycasts / n
        S = c.march_steps / max(c.raycasts, 1)
        return F_CAMERA + B * (S * fs + F_NORMAL["world"] + F_SURFACE + F_RR) + (c.sky_lookups / n) * F_SKY


def get(name, width=0, height=0, spp=0, bounces=0):
    """c1..c5 = BASELINE.json configs[0..4]; src = the library pipeline at 1920x1080, 256 bounce-steps per step."""
    if name in ("c1", "c2", "c5"):
        W, H, K, B = {"c1": (256, 256, 16, 4), "c2": (1920, 1080, 256, 8), "c5": (7680, 4320, 256, 8)}[name]
        W, H, K, B = width or W, height or H, spp or K, bounces or B
        t = f"Cornell Box (cornell_box_v3 variant) {W}x{H}, {K} spp, {B} bounces, seed 0"
        if name == "c5":
            t += " (configs[4]'s frame; its 4096 spp are 16 such progressive steps)"
        return Workload(name, t, cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, B), K,
                        virtual_world=8 if name == "c5" else 1)
    if name == "c3":
        W, H, K, B = width or 1920, height or 1080, spp or 1024, bounces or 16
        return Workload(name, f"SDF glass bunny (bunny_sdf_glass variant, neural SDF) {W}x{H}, {K} spp, {B} bounces, 3072x1536 synthetic env, seed 0",
                        bunny(aspect=W / H), Config.bunny_glass(W, H, 0, B, frame=0), K, env=(3072, 1536, 1.8, 2.2), bunny=True,
                        family="bunny")
    if name == "c4":
        W, H, K, B = width or 3840, height or 2160, spp or 512, bounces or 512
        return Workload(name, f"Tokyo IBL scene_demo (tokyo_ibl variant, 7 objects) {W}x{H}, {K} spp, MAX_RAYTRACE {B}, 3072x1536 synthetic env, seed 0",
                        src_scene(aspect=W / H, tokyo=True), Config.tokyo_ibl(W, H, 0, B), K, env=(3072, 1536, 1.8, 2.2),
                        family="tokyo", virtual_world=4)
    if name == "src":
        W, H, K = width or 1920, height or 1080, spp or 256
        return Workload(name, f"src/ persistent-ray pipeline (src/pathtracer.py: one bounce-step per pixel and launch) {W}x{H}, "
                              f"{K} launches fused, 7 objects, 3072x1536 synthetic env, seed 0",
                        src_scene(aspect=W / H), Config.src(W, H, 0, 1), K, unit="Mbounce-steps/s", env=(3072, 1536, 1.4, 2.2),
                        family="src")
    raise KeyError(name)


NAMES = ("c1", "c2", "c3", "c4", "c5", "src")
