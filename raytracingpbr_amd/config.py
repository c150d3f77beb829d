"""Render configuration: the reference's module constants as one POD struct.

reference: src/config.py:5-28 and the constant block at the top of every example
(e.g. examples/cornell_box/cornell_box_v3/config.py:3-25).  There they are Python globals
consumed through ``ti.static`` at JIT time; here they are the runtime fields of
``rtpbr_config`` (include/rtpbr.h).  ``Config.<preset>()`` reproduces each script's values
(SURVEY.md Appendix B lists every knob and its source line).
"""
import ctypes as C
from enum import IntEnum


class FORM(IntEnum):
    COMPLETE_PATH = 0
    PERSISTENT_RAY = 1


class MARCH(IntEnum):
    PLAIN = 0
    RELAXED = 1
    SRC = 2


class TONEMAP(IntEnum):
    GAMMA_ACES_CLAMP = 0
    ACES_GAMMA = 1
    ACES_CLAMP_GAMMA = 2
    ACES_GAMMA_CLAMP = 3


class SKY(IntEnum):
    BLACK = 0
    ENVMAP = 1
    GRADIENT = 2


class PRIMARY(IntEnum):
    AS_SKY = 0
    BLACK = 1
    WHITE = 2


F32_MAX = 3.4028234663852886e38


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("seed", C.c_uint32),
        ("kernel_form", C.c_int32), ("max_raymarch", C.c_int32), ("max_raytrace", C.c_int32),
        ("march_kind", C.c_int32), ("min_dis", C.c_float), ("max_dis", C.c_float),
        ("hit_eps", C.c_float), ("omega0", C.c_float), ("omega_guard", C.c_int32),
        ("omega_fb_a", C.c_float), ("omega_fb_b", C.c_float),
        ("box_round", C.c_float), ("nearest_init", C.c_int32),
        ("normal_h", C.c_float), ("normal_space", C.c_int32),
        ("rr_kind", C.c_int32), ("light_quality", C.c_float), ("quality_per_sample", C.c_float),
        ("surface_kind", C.c_int32), ("fresnel_kind", C.c_int32), ("fresnel_roughness_mix", C.c_int32),
        ("below_horizon", C.c_int32), ("origin_mode", C.c_int32), ("env_ior", C.c_float),
        ("sky_kind", C.c_int32), ("primary_miss", C.c_int32),
        ("vis_lo", C.c_float), ("vis_hi", C.c_float),
        ("camera_kind", C.c_int32),
        ("tonemap_order", C.c_int32), ("aces_truncated", C.c_int32),
        ("exposure", C.c_float), ("gamma", C.c_float),
        ("frame", C.c_int32), ("steps_per_launch", C.c_int32),
        ("adaptive_sampling", C.c_int32), ("noise_threshold", C.c_float),
        ("anim_bob", C.c_float),
    ]

    def copy(self, **kw):
        c = Config.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise AttributeError(k)
            setattr(c, k, v)
        return c

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    # ------------------------------------------------------------------ presets
    @staticmethod
    def _examples_base(width, height, seed):
        c = Config()
        c.width, c.height, c.seed = width, height, seed
        c.kernel_form = FORM.COMPLETE_PATH
        c.max_raymarch, c.max_raytrace = 512, 3
        c.march_kind = MARCH.RELAXED
        c.min_dis, c.max_dis = 0.05, 2000.0
        c.hit_eps = 0.5 * min(1.0 / width, 1.0 / height)          # PIXEL_RADIUS
        c.omega0, c.omega_guard, c.omega_fb_a, c.omega_fb_b = 1.6, 1, 1.0, 0.0
        c.box_round, c.nearest_init = 0.01, 0
        c.normal_h, c.normal_space = 0.5773 * 0.005, 0
        c.rr_kind, c.light_quality, c.quality_per_sample = 0, 128.0, 0.8
        c.surface_kind, c.fresnel_kind, c.fresnel_roughness_mix = 0, 0, 1
        c.below_horizon, c.origin_mode, c.env_ior = 0, 0, 1.000277
        c.sky_kind, c.primary_miss = SKY.BLACK, PRIMARY.AS_SKY
        c.vis_lo, c.vis_hi = 0.000001, F32_MAX
        c.camera_kind = 0
        c.tonemap_order, c.aces_truncated, c.exposure, c.gamma = TONEMAP.GAMMA_ACES_CLAMP, 0, 1.0, 2.2
        c.frame, c.steps_per_launch = 0, 1
        c.adaptive_sampling, c.noise_threshold = 0, 1e-4          # src/config.py:14,17
        c.anim_bob = 0.0
        return c

    @staticmethod
    def cornell_v3(width=512, height=512, seed=0, max_raytrace=3):
        """examples/cornell_box/cornell_box_v3/config.py:3-25 (the metric scene's variant)."""
        c = Config._examples_base(width, height, seed)
        c.max_raytrace = max_raytrace
        return c

    @staticmethod
    def cornell_v2(width=512, height=512, seed=0, max_raytrace=3):
        """examples/cornell_box/cornell_box_v2.py:7-30,186-196,336-341."""
        c = Config._examples_base(width, height, seed)
        c.max_raytrace = max_raytrace
        c.march_kind, c.hit_eps, c.omega0 = MARCH.PLAIN, 0.001, 1.0
        c.normal_h = 0.001
        c.tonemap_order = TONEMAP.ACES_GAMMA
        return c

    @staticmethod
    def cornell_v1(width=480, height=480, seed=0, max_raytrace=128):
        """examples/cornell_box/cornell_box.py:6-34,213-223 (unit scale, 128 bounces)."""
        c = Config.cornell_v2(width, height, seed, max_raytrace)
        c.min_dis, c.hit_eps, c.normal_h = 0.005, 0.0001, 0.0001
        c.box_round = 0.0
        c.exposure = 0.6
        return c

    @staticmethod
    def cornell_shortest(width=512, height=512, seed=0, max_raytrace=3):
        """examples/cornell_box/cornell_box_shortest.py:34-129 (diffuse only, pinhole)."""
        c = Config._examples_base(width, height, seed)
        c.max_raymarch, c.max_raytrace = 256, max_raytrace
        c.march_kind, c.min_dis, c.hit_eps, c.omega0 = MARCH.PLAIN, 0.0005, 0.00001, 1.0
        c.box_round = 0.0
        c.surface_kind, c.camera_kind, c.aces_truncated = 1, 1, 1
        return c

    @staticmethod
    def scene_demo(width=480, height=270, seed=0, max_raytrace=128):
        """examples/scene_demo/main.py:8-35,200-248 (procedural gradient sky)."""
        c = Config._examples_base(width, height, seed)
        c.max_raytrace, c.min_dis = max_raytrace, 0.005
        c.box_round, c.nearest_init = 0.03, 1
        c.fresnel_kind = 1
        c.sky_kind = SKY.GRADIENT
        c.tonemap_order = TONEMAP.ACES_GAMMA_CLAMP
        return c

    @staticmethod
    def tokyo_ibl(width=2880, height=1620, seed=0, max_raytrace=512):
        """examples/scene_demo/tokyo_ibl.py:9-37,246-265 (decaying omega, env map)."""
        c = Config.scene_demo(width, height, seed, max_raytrace)
        c.omega_guard, c.omega_fb_a, c.omega_fb_b = 0, 0.5, 0.5
        c.sky_kind = SKY.ENVMAP
        return c

    @staticmethod
    def bunny_glass(width=1920, height=1080, seed=0, max_raytrace=512, frame=0):
        """examples/bunny/bunny_sdf_glass.py:9-37,248-267."""
        c = Config._examples_base(width, height, seed)
        c.max_raymarch, c.max_raytrace, c.min_dis = 2048, max_raytrace, 0.005
        c.omega0, c.omega_guard, c.omega_fb_a, c.omega_fb_b = 0.5, 1, 0.4, 0.0
        c.normal_h = 0.0001
        c.light_quality = 512.0
        c.sky_kind = SKY.ENVMAP
        c.tonemap_order, c.exposure = TONEMAP.ACES_CLAMP_GAMMA, 0.8
        c.frame = frame
        c.anim_bob = 0.1                                          # bunny_sdf_glass.py:216  p += vec3(0, 0, 0.1*sin(t))
        return c

    @staticmethod
    def bunny_sdf(width=3840, height=2160, seed=0, max_raytrace=128, frame=0, v2=False):
        """examples/bunny/bunny_sdf.py / bunny_sdf_v2.py (chrome bunny, omega 1.6 -> 0.7)."""
        c = Config.bunny_glass(width, height, seed, max_raytrace, frame)
        c.max_raymarch, c.light_quality = 512, 128.0
        c.omega0, c.omega_fb_a = 1.6, 0.7
        c.primary_miss = PRIMARY.WHITE if v2 else PRIMARY.BLACK
        c.exposure = 0.8 if v2 else 0.6
        c.anim_bob = 0.1 if v2 else 0.0                          # bunny_sdf_v2.py:216 bobs; bunny_sdf.py:213-214 only rotates
        return c

    @staticmethod
    def src(width=768, height=432, seed=0, steps_per_launch=1):
        """src/config.py:7-28 — the persistent-ray library pipeline."""
        c = Config._examples_base(width, height, seed)
        c.kernel_form = FORM.PERSISTENT_RAY
        c.max_raymarch, c.max_raytrace = 512, 512
        c.march_kind = MARCH.SRC
        pixel_radius = 1.0 * min(1.0 / width, 1.0 / height)
        c.hit_eps, c.min_dis, c.max_dis = pixel_radius, 2.5 * pixel_radius, 1e3
        c.box_round, c.nearest_init = 0.03, 1
        c.normal_space = 1
        c.rr_kind = 1
        c.fresnel_kind, c.fresnel_roughness_mix = 1, 0
        c.below_horizon, c.origin_mode = 1, 1
        c.sky_kind = SKY.ENVMAP
        c.vis_lo, c.vis_hi = 1e-4, 1e4
        c.steps_per_launch = steps_per_launch
        return c
