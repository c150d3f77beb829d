#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY: pin the CPU oracle to the reference's own code.

Taichi cannot be installed here, so this script executes the reference's own @ti.func /
@ti.kernel bodies (read from /root/reference at run time, nothing is copied) on top of the
minimal stand-in runtime in tools/ti_standin/ (NOT Taichi: NumPy float32 vector algebra with
Taichi's typing rules, see its docstring), with ti.random() replaced by the repo's counter-based
stream, and writes NUMBERS ONLY — inputs and the reference functions' outputs — to
tests/golden/ref_*.npz.  tests/test_oracle_refpin.py then checks oracle/rt_oracle.c against those
fixtures on any machine (the GPU box has neither /root/reference nor this stand-in's inputs).

    python tools/ref_crosscheck.py v3      # examples/cornell_box/cornell_box_v3/*.py   (v3b8, v3b8_wide: MAX_RAYTRACE 8; 1920x1080)
    python tools/ref_crosscheck.py src     # src/*.py (persistent-ray form)
    python tools/ref_crosscheck.py bunny   # examples/bunny/bunny_sdf_glass.py (sd_bunny, raycast)
    python tools/ref_crosscheck.py v2 | v1 | shortest | scene_demo | tokyo | bunny_glass | bunny_sdf | bunny_sdf_v2
                                           # the single-file example scripts, each through its own kernels
    python tools/ref_crosscheck.py all     # each of the above in its own interpreter

What runs is the reference's code: functions are called directly with inputs chosen here, or
observed in situ (a recording wrapper around a module-level name) while the reference's own
render kernel runs on a subset of pixels.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RTPBR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0
QUICK = int(os.environ.get("XCHECK_QUICK", "0"))     # debugging: shrink the in-situ pixel grids
M32 = 0xFFFFFFFF


# ------------------------------------------------------------------ the repo's RNG (oracle/rt_oracle_math.h)
def mix32(z):
    z ^= z >> 16
    z = (z * 0x85EBCA6B) & M32
    z ^= z >> 13
    z = (z * 0xC2B2AE35) & M32
    z ^= z >> 16
    return z


def rng_key(seed, x, y, sample):
    k = mix32((seed + 0x9E3779B9) & M32)
    k = mix32(k ^ ((x | (y << 16)) & M32))
    k = mix32(k ^ (sample & M32))
    return k


class Stream:
    """draw number n of the stream keyed by (seed, pixel x, pixel y, sample)"""

    def __init__(self):
        self.key, self.n, self.px, self.py, self.sample = 0, 0, 0, 0, 0

    def seek(self, px, py, sample, n=0):
        self.px, self.py, self.sample = int(px), int(py), int(sample)
        self.key = rng_key(SEED, self.px, self.py, self.sample)
        self.n = n

    def __call__(self):
        z = mix32((self.key + self.n * 0x9E3779B9) & M32)
        self.n += 1
        return (z >> 8) * 5.9604644775390625e-8


def install_standin():
    sys.path.insert(0, os.path.join(ROOT, "tools", "ti_standin"))
    import taichi as ti
    from taichi import _rt
    assert "STAND-IN" in ti.__doc__
    return ti, _rt


def vec_np(v):
    return np.array(v._d, dtype=np.float32)


def f32(x):
    return np.float32(x)


class Recorder:
    def __init__(self):
        self.rows = {}

    def add(self, table, **kw):
        t = self.rows.setdefault(table, {})
        for k, v in kw.items():
            t.setdefault(k, []).append(v)

    def arrays(self):
        out = {}
        for table, cols in self.rows.items():
            for k, v in cols.items():
                out[f"{table}__{k}"] = np.array(v)
        return out


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    arrays = dict(arrays)
    arrays["meta"] = np.array(repr(meta))
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: v.shape for k, v in arrays.items() if k != "meta"})


def grid_pixels(W, H, nx, ny):
    if QUICK:
        nx, ny = QUICK, QUICK
    xs = [int((i + 0.5) * W / nx) for i in range(nx)]
    ys = [int((j + 0.5) * H / ny) for j in range(ny)]
    return [(x, y) for x in xs for y in ys]


# =================================================================== Cornell Box v3
def run_v3(max_raytrace=None, out="ref_v3.npz", grid=(32, 32), SPP=4, resolution=None):
    """max_raytrace: override of the config constant MAX_RAYTRACE (the BASELINE configs use 4 / 8 bounces with the
    same functions; cornell_box_v3/config.py:15 ships 3).  resolution: override of config.py:3 image_resolution — the
    reference's OWN config module is executed with that one line replaced (its text is read here at run time, never
    stored), so SCREEN_PIXEL_SIZE, PIXEL_RADIUS and aspect_ratio are derived by its own expressions (BASELINE configs[1]
    is this scene at 1920x1080: 16:9 side margins, primary misses, outer wall faces)."""
    ti, _rt = install_standin()
    vdir = os.path.join(REF, "examples", "cornell_box", "cornell_box_v3")
    sys.path.insert(0, vdir)
    if resolution is not None:
        import re
        import types
        text = open(os.path.join(vdir, "config.py")).read()
        text, n = re.subn(r"(?m)^image_resolution\s*=.*$", "image_resolution = (%d, %d)" % tuple(resolution), text)
        assert n == 1, "cornell_box_v3/config.py: image_resolution line not found"
        mod = types.ModuleType("config")
        mod.__file__ = os.path.join(vdir, "config.py")
        sys.modules["config"] = mod
        exec(compile(text, mod.__file__, "exec"), mod.__dict__)
    import config, scene, sdf, util, pbr, pathtracer, postprocessor, renderer   # noqa: E401  (the reference's modules)
    from taichi.math import vec2, vec3, vec4

    if max_raytrace is not None:
        pathtracer.MAX_RAYTRACE = max_raytrace            # the module-level constant raytrace() loops to
    rng = Stream()
    _rt.rng = rng
    rec = Recorder()
    rs = np.random.RandomState(1234)
    W, H = config.image_resolution
    meta = dict(variant="cornell_v3", width=W, height=H, max_raytrace=pathtracer.MAX_RAYTRACE, seed=SEED,
                max_raymarch=pathtracer.MAX_RAYMARCH, generator="tools/ref_crosscheck.py v3")
    objs = [scene.objects[i] for i in range(scene.objects_num)]
    pos_of = [tuple(vec_np(o.transform.position).tolist()) for o in objs]

    def obj_index(o):
        return pos_of.index(tuple(vec_np(o.transform.position).tolist()))

    # ---- pure functions with inputs chosen here -------------------------------------------------
    P = rs.uniform(-14, 14, size=(400, 3)).astype(np.float32)
    B = rs.uniform(0.1, 10, size=(400, 3)).astype(np.float32)
    rec.rows["sd_box"] = dict(p=list(P), b=list(B), out=[f32(sdf.sd_box(vec3(*p), vec3(*b))) for p, b in zip(P, B)])
    for i, o in enumerate(objs):
        # rotation matrix the reference builds inside every signed_distance (util.angle)
        rec.add("angle", obj=i, out=util.angle(ti.math.radians(o.transform.rotation)).to_numpy())
    Q = rs.uniform(-13, 13, size=(150, 3)).astype(np.float32)
    for i, o in enumerate(objs):
        for q in Q:
            rec.add("signed_distance", obj=i, p=q, out=f32(sdf.signed_distance(o, vec3(*q))))
    C = np.concatenate([rs.uniform(0, 4, size=(60, 3)), rs.uniform(0, 0.05, size=(20, 3)), np.zeros((1, 3)), np.ones((1, 3))]).astype(np.float32)
    for c in C:
        rec.add("aces", inp=c, out=vec_np(postprocessor.ACESFitted(vec3(*c))))
    for c in C:
        n = f32(rs.randint(1, 300))
        buf = np.array([c[0] * n, c[1] * n, c[2] * n, n], dtype=np.float32)
        rec.add("post_process", buffer=buf, out=vec_np(postprocessor.post_process(vec4(*buf))))
    for x in np.linspace(-1, 0, 21).astype(np.float32):
        for F0, r in ((0.04, 0.0), (0.0877, 1.0), (0.5, 0.3)):
            rec.add("fresnel", NoI=x, F0=f32(F0), rough=f32(r), out=f32(pbr.fresnel_schlick(f32(x), f32(F0), f32(r))))

    # ---- in situ: the reference's own render kernel on a pixel subset ----------------------------
    pixels = grid_pixels(W, H, *grid)
    _rt.pixels = lambda field: pixels if len(field.shape) == 2 else None
    state = dict(sample=0, cur=None, steps=0)

    def on_index(ix):
        if len(ix) == 2:
            rng.seek(ix[0], ix[1], state["sample"])
    _rt.on_index = on_index

    o_get_ray, o_raytrace = renderer.get_ray, renderer.raytrace
    o_raycast, o_rsi, o_nearest = pathtracer.raycast, pathtracer.ray_surface_interaction, pathtracer.nearest_object
    o_normal = pbr.calc_normal

    def get_ray(c, uv, color):
        r = o_get_ray(c, uv, color)
        state["cur"] = dict(px=rng.px, py=rng.py, sample=rng.sample, ro=vec_np(r.origin), rd=vec_np(r.direction),
                            raycasts=0, steps=0)
        return r

    def raytrace(ray):
        out = o_raytrace(ray)
        c = state["cur"]
        rec.add("samples", px=c["px"], py=c["py"], sample=c["sample"], ro=c["ro"], rd=c["rd"],
                color=vec_np(out.color), raycasts=c["raycasts"], steps=c["steps"], draws=rng.n)
        return out

    def nearest_object(p):
        state["steps"] += 1
        return o_nearest(p)

    def raycast(ray):
        state["steps"] = 0
        r = o_raycast(ray)
        c = state["cur"]
        c["raycasts"] += 1
        c["steps"] += state["steps"]
        rec.add("raycasts", px=c["px"], py=c["py"], sample=c["sample"], ro=vec_np(ray.origin), rd=vec_np(ray.direction),
                hit=bool(r.hit), pos=vec_np(r.position), obj=obj_index(r.object), dist=f32(r.object.distance),
                steps=state["steps"])
        return r

    def calc_normal(obj, p):
        n = o_normal(obj, p)
        state["normal"] = vec_np(n)
        return n

    def ray_surface_interaction(ray, record):
        n0 = rng.n
        cin, din = vec_np(ray.color), vec_np(ray.direction)
        out = o_rsi(ray, record)
        c = state["cur"]
        rec.add("surface", px=c["px"], py=c["py"], sample=c["sample"], n0=n0, n1=rng.n, obj=obj_index(record.object),
                pos=vec_np(record.position), dir_in=din, color_in=cin, normal=state["normal"],
                dir_out=vec_np(out.direction), color_out=vec_np(out.color), origin_out=vec_np(out.origin))
        return out

    renderer.get_ray, renderer.raytrace = get_ray, raytrace
    pathtracer.raycast, pathtracer.ray_surface_interaction, pathtracer.nearest_object = raycast, ray_surface_interaction, nearest_object
    pbr.calc_normal = calc_normal

    cam = ti.ui.Camera()
    cam.position(0, 0, 3.5 * 10)                       # what cornell_box_v3/main.py:12-13 does
    meta["camera"] = dict(position=[0, 0, 35.0], lookat=list(map(float, vec_np(cam.curr_lookat))), up=list(map(float, vec_np(cam.curr_up))))
    t0 = time.time()
    for s in range(SPP):
        state["sample"] = s
        renderer.render(cam.curr_position, cam.curr_lookat, cam.curr_up, s == 0)
        print(f"v3 frame {s}: {time.time() - t0:.0f} s", flush=True)
    ib, ip = scene.image_buffer.to_numpy(), scene.image_pixels.to_numpy()
    px = np.array(pixels)
    arrays = rec.arrays()
    arrays["frame__pixels"] = px
    arrays["frame__image_buffer"] = ib[px[:, 0], px[:, 1]]
    arrays["frame__image_pixels"] = ip[px[:, 0], px[:, 1]]
    meta["spp"] = SPP
    save(out, arrays, meta)


# =================================================================== src/ persistent-ray form
def synthetic_env_u8(w, h):
    """deterministic (W,H,3) uint8 image, [x][y], y up — DATA chosen here (the reference's asset is absent)"""
    x = np.arange(w)[:, None, None]
    y = np.arange(h)[None, :, None]
    c = np.arange(3)[None, None, :]
    v = (x * 7 + y * 13 + c * 29 + (x * y) % 31) % 200 + 30
    v = np.where((abs(x - w // 4) < 3) & (abs(y - 3 * h // 4) < 3), 255, v)
    return v.astype(np.uint8)


def run_src(adaptive=False, out="ref_src.npz", K=24, grid=(24, 16), noise_threshold=None, spp=1, black=False):
    """adaptive: the reference's ADAPTIVE_SAMPLING = True (src/config.py:14) — the constants are changed in its config
    module before any of its other modules is imported, exactly what editing that line does"""
    ti, _rt = install_standin()
    sys.path.insert(0, REF)
    rng = Stream()
    _rt.rng = rng
    EW, EH = 64, 32
    env = synthetic_env_u8(EW, EH)
    _rt.imread = lambda path: env.copy()
    import src.config as config                          # noqa: E402  (the reference's package)
    if adaptive:
        config.ADAPTIVE_SAMPLING = True
        if noise_threshold is not None:
            config.NOISE_THRESHOLD = noise_threshold
    config.SAMPLES_PER_PIXEL = spp                        # bounce-steps per launch (src/config.py:10)
    config.BLACK_BACKGROUND = black                       # src/config.py:13
    import src.scene as scene
    import src.sdf as sdf
    import src.pbr as pbr
    import src.util as util
    import src.ibl as ibl
    import src.camera as camera
    import src.fileds as fileds
    import src.pathtracer as pathtracer
    import src.renderer as renderer
    import src.postprocessor as postprocessor
    import src.aces as aces
    from taichi.math import vec2, vec3, vec4

    scene.build_scene()                                  # src/main.py does this before the frame loop
    rec = Recorder()
    rs = np.random.RandomState(4321)
    W, H = config.image_resolution
    meta = dict(variant="src", width=W, height=H, seed=SEED, env_w=EW, env_h=EH, generator="tools/ref_crosscheck.py src",
                max_raytrace=config.MAX_RAYTRACE, pixel_radius=float(config.PIXEL_RADIUS), min_dis=float(config.MIN_DIS))
    n_obj = scene.objects.shape[0]
    objs = [scene.objects[i] for i in range(n_obj)]
    rec.rows["objects"] = dict(type=[int(o.type) for o in objs], position=[vec_np(o.transform.position) for o in objs],
                               scale=[vec_np(o.transform.scale) for o in objs], matrix=[o.transform.matrix.to_numpy() for o in objs],
                               albedo=[vec_np(o.material.albedo) for o in objs], emission=[vec_np(o.material.emission) for o in objs],
                               rmti=[np.array([o.material.roughness, o.material.metallic, o.material.transmission, o.material.ior], dtype=np.float32) for o in objs])
    pos_of = [tuple(vec_np(o.transform.position).tolist()) for o in objs]

    def obj_index(o):
        return pos_of.index(tuple(vec_np(o.transform.position).tolist()))

    # ---- pure functions --------------------------------------------------------------------------
    P = rs.uniform(-3, 3, size=(200, 3)).astype(np.float32)
    S = rs.uniform(0.1, 2, size=(200, 3)).astype(np.float32)
    for shape in sdf.SHAPE:
        fn = sdf.SHAPE_FUNC[shape]
        rec.rows[f"sd_{shape.name.lower()}"] = dict(p=list(P), s=list(S), out=[f32(fn(vec3(*p), vec3(*s))) for p, s in zip(P, S)])
    Q = rs.uniform(-4, 6, size=(300, 3)).astype(np.float32)
    for q in Q:
        i, d = scene.nearest(vec3(*q))
        rec.add("nearest", p=q, index=int(i), dist=f32(d))
    for a in rs.uniform(-200, 200, size=(20, 3)).astype(np.float32):
        rec.add("rotate", deg=a, out=util.rotate(ti.math.radians(vec3(*a))).to_numpy())
    D = rs.normal(size=(200, 3)).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True).astype(np.float32)
    for d in D:
        rec.add("spherical_map", d=d, uv=vec_np(util.sample_spherical_map(vec3(*d))))
    C = np.concatenate([rs.uniform(0, 4, size=(60, 3)), rs.uniform(0, 0.05, size=(20, 3)), np.ones((1, 3))]).astype(np.float32)
    for c in C:
        rec.add("aces", inp=c, out=vec_np(aces.ACESFitted(vec3(*c))))
    meta["env_processed"] = "ibl.hdr_map.img after Image.process(1.4, 2.2)"
    env_ref = ibl.hdr_map.img.to_numpy()                  # (EW,EH,3) after the reference's own preprocess

    # ---- in situ: pathtrace() on a pixel subset, K launches ----------------------------------------
    pixels = grid_pixels(W, H, *grid)
    _rt.pixels = lambda field: pixels if len(field.shape) == 2 else None
    state = dict(step=0, steps=0)

    def on_index(ix):
        if len(ix) == 2:
            state["sub"] = 0
            rng.seek(ix[0], ix[1], state["step"] * spp)
    _rt.on_index = on_index
    o_rr = pathtracer.russian_roulette

    def russian_roulette(ray, i, j):                      # one bounce-step = one stream, keyed by the absolute step index
        rng.seek(i, j, state["step"] * spp + state["sub"])
        if spp != 1:                                      # the ray state each bounce-step starts from (the launch records only its end)
            rec.add("steps", px=i, py=j, step=state["step"] * spp + state["sub"],
                    ray=np.concatenate([vec_np(ray.origin), vec_np(ray.direction), vec_np(ray.color), [np.float32(int(ray.depth))]]).astype(np.float32))
        state["sub"] += 1
        return o_rr(ray, i, j)
    pathtracer.russian_roulette = russian_roulette

    o_raycast, o_rsi, o_gen, o_sky = pathtracer.raycast, pathtracer.ray_surface_interaction, pathtracer.gen_ray, pathtracer.sky_color
    o_nearest = scene.nearest
    o_normal = pbr.calc_normal

    def nearest(p):
        state["steps"] += 1
        return o_nearest(p)

    def raycast(ray):
        state["steps"] = 0
        rin = (vec_np(ray.origin), vec_np(ray.direction), int(ray.depth))
        r, obj, hit = o_raycast(ray)
        rec.add("raycasts", px=rng.px, py=rng.py, step=rng.sample, ro=rin[0], rd=rin[1], depth_in=rin[2], hit=bool(hit),
                origin_out=vec_np(r.origin), obj=obj_index(obj), depth_out=int(r.depth), steps=state["steps"])
        return r, obj, hit

    def calc_normal(obj, p):
        n = o_normal(obj, p)
        state["normal"] = vec_np(n)
        return n

    def ray_surface_interaction(ray, obj):
        n0 = rng.n
        cin, din, oin = vec_np(ray.color), vec_np(ray.direction), vec_np(ray.origin)
        out = o_rsi(ray, obj)
        rec.add("surface", px=rng.px, py=rng.py, step=rng.sample, n0=n0, n1=rng.n, obj=obj_index(obj), origin_in=oin,
                dir_in=din, color_in=cin, normal=state["normal"], dir_out=vec_np(out.direction),
                color_out=vec_np(out.color), origin_out=vec_np(out.origin))
        return out

    def gen_ray(uv):
        r = o_gen(uv)
        rec.add("gen_ray", px=rng.px, py=rng.py, step=rng.sample, uv=vec_np(uv), ro=vec_np(r.origin), rd=vec_np(r.direction))
        return r

    def sky_color(ray):
        c = o_sky(ray)
        rec.add("sky", d=vec_np(ray.direction), color=vec_np(c))
        return c

    pathtracer.raycast, pathtracer.ray_surface_interaction = raycast, ray_surface_interaction
    pathtracer.gen_ray, pathtracer.sky_color = gen_ray, sky_color
    scene.nearest = nearest
    pbr.calc_normal = calc_normal

    # camera pose: src/main.py:16-18 (ti.ui.Camera defaults + position(0,-0.2,4)) -> smooth.init(camera)
    cam = ti.ui.Camera()
    cam.position(0, -0.2, 4)
    camera.smooth.init(cam)
    meta["camera"] = dict(position=[0, -0.2, 4.0], lookat=list(map(float, vec_np(cam.curr_lookat))), up=list(map(float, vec_np(cam.curr_up))),
                          aspect=float(camera.aspect_ratio[None]), vfov=float(camera.camera_vfov[None]),
                          aperture=float(camera.camera_aperture[None]), focus=float(camera.camera_focus[None]))
    px = np.array(pixels)
    hist_ray, hist_img, hist_pix, hist_dbuf, hist_dpix = [], [], [], [], []
    t0 = time.time()
    for k in range(K):
        state["step"] = k
        renderer.render(k == 0)                           # src/renderer.py:25-32: refresh on the first call
        rb = np.zeros((len(pixels), 10), dtype=np.float32)
        dep = np.zeros(len(pixels), dtype=np.int32)
        for n, (x, y) in enumerate(pixels):
            r = fileds.ray_buffer[x, y]
            rb[n, 0:3], rb[n, 3:6], rb[n, 6:9] = vec_np(r.origin), vec_np(r.direction), vec_np(r.color)
            dep[n] = int(r.depth)
        hist_ray.append(np.concatenate([rb[:, :9], dep[:, None].astype(np.float32)], axis=1))
        hist_img.append(fileds.image_buffer.to_numpy()[px[:, 0], px[:, 1]])
        hist_pix.append(fileds.image_pixels.to_numpy()[px[:, 0], px[:, 1]])
        if adaptive:
            hist_dbuf.append(fileds.diff_buffer.to_numpy()[px[:, 0], px[:, 1]])
            hist_dpix.append(fileds.diff_pixels.to_numpy()[px[:, 0], px[:, 1]])
        print(f"src launch {k}: {time.time() - t0:.0f} s", flush=True)
    arrays = rec.arrays()
    if adaptive:                                          # the per-function observations are in ref_src.npz already
        arrays = dict(frame__diff_buffer=np.array(hist_dbuf), frame__diff_pixels=np.array(hist_dpix))
        meta["adaptive_sampling"], meta["noise_threshold"] = 1, float(config.NOISE_THRESHOLD)
    elif spp != 1 or black:
        arrays = {k: v for k, v in arrays.items() if k.startswith("steps__")}
    meta["steps_per_launch"], meta["black_background"] = spp, int(black)
    arrays.update(frame__pixels=px, frame__ray_buffer=np.array(hist_ray), frame__image_buffer=np.array(hist_img),
                  frame__image_pixels=np.array(hist_pix), env__u8=env, env__processed=env_ref)
    meta["launches"] = K
    save(out, arrays, meta)


# =================================================================== bunny (sd_bunny + one raycast set)
def run_bunny():
    ti, _rt = install_standin()
    rng = Stream()
    _rt.rng = rng
    _rt.pixels = lambda field: [] if len(field.shape) == 2 else None   # the script's module-level 241-frame loop renders nothing
    env = synthetic_env_u8(64, 32)
    _rt.imread = lambda path: env.copy()
    sys.path.insert(0, os.path.join(REF, "examples", "bunny"))
    import bunny_sdf_glass as m                          # noqa: E402  (the reference's script)
    from taichi.math import vec3
    rec = Recorder()
    rs = np.random.RandomState(99)
    P = np.concatenate([rs.uniform(-1, 1, size=(600, 3)), rs.uniform(-1.5, 1.5, size=(100, 3))]).astype(np.float32)
    for p in P:
        rec.add("sd_bunny", p=p, out=f32(m.sd_bunny(vec3(*p))))
    obj = m.objects[0]
    for frame in (0, 30, 77):
        m.u_frame[None] = frame
        for p in P[:200]:
            rec.add("signed_distance", frame=frame, p=p, out=f32(m.signed_distance(obj, vec3(*p))))
    m.u_frame[None] = 0
    # raycasts towards the bunny from the camera side
    cnt = dict(n=0)
    o_near = m.nearest_object

    def nearest_object(p):
        cnt["n"] += 1
        return o_near(p)
    m.nearest_object = nearest_object
    for _ in range(120):
        ro = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.3, 0.3), 4.0], dtype=np.float32)
        tgt = rs.uniform(-0.7, 0.7, size=3).astype(np.float32)
        rd = tgt - ro
        rd = (rd / np.sqrt((rd * rd).sum())).astype(np.float32)
        cnt["n"] = 0
        r = m.raycast(m.Ray(vec3(*ro), vec3(*rd), vec3(1)))
        rec.add("raycasts", ro=ro, rd=rd, hit=bool(r.hit), pos=vec_np(r.position), steps=cnt["n"])
    meta = dict(variant="bunny_glass", width=m.image_resolution[0], height=m.image_resolution[1], max_raymarch=m.MAX_RAYMARCH,
                generator="tools/ref_crosscheck.py bunny")
    save("ref_bunny.npz", rec.arrays(), meta)


# =================================================================== single-file example scripts
# Each script's module-level GUI loop falls through (the stand-in window is never running; the
# bunny scripts' unconditional frame loops run with an empty pixel set), then the script's own
# kernels are called the way its main loop calls them.
def run_script(tag, subdir, modname, cam_pos, step, grid, calls, env=False, frame=0, finish=None, per_call=None):
    """calls = number of times the script's sampling kernel is launched; per_call = samples that kernel takes per
    pixel and launch (the script's SAMPLE_PER_PIXEL when the loop is inside the kernel).  Sample k of a pixel
    always draws from the stream keyed (seed, x, y, k): the recording wrapper around raytrace() moves the stream
    on to the next sample index when a sample ends."""
    ti, _rt = install_standin()
    rng = Stream()
    _rt.rng = rng
    env_u8 = synthetic_env_u8(64, 32) if env else None
    env_shape = env_u8.shape[:2] if env else None
    # import: module-level frame loops (the bunny scripts' are unconditional) render nothing; 1-D fields (objects)
    # and the environment image are iterated normally (init_scene(), Image.process())
    _rt.pixels = lambda field: [] if (len(field.shape) == 2 and field.shape != env_shape) else None
    if env:
        _rt.imread = lambda path: env_u8.copy()
    sys.path.insert(0, os.path.join(REF, "examples", subdir))
    import importlib
    m = importlib.import_module(modname)                 # the reference's script
    rec = Recorder()
    W, H = m.image_resolution
    pixels = grid_pixels(W, H, *grid)
    _rt.pixels = lambda field: pixels if field.shape == (W, H) else None
    state = dict(sample=0)
    k_in = per_call(m) if per_call else 1

    def on_index(ix):
        if len(ix) == 2:
            rng.seek(ix[0], ix[1], state["sample"])
    _rt.on_index = on_index
    o_raytrace = m.raytrace
    # in-situ observation of the script's own raycast / ray_surface_interaction / calc_normal / nearest_object while its
    # render kernel runs (as run_v3 does): every event of every sample, so that a sample which differs in the oracle can be
    # traced to its first differing event (tests/test_oracle_refpin.py, the decision classifier)
    events = all(hasattr(m, n) for n in ("raycast", "ray_surface_interaction", "calc_normal", "nearest_object"))
    cur = dict(raycasts=0, steps=0, n_steps=0, normal=np.zeros(3, np.float32))
    if events:
        o_raycast, o_rsi, o_normal, o_nearest = m.raycast, m.ray_surface_interaction, m.calc_normal, m.nearest_object

        def obj_index(ob):          # which entry of the script's object table (matched by position: the table is small and distinct)
            want = tuple(vec_np(ob.transform.position).tolist())
            for i in range(m.objects.shape[0]):
                if tuple(vec_np(m.objects[i].transform.position).tolist()) == want:
                    return i
            return -1

        def nearest_object(p):
            cur["n_steps"] += 1
            return o_nearest(p)

        def raycast(ray):
            cur["n_steps"] = 0
            r = o_raycast(ray)
            cur["raycasts"] += 1
            cur["steps"] += cur["n_steps"]
            # HitRecord (cornell / bunny scripts) or the tuple (object, position, hit) of the scene_demo scripts
            hit, pos, ob = (r[2], r[1], r[0]) if isinstance(r, tuple) else (r.hit, r.position, r.object)
            rec.add("raycasts", px=rng.px, py=rng.py, sample=rng.sample, ro=vec_np(ray.origin), rd=vec_np(ray.direction),
                    hit=bool(hit), pos=vec_np(pos), obj=obj_index(ob), steps=cur["n_steps"])
            return r

        def calc_normal(obj, p):
            n = o_normal(obj, p)
            cur["normal"] = vec_np(n)
            return n

        def ray_surface_interaction(ray, *args):       # (ray, record) or (ray, object, position)
            n0 = rng.n
            cin, din = vec_np(ray.color), vec_np(ray.direction)
            out = o_rsi(ray, *args)
            position, ob = (args[1], args[0]) if len(args) == 2 else (args[0].position, args[0].object)
            rec.add("surface", px=rng.px, py=rng.py, sample=rng.sample, n0=n0, n1=rng.n, obj=obj_index(ob), pos=vec_np(position),
                    dir_in=din, color_in=cin, normal=cur["normal"], dir_out=vec_np(out.direction), color_out=vec_np(out.color),
                    origin_out=vec_np(out.origin))
            return out
        m.raycast, m.ray_surface_interaction, m.calc_normal, m.nearest_object = raycast, ray_surface_interaction, calc_normal, nearest_object

    def raytrace(ray):
        rin = (vec_np(ray.origin), vec_np(ray.direction))
        cur["raycasts"] = cur["steps"] = 0
        out = o_raytrace(ray)
        extra = dict(raycasts=cur["raycasts"], steps=cur["steps"]) if events else {}
        rec.add("samples", px=rng.px, py=rng.py, sample=rng.sample, ro=rin[0], rd=rin[1], color=vec_np(out.color), draws=rng.n, **extra)
        rng.seek(rng.px, rng.py, rng.sample + 1)
        return out
    m.raytrace = raytrace
    cam = ti.ui.Camera()
    cam.position(*cam_pos)
    t0 = time.time()
    for s in range(calls):
        state["sample"] = s * k_in
        step(m, cam, s, frame)
        print(f"{tag} launch {s}: {time.time() - t0:.0f} s", flush=True)
    if finish is not None:
        finish(m)
    px = np.array(pixels)
    arrays = rec.arrays()
    arrays["frame__pixels"] = px
    arrays["frame__image_buffer"] = m.image_buffer.to_numpy()[px[:, 0], px[:, 1]]
    arrays["frame__image_pixels"] = m.image_pixels.to_numpy()[px[:, 0], px[:, 1]]
    if env:
        arrays["env__u8"] = env_u8
    meta = dict(variant=tag, width=W, height=H, max_raytrace=getattr(m, "MAX_RAYTRACE", 3), max_raymarch=getattr(m, "MAX_RAYMARCH", 0),
                spp=calls * k_in, seed=SEED, frame=frame, camera_position=list(map(float, cam_pos)),
                generator=f"tools/ref_crosscheck.py {tag}")
    save(f"ref_{tag}.npz", arrays, meta)


def _fused(m, cam, s, frame):        # render(camera_position, camera_lookat, camera_up, moving)
    m.render(cam.curr_position, cam.curr_lookat, cam.curr_up, s == 0)


def _fused_frame(m, cam, s, frame):  # bunny_sdf.py / bunny_sdf_v2.py: render(..., moving, frame); buffer cleared when moving
    m.render(cam.curr_position, cam.curr_lookat, cam.curr_up, s == 0, frame)


def _shortest(m, cam, s, frame):
    from taichi.math import vec3
    m.render(vec3(0, 0, 3.5), vec3(0, 0, -1), vec3(0, 1, 0))      # cornell_box_shortest.py:135


def _split(m, cam, s, frame):        # scene_demo/main.py, tokyo_ibl.py: sample(pos, lookat, up); render() afterwards
    m.sample(cam.curr_position, cam.curr_lookat, cam.curr_up)


def _split_frame(m, cam, s, frame):  # bunny_sdf_glass.py: refresh(); sample(..., frame) x spp; render()
    if s == 0:
        m.refresh()
    m.sample(cam.curr_position, cam.curr_lookat, cam.curr_up, frame)


def _render(m):
    m.render()


LEGS = dict(
    v3=run_v3, src=run_src, bunny=run_bunny,
    src_adaptive=lambda: run_src(True, "ref_src_adaptive.npz", 40, (12, 8), 0.05),
    src_spp4_black=lambda: run_src(False, "ref_src_spp4_black.npz", 12, (16, 10), spp=4, black=True),
    v3b8=lambda: run_v3(8, "ref_v3b8.npz", (20, 20), 3),
    # the headline's geometry through the reference's own functions: 1920x1080 aspect, MAX_RAYTRACE 8, 64 x 36 pixels x 5 spp
    v3b8_wide=lambda: run_v3(8, "ref_v3b8_wide.npz", (64, 36), 5, resolution=(1920, 1080)),
    v2=lambda: run_script("v2", "cornell_box", "cornell_box_v2", (0, 0, 35.0), _fused, (24, 24), 4),
    v1=lambda: run_script("v1", "cornell_box", "cornell_box", (0, 0, 3.0), _fused, (16, 16), 2),
    shortest=lambda: run_script("shortest", "cornell_box", "cornell_box_shortest", (0, 0, 3.5), _shortest, (24, 24), 4),
    scene_demo=lambda: run_script("scene_demo", "scene_demo", "main", (0, -0.2, 4.0), _split, (24, 16), 3, finish=_render),
    tokyo=lambda: run_script("tokyo", "scene_demo", "tokyo_ibl", (0, -0.2, 4.0), _split, (24, 16), 3, env=True, finish=_render),
    bunny_glass=lambda: run_script("bunny_glass", "bunny", "bunny_sdf_glass", (0, 0, 4.0), _split_frame, (20, 12), 2, env=True, frame=17,
                                   finish=_render),
    bunny_sdf=lambda: run_script("bunny_sdf", "bunny", "bunny_sdf", (0, 0, 5.0), _fused_frame, (12, 8), 1, env=True, frame=30,
                                 per_call=lambda m: m.SAMPLE_PER_PIXEL),
    bunny_sdf_v2=lambda: run_script("bunny_sdf_v2", "bunny", "bunny_sdf_v2", (0, 0, 4.0), _fused_frame, (8, 6), 1, env=True, frame=30,
                                    per_call=lambda m: m.SAMPLE_PER_PIXEL),
)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s — this tool only runs in the build container" % REF)
    if which == "all":
        for leg in LEGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), leg], check=True)
    else:
        LEGS[which]()
