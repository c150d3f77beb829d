"""Per-scene instances compiled at run time (rt_jit.hip; run with -m gpu): any scene of <= 8 analytic shapes gets the
unrolled kernels with its object count, shape types and rotation classes as compile-time constants; results must be
bit-identical to the ahead-of-time instances and to the oracle, and the code objects are cached on disk."""
import glob
import os
import time

import numpy as np
import pytest

from cases import case_by_name
from fuzz import random_box8_case, random_case, run
from oracle_backend import OracleRenderer
from raytracingpbr_amd import SHAPE, Config, Material, Renderer, Scene, SDFObject, Transform, cornell_box

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_catalog(tmp_path, monkeypatch):
    """These tests count the code objects the run-time compiler writes: hide the catalog compiled at build time
    (raytracingpbr_amd/data/jit), which would serve some of their scenes without compiling anything."""
    monkeypatch.setenv("RTPBR_JIT_CATALOG", str(tmp_path / "no-catalog"))



def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def axis_aligned_boxes(seed):
    """8 boxes whose rotations are multiples of 90 degrees about one axis (classes identity / X / Y / Z) — NOT the
    listed ahead-of-time signature"""
    rng = np.random.default_rng(seed)
    objs = []
    for i in range(8):
        axis = int(rng.integers(0, 4))
        rot = [0.0, 0.0, 0.0]
        if axis < 3:
            rot[axis] = float(rng.choice([90.0, -90.0, 180.0, 37.0]))
        pos = tuple(float(v) for v in rng.uniform(-6, 6, 3))
        size = tuple(float(v) for v in rng.uniform(0.5, 3.0, 3))
        alb = tuple(float(v) for v in rng.uniform(0.2, 0.9, 3))
        em = (30.0, 30.0, 30.0) if i == 7 else (1.0, 1.0, 1.0)
        objs.append(SDFObject(SHAPE.BOX, Transform(pos, tuple(rot), size), Material(alb, em, 1.0, 0.0, 0.0, 1.5)))
    sc = cornell_box("v3", aspect=96 / 54)
    return Scene(objs, False, sc.camera, "axis_boxes")


def test_axis_aligned_box_scene_gets_its_own_instance(tmp_path, monkeypatch):
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    sc = axis_aligned_boxes(3)
    cfg = Config.cornell_v3(96, 54, seed=5, max_raytrace=6)
    o = OracleRenderer(sc, cfg); o.sample(4)
    a = Renderer(sc, cfg); a.set_option("jit", 0); a.sample(4)              # ahead-of-time general instance
    assert a.counter("jit_active") == 0
    t0 = time.time()
    j = Renderer(sc, cfg); j.set_option("jit", 2); j.set_option("primary_split", 2); j.sample(4)
    t_first = time.time() - t0
    assert j.counter("jit_active") == 1
    assert np.array_equal(bits(j.image_buffer), bits(o.image_buffer))
    assert np.array_equal(bits(a.image_buffer), bits(o.image_buffer))
    assert len(glob.glob(str(tmp_path / "k1_n8_*.hsaco"))) <= 1     # (0 if an earlier test of this process built the key)
    # the default (-1) also compiles here: no listed ahead-of-time signature fits this scene
    d = Renderer(sc, cfg); d.sample(4)
    assert d.counter("jit_active") == 1 and np.array_equal(bits(d.image_buffer), bits(o.image_buffer))
    # ... but not for the Cornell Box, which the listed signature serves
    c = Renderer(cornell_box("v3"), Config.cornell_v3(64, 64, 0, 4)); c.sample(2)
    assert c.counter("jit_active") == 0
    print(f"first use (compile + load) {t_first:.1f} s")


@pytest.mark.parametrize("name", ["c1_cornell_v3_256_16spp_4b", "cornell_v3_8b_wide", "cornell_v2", "cornell_v1_128b", "cornell_shortest",
                                  "scene_demo_gradient", "tokyo_ibl_env", "bunny_glass", "bunny_chrome_frame30"])
def test_baked_instances_match_oracle_and_golden(name, tmp_path, monkeypatch):
    """option jit_bake: the scene's march table and the whole render configuration (every variant knob) as compile-time
    constants — every complete-path variant of the reference, fused and split primary kernels"""
    from test_oracle_golden import check_fingerprint
    from cases import fingerprint
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    case = case_by_name(name)
    o = OracleRenderer(case.scene, case.cfg); case.run(o)
    for split in (0, 2):
        g = Renderer(case.scene, case.cfg)
        g.set_option("jit", 2); g.set_option("jit_bake", 1); g.set_option("primary_split", split)
        case.run(g)
        assert g.counter("jit_active") == 1
        assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer)), split
        assert np.array_equal(bits(g.image_pixels), bits(o.image_pixels)), split
        cg, co = g.counters(), o.counters()
        assert (cg.samples, cg.raycasts, cg.march_steps, cg.hits, cg.sky_lookups, cg.deposits) == \
               (co.samples, co.raycasts, co.march_steps, co.hits, co.sky_lookups, co.deposits)
        check_fingerprint(fingerprint(g), case.name)
    n0 = len(glob.glob(str(tmp_path / "*.hsaco")))
    assert n0 <= 1
    # a different seed or frame number does not recompile (they stay launch arguments) ...
    g = Renderer(case.scene, case.cfg.copy(seed=case.cfg.seed + 1))
    g.set_option("jit", 2); g.set_option("jit_bake", 1); case.run(g)
    assert len(glob.glob(str(tmp_path / "*.hsaco"))) == n0
    o2 = OracleRenderer(case.scene, case.cfg.copy(seed=case.cfg.seed + 1)); case.run(o2)
    assert np.array_equal(bits(g.image_buffer), bits(o2.image_buffer))
    # ... any other knob does
    g = Renderer(case.scene, case.cfg.copy(max_raytrace=case.cfg.max_raytrace + 1))
    g.set_option("jit", 2); g.set_option("jit_bake", 1); case.run(g)
    assert len(glob.glob(str(tmp_path / "*.hsaco"))) == n0 + 1
    # jit_bake = 2 bakes the camera frame as well (fixed-camera offline renders): same bits, one more code object, and a
    # moved camera compiles another one
    n1 = len(glob.glob(str(tmp_path / "*.hsaco")))
    g = Renderer(case.scene, case.cfg)
    g.set_option("jit", 2); g.set_option("jit_bake", 2); case.run(g)
    assert g.counter("jit_active") == 1 and len(glob.glob(str(tmp_path / "*.hsaco"))) == n1 + 1
    assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer)) and np.array_equal(bits(g.image_pixels), bits(o.image_pixels))


@pytest.mark.parametrize("name", ["src_persistent", "src_persistent_4steps_blackbg", "src_adaptive_sampling"])
def test_persistent_ray_form_through_run_time_instances(name, tmp_path, monkeypatch):
    """the src/ pipeline (pathtrace() launches, ray state in ray_buffer): both of its schedulers, plain and baked"""
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    case = case_by_name(name)
    o = OracleRenderer(case.scene, case.cfg); case.run(o)
    # (the third setting makes the pool kernel's waves own more pixels than they hold: pass-major residencies of 4 steps)
    for sched, extra in ((0, {}), (1, {}), (1, {"grid_blocks": 2, "residency": 4})):
        for bake in (0, 1):
            g = Renderer(case.scene, case.cfg)
            g.set_option("jit", 2); g.set_option("jit_bake", bake); g.set_option("scheduler", sched)
            for k, v in extra.items():
                g.set_option(k, v)
            case.run(g)
            assert g.counter("jit_active") == 1
            assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer)), (sched, bake, extra)
            assert np.array_equal(bits(g.ray_buffer), bits(o.ray_buffer)), (sched, bake, extra)
            assert np.array_equal(bits(g.image_pixels), bits(o.image_pixels)), (sched, bake, extra)
            cg, co = g.counters(), o.counters()
            assert (cg.samples, cg.raycasts, cg.march_steps, cg.hits, cg.sky_lookups, cg.deposits) == \
                   (co.samples, co.raycasts, co.march_steps, co.hits, co.sky_lookups, co.deposits)


def test_tokyo_scene_instance_and_disk_cache(tmp_path, monkeypatch):
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    case = case_by_name("tokyo_ibl_env")
    o = OracleRenderer(case.scene, case.cfg); case.run(o)
    g = Renderer(case.scene, case.cfg); g.set_option("jit", 2); g.set_option("primary_split", 2); case.run(g)
    assert g.counter("jit_active") == 1
    assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer))
    assert np.array_equal(bits(g.image_pixels), bits(o.image_pixels))
    # (the un-baked key of this scene may or may not have been built earlier in this process — modules are shared per device —
    # so the disk cache is checked with the baked code object of a configuration no other test uses)
    cfg = case.cfg.copy(max_raytrace=37)
    o2 = OracleRenderer(case.scene, cfg); o2.set_env(case.env, case.env_exposure, case.env_gamma); o2.sample(3)
    g2 = Renderer(case.scene, cfg); g2.set_env(case.env, case.env_exposure, case.env_gamma)
    g2.set_option("jit", 2); g2.set_option("jit_bake", 1); g2.sample(3)
    files = [f for f in glob.glob(str(tmp_path / "k0_n7_*.hsaco")) if "_b0000000000000000_" not in f]     # the baked code object
    assert len(files) == 1 and "_f0_" in files[0] and g2.counter("jit_active") == 1
    assert np.array_equal(bits(g2.image_buffer), bits(o2.image_buffer))
    stamp = os.path.getmtime(files[0])
    g3 = Renderer(case.scene, cfg); g3.set_env(case.env, case.env_exposure, case.env_gamma)
    g3.set_option("jit", 2); g3.set_option("jit_bake", 1); g3.sample(3)
    assert g3.counter("jit_active") == 1 and os.path.getmtime(files[0]) == stamp
    assert np.array_equal(bits(g3.image_buffer), bits(o2.image_buffer))


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_scenes_through_run_time_instances(seed, tmp_path, monkeypatch):
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    sc, cfg, env, n = random_case(100 + seed) if seed % 2 else random_box8_case(100 + seed)
    analytic = all(ob.type != SHAPE.BUNNY for ob in sc.objects)
    eligible = len(sc.objects) <= 8 and analytic
    o = run(OracleRenderer(sc, cfg), env, n, False)
    g = Renderer(sc, cfg)
    if not eligible:
        # more than 8 objects, or the neural shape without baking: "jit" = 1 must fall back to the ahead-of-time kernels
        # silently and give the same bits; "jit" = 2 (strict) must refuse
        g.set_option("jit", 1)
        g = run(g, env, n, False)
        assert g.counter("jit_active") == 0
        assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer))
        s = Renderer(sc, cfg)
        s.set_option("jit", 2)
        with pytest.raises(Exception):
            run(s, env, n, False)
        return
    g.set_option("jit", 2)
    g.set_option("jit_bake", seed // 2 % 2)
    if seed % 4 < 2:
        g.set_option("primary_split", 2)
    g = run(g, env, n, False)
    assert g.counter("jit_active") == 1
    assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer))
    assert np.array_equal(bits(g.image_pixels), bits(o.image_pixels))
    cg, co = g.counters(), o.counters()
    assert (cg.samples, cg.raycasts, cg.march_steps, cg.hits, cg.sky_lookups) == (co.samples, co.raycasts, co.march_steps, co.hits, co.sky_lookups)


_RANK_SCRIPT = r"""
import sys, zlib
import numpy as np
sys.path[:0] = [{root!r}, {tests!r}]
from cases import case_by_name
from raytracingpbr_amd import Renderer
case = case_by_name("cornell_v3_8b_wide")
r = Renderer(case.scene, case.cfg)
r.set_option("jit", 2)
r.set_option("jit_bake", 1)
r.set_tiles(8, 8, 0, 4)
r.sample(5)
assert r.counter("jit_active") == 1
print("CRC", zlib.crc32(np.ascontiguousarray(r.image_buffer).tobytes()))
"""


def test_ranks_of_one_job_build_the_same_key_concurrently(tmp_path):
    """bench.py under torchrun: every rank asks for the same baked key at the same moment, with an empty cache.  Each
    process writes only pid-suffixed files and renames the code object into place, so all of them must succeed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTPBR_JIT_CACHE=str(tmp_path))
    code = _RANK_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(4)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    crcs = {so.strip().splitlines()[-1] for so, _ in outs}
    assert len(crcs) == 1
    left = sorted(os.listdir(tmp_path))
    assert len([f for f in left if f.endswith(".hsaco")]) == 1 and not [f for f in left if ".tmp." in f], left


def test_catalog_instance_runs_without_a_compiler(tmp_path, monkeypatch):
    """A BASELINE scene on a target without hipcc (round 6): HIPCC points nowhere, the code-object cache is empty, option jit = 2
    (strict: no fall-back to the ahead-of-time kernels) — the baked instance comes from the catalog compiled at build time
    (raytracingpbr_amd/data/jit, rt_jit.hip) and its frame is the oracle's bit for bit; a scene that is not in the catalog is an
    error in that situation, not a silent fall-back."""
    from raytracingpbr_amd import workloads
    from raytracingpbr_amd._capi import RtpbrError
    monkeypatch.delenv("RTPBR_JIT_CATALOG", raising=False)
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))
    wl = workloads.get("c1")
    g = Renderer(wl.scene, wl.cfg)
    g.set_option("jit", 2); g.set_option("jit_bake", 2)
    g.sample(wl.spp)
    assert g.counter("jit_active") == 1 and not glob.glob(str(tmp_path / "*.hsaco"))
    o = OracleRenderer(wl.scene, wl.cfg); o.sample(wl.spp)
    assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer))
    g.close()
    wl = workloads.get("c1", 200, 120)
    g = Renderer(wl.scene, wl.cfg)
    g.set_option("jit", 2); g.set_option("jit_bake", 2)
    with pytest.raises(RtpbrError):
        g.sample(1)
    g.close()
