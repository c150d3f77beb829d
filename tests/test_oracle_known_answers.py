"""K1: analytic known answers for every deterministic function of the path
(SURVEY.md §8(c) list).  The reference has no tests; these are derived from the maths of
the cited reference functions."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle_backend import OracleRenderer
from raytracingpbr_amd import SHAPE, Camera, Config, Material, SDFObject, Scene, Transform, cornell_box, src_scene

F3 = C.c_float * 3


def sdf(lib, shape, p, s, rho=0.0):
    f = lib.rto_test_sdf
    f.restype, f.argtypes = C.c_float, [C.c_int, F3, F3, C.c_float]
    return f(int(shape), F3(*p), F3(*s), rho)


def test_sdf_primitives(oracle_lib):
    # src/sdf.py:26-51
    assert sdf(oracle_lib, SHAPE.SPHERE, (3, 4, 0), (1, 1, 1)) == 4.0
    assert sdf(oracle_lib, SHAPE.BOX, (2, 0, 0), (1, 1, 1), 0.03) == pytest.approx(1 - 0.03, abs=1e-7)
    assert sdf(oracle_lib, SHAPE.BOX, (0, 0, 0), (1, 1, 1), 0.03) == pytest.approx(-1 - 0.03, abs=1e-7)
    assert sdf(oracle_lib, SHAPE.BOX, (2, 2, 0), (1, 1, 1), 0.0) == pytest.approx(math.sqrt(2), abs=1e-6)
    assert sdf(oracle_lib, SHAPE.CYLINDER, (2, 0, 0), (1, 1, 0)) == 1.0
    assert sdf(oracle_lib, SHAPE.CYLINDER, (0, 3, 0), (1, 1, 0)) == 2.0
    assert sdf(oracle_lib, SHAPE.CYLINDER, (0, 0, 0), (1, 2, 0)) == -1.0
    assert sdf(oracle_lib, SHAPE.PLANE, (5, 3, 1), (0, 1, 0)) == 2.0
    assert sdf(oracle_lib, SHAPE.NONE, (0, 0, 0), (1, 1, 1)) == 1e3
    # cone: max(dot(rh.xz, (|p.xz|, p.y)), -rh.y - p.y)
    assert sdf(oracle_lib, SHAPE.CONE, (1, 0, 0), (0.6, 1.0, 0.8)) == pytest.approx(0.6, abs=1e-7)


def test_rotate(oracle_lib):
    # src/util.py:36-42, SURVEY.md A.3
    f = oracle_lib.rto_rotate
    f.restype, f.argtypes = None, [F3, C.c_float * 9]
    m = (C.c_float * 9)()
    f(F3(0, 0, 0), m)
    assert list(m) == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    f(F3(math.radians(90), 0, 0), m)
    M = np.array(list(m)).reshape(3, 3)
    assert np.allclose(M @ np.array([0, 1, 0]), [0, 0, -1], atol=1e-6)
    ax, ay, az = 0.3, -1.1, 2.0
    f(F3(ax, ay, az), m)
    M = np.array(list(m)).reshape(3, 3)
    sx, cx, sy, cy, sz, cz = math.sin(ax), math.cos(ax), math.sin(ay), math.cos(ay), math.sin(az), math.cos(az)
    want = np.array([[cz * cy, cz * sy * sx + sz * cx, -cz * sy * cx + sz * sx],
                     [-sz * cy, -sz * sy * sx + cz * cx, sz * sy * cx + cz * sx],
                     [sy, -cy * sx, cy * cx]])
    assert np.allclose(M, want, atol=1e-6)
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-6)


def test_spherical_map_and_brightness(oracle_lib):
    f = oracle_lib.rto_test_spherical_map
    f.restype, f.argtypes = None, [F3, C.c_float * 2]
    uv = (C.c_float * 2)()
    f(F3(1, 0, 0), uv)
    assert list(uv) == [0.5, 0.5]
    f(F3(0, 1, 0), uv)
    assert uv[1] == pytest.approx(1.0, abs=1e-6)       # the out-of-range edge (G6)
    f(F3(0, -1, 0), uv)
    assert uv[1] == pytest.approx(0.0, abs=1e-6)
    f(F3(0, 0, 1), uv)
    assert uv[0] == pytest.approx(0.75, abs=1e-6)
    b = oracle_lib.rto_test_brightness
    b.restype, b.argtypes = C.c_float, [F3]
    assert b(F3(1, 1, 1)) == pytest.approx(1.0, abs=1e-6)
    assert b(F3(1, 0, 0)) == pytest.approx(0.299, abs=1e-7)


def test_aces_known_values(oracle_lib):
    # src/aces.py:5-30; values from SURVEY.md A.9
    f = oracle_lib.rto_test_aces
    f.restype, f.argtypes = None, [F3, C.c_int, F3]
    o = F3()
    f(F3(0, 0, 0), 0, o)
    assert all(abs(v - (-0.00038)) < 1e-5 for v in o)
    f(F3(1, 1, 1), 0, o)
    assert np.allclose(list(o), [0.619115, 0.619115, 0.619109], atol=2e-5)
    f(F3(.5, .5, .5), 0, o)
    assert np.allclose(list(o), 0.374308, atol=2e-5)
    f(F3(1, 0, 0), 0, o)
    assert np.allclose(list(o), [0.688028, -0.014495, 0.002639], atol=2e-5)


def test_tonemap_orders(oracle_lib):
    f = oracle_lib.rto_test_tonemap
    f.restype, f.argtypes = None, [C.POINTER(Config), C.c_float * 4, F3]
    o = F3()
    cfg = Config.src()
    f(C.byref(cfg), (C.c_float * 4)(1.0, 1.0, 1.0, 2.0), o)          # mean 0.5, gamma->ACES->clamp
    assert np.allclose(list(o), 0.510601, atol=3e-5)
    cfg = Config.cornell_v2()
    f(C.byref(cfg), (C.c_float * 4)(1.0, 1.0, 1.0, 2.0), o)          # ACES->gamma
    assert np.allclose(list(o), 0.639755, atol=3e-5)
    f(C.byref(cfg), (C.c_float * 4)(0.0, 0.0, 0.0, 1.0), o)          # negative ACES value under pow -> NaN (reference behaviour)
    assert all(math.isnan(v) for v in o)
    cfg = Config.tokyo_ibl(64, 36)
    f(C.byref(cfg), (C.c_float * 4)(0.0, 0.0, 0.0, 1.0), o)          # ... clamped to 0 in tokyo/scene_demo
    assert list(o) == [0.0, 0.0, 0.0]


def _renderer(scene, cfg, cam=None):
    return OracleRenderer(scene, cfg, cam)


def test_camera_centre_ray(oracle_lib):
    # get_ray with aperture 0 through the image centre -> normalize(lookat - lookfrom) (src/camera.py:11-36)
    cfg = Config.cornell_v3(64, 64)
    cam = Camera((1, 2, 3), (4, 6, 3), (0, 1, 0), 35, 1.0, 0.0, 4)
    r = _renderer(cornell_box("v3"), cfg, cam)
    g = oracle_lib.rto_test_get_ray
    g.restype, g.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_float * 6]
    out = (C.c_float * 6)()
    dirs = []
    for s in range(256):
        g(r._ctx, 31, 31, s, out)
        assert list(out)[:3] == [1, 2, 3]
        dirs.append(list(out)[3:])
    d = np.mean(dirs, axis=0)
    d /= np.linalg.norm(d)
    # pixel (31,31)+jitter averages to uv=(0.4922..), i.e. half a pixel off the centre
    want = np.array([3, 4, 0]) / 5.0
    assert np.allclose(d, want, atol=0.02)
    assert all(abs(np.linalg.norm(x) - 1) < 1e-6 for x in dirs)


def test_sphere_normal_and_nearest(oracle_lib):
    sc = Scene([SDFObject(SHAPE.SPHERE, Transform((0, 0, 0), 0, (2, 2, 2)), Material()),
                SDFObject(SHAPE.BOX, Transform((10, 0, 0), 0, (1, 1, 1)), Material())], False, Camera())
    cfg = Config.scene_demo(64, 36)
    r = _renderer(sc, cfg)
    n = oracle_lib.rto_test_normal
    n.restype, n.argtypes = C.c_int, [C.c_void_p, C.c_int, F3, F3]
    o = F3()
    n(r._ctx, 0, F3(2, 0, 0), o)
    assert np.allclose(list(o), [1, 0, 0], atol=1e-4)
    nn = oracle_lib.rto_test_nearest
    nn.restype, nn.argtypes = C.c_int, [C.c_void_p, F3, C.POINTER(C.c_float)]
    d = C.c_float()
    assert nn(r._ctx, F3(3, 0, 0), C.byref(d)) == 0 and d.value == pytest.approx(1.0, abs=1e-6)
    assert nn(r._ctx, F3(8, 0, 0), C.byref(d)) == 1 and d.value == pytest.approx(1 - 0.03, abs=1e-6)
    assert nn(r._ctx, F3(0, 0, 0), C.byref(d)) == 0 and d.value == pytest.approx(2.0, abs=1e-6)   # |sdf| inside


def test_src_normal_is_local_not_rotated(oracle_lib):
    # G3: src normals stay in the object's local frame (src/sdf.py:77-87)
    box = SDFObject(SHAPE.BOX, Transform((0, 0, 0), (0, 90, 0), (1, 2, 3)), Material())
    sc = Scene([box], False, Camera())
    n = oracle_lib.rto_test_normal
    n.restype, n.argtypes = C.c_int, [C.c_void_p, C.c_int, F3, F3]
    o = F3()
    r = _renderer(sc, Config.src(64, 36))
    n(r._ctx, 0, F3(3.5, 0, 0), o)        # world +x face of the rotated box
    local = np.array(list(o))
    r2 = _renderer(sc, Config.scene_demo(64, 36))
    n(r2._ctx, 0, F3(3.5, 0, 0), o)
    world = np.array(list(o))
    assert np.allclose(world, [1, 0, 0], atol=1e-3)
    assert abs(local[2]) > 0.99 and abs(local[0]) < 1e-3     # local z axis: NOT rotated back


def test_raycast_hits_sphere(oracle_lib):
    sc = Scene([SDFObject(SHAPE.SPHERE, Transform((0, 0, 0), 0, (1, 1, 1)), Material())], False, Camera())
    rc = oracle_lib.rto_test_raycast
    rc.restype, rc.argtypes = C.c_int, [C.c_void_p, F3, F3, F3, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    pos, hit, steps = F3(), C.c_int(), C.c_int()
    for cfg in (Config.scene_demo(256, 256), Config.cornell_v2(256, 256), Config.src(256, 256)):
        r = _renderer(sc, cfg)
        rc(r._ctx, F3(0, 0, 5), F3(0, 0, -1), pos, C.byref(hit), C.byref(steps))
        assert hit.value == 1 and steps.value < 40
        assert abs(pos[2] - 1.0) < 0.05 and abs(pos[0]) < 1e-6
        rc(r._ctx, F3(0, 3, 5), F3(0, 0, -1), pos, C.byref(hit), C.byref(steps))
        assert hit.value == 0


def test_surface_branches(oracle_lib):
    sf = oracle_lib.rto_test_surface
    sf.restype, sf.argtypes = C.c_int, [C.c_void_p, C.c_int, F3, F3, C.c_uint32, C.c_float * 9]
    out = (C.c_float * 9)()
    I = np.array([0.6, -0.8, 0.0])
    # perfect mirror: metallic 1, roughness 0 -> D = reflect(I, n), 3 draws (a, b, c1)
    mirror = SDFObject(SHAPE.SPHERE, Transform((0, -1, 0), 0, (1, 1, 1)), Material((0.5, 0.25, 1), (1, 1, 1), 0, 1, 0, 1.5))
    r = _renderer(Scene([mirror], False, Camera()), Config.scene_demo(64, 36))
    draws = sf(r._ctx, 0, F3(0, 0, 0), F3(*I), 0, out)
    assert draws == 3
    assert np.allclose(list(out)[:3], [0.6, 0.8, 0.0], atol=1e-3)
    assert np.allclose(list(out)[3:6], [0.5, 0.25, 1.0])            # color *= albedo on every branch (G9)
    # glass: transmission 1, roughness 0: refracted ray obeys Snell (when not Fresnel-reflected)
    glass = SDFObject(SHAPE.SPHERE, Transform((0, -1, 0), 0, (1, 1, 1)), Material((1, 1, 1), (1, 1, 1), 0, 0, 1, 1.5))
    r = _renderer(Scene([glass], False, Camera()), Config.scene_demo(64, 36))
    n_refr = 0
    for s in range(64):
        draws = sf(r._ctx, 0, F3(0, 0, 0), F3(*I), s, out)
        D = np.array(list(out)[:3])
        assert abs(np.linalg.norm(D) - 1) < 1e-5
        if D[1] < 0:                                                  # transmitted
            assert draws == 4
            eta = 1.000277 / 1.5
            assert abs(abs(D[0]) - eta * 0.6) < 2e-3                  # sin(theta_t) = eta * sin(theta_i)
            n_refr += 1
    assert n_refr > 32
    # diffuse: roughness 1 -> direction in the upper hemisphere of the face-forward normal
    # (draws == 4 selects the samples that did not take the Fresnel-reflect branch)
    diff = SDFObject(SHAPE.SPHERE, Transform((0, -1, 0), 0, (1, 1, 1)), Material((1, 1, 1), (1, 1, 1), 1, 0, 0, 1.5))
    r = _renderer(Scene([diff], False, Camera()), Config.scene_demo(64, 36))
    ups = []
    for s in range(400):
        if sf(r._ctx, 0, F3(0, 0, 0), F3(*I), s, out) == 4:
            ups.append(out[1])
    assert len(ups) > 250 and min(ups) > -1e-3 and 0.55 < np.mean(ups) < 0.78   # cosine weighted: E[cos] = 2/3


def test_scene_catalogue_matches_reference_tables():
    # object counts, sorting by type, x10 flag
    c = cornell_box("v3")
    assert len(c) == 8 and c.scale10 and all(o.type == SHAPE.BOX for o in c.objects)
    assert tuple(c.objects[7].material.emission) == (100, 100, 100)
    assert tuple(c.objects[5].transform.rotation) == (0, -253, 0)
    assert tuple(cornell_box("shortest").objects[5].transform.rotation) == (0, 112, 0)
    s = src_scene()
    assert [o.type for o in s.objects] == [1, 1, 1, 1, 2, 2, 3]     # spheres, boxes, cylinder (src/scene.py:33)
    assert s.objects[4].transform.position[2] == 5 and s.objects[6].transform.scale[0] == pytest.approx(0.3)
