"""Pins the oracle's exactly-specified math layer (oracle/rt_oracle_math.h) against
numpy / libm: these functions replace taichi.math.{sin,cos,exp,atan2,asin} and ti.random()
(call sites: src/util.py:13-28,45-62; cornell_box_v3/pathtracer.py:84-87)."""
import ctypes as C

import numpy as np
import pytest


def _f(lib, name, res, args):
    f = getattr(lib, name)
    f.restype, f.argtypes = res, args
    return f


def test_sincos_matches_numpy(oracle_lib):
    f = _f(oracle_lib, "rto_test_sincos", None, [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 4000), rng.uniform(-40, 40, 2000),
                         [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi]]).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in xs:
        f(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
        assert abs(s.value * s.value + c.value * c.value - 1.0) < 1e-6
    assert worst < 2.5e-7, worst


def test_sincos_known_values(oracle_lib):
    f = _f(oracle_lib, "rto_test_sincos", None, [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    s, c = C.c_float(), C.c_float()
    f(0.0, C.byref(s), C.byref(c))
    assert s.value == 0.0 and c.value == 1.0


def test_sin_pi_for_mlp_activations(oracle_lib):
    f = _f(oracle_lib, "rto_test_sin_pi", C.c_float, [C.c_float])
    xs = np.concatenate([np.linspace(-40, 40, 8001), [0.0, np.pi / 2, np.pi, -np.pi / 2]]).astype(np.float32)
    worst = max(abs(f(float(x)) - np.sin(np.float64(x))) for x in xs)
    assert worst < 4e-7, worst
    assert f(0.0) == 0.0


def test_log_pow(oracle_lib):
    fl = _f(oracle_lib, "rto_test_log", C.c_float, [C.c_float])
    fp = _f(oracle_lib, "rto_test_pow", C.c_float, [C.c_float, C.c_float])
    for x in np.concatenate([np.logspace(-30, 30, 1201), [1.0, 0.5, 2.0, 1e-40]]).astype(np.float32):
        assert abs(fl(float(x)) - np.log(np.float64(x))) <= 2e-7 * max(1.0, abs(np.log(np.float64(x))))
    assert fl(1.0) == 0.0
    for x in np.linspace(1e-4, 20, 2001).astype(np.float32):
        for y in (1 / 2.2, 2.2):
            ref = np.float64(x) ** y
            assert abs(fp(float(x), y) - ref) <= 2e-6 * ref
    assert fp(0.0, 0.45) == 0.0 and np.isnan(fp(-1.0, 0.45))
    assert abs(fp(1.4, 2.2) - 2.0964) < 1e-3                 # max env value (1.4)^2.2 = 2.10


def test_exp(oracle_lib):
    f = _f(oracle_lib, "rto_test_exp", C.c_float, [C.c_float])
    assert f(0.0) == 1.0
    xs = np.linspace(-10, 10, 2001).astype(np.float32)
    for x in xs:
        ref = np.exp(np.float64(x))
        assert abs(f(float(x)) - ref) <= 3e-7 * ref
    # the russian-roulette table of the examples: p_i = 1 - exp(-i/128)
    want = [0.0, 0.0078, 0.0155, 0.0232, 0.0308, 0.0383, 0.0458, 0.0532]
    for i, w in enumerate(want):
        p = 1.0 - 1.0 / f(i / 128.0)
        assert abs(p - w) < 6e-5, (i, p)


def test_atan2_asin(oracle_lib):
    fa = _f(oracle_lib, "rto_test_atan2", C.c_float, [C.c_float, C.c_float])
    fs = _f(oracle_lib, "rto_test_asin", C.c_float, [C.c_float])
    rng = np.random.default_rng(1)
    for y, x in rng.normal(size=(3000, 2)).astype(np.float32):
        assert abs(fa(float(y), float(x)) - np.arctan2(np.float64(y), np.float64(x))) < 4e-7
    for x in np.linspace(-1, 1, 2001).astype(np.float32):
        assert abs(fs(float(x)) - np.arcsin(np.float64(x))) < 4e-7
    assert fa(0.0, 1.0) == 0.0
    assert abs(fa(1.0, 0.0) - np.pi / 2) < 1e-7
    assert abs(fa(0.0, -1.0) - np.pi) < 1e-6
    # out-of-range argument is clamped (the reference would produce NaN): documented deviation
    assert abs(fs(1.0000001) - np.pi / 2) < 1e-6


def test_rng_determinism_and_uniformity(oracle_lib):
    f = _f(oracle_lib, "rto_test_rand", C.c_float, [C.c_uint32] * 5)
    a = np.array([f(0, x, y, s, n) for x in range(8) for y in range(8) for s in range(8) for n in range(16)])
    b = np.array([f(0, x, y, s, n) for x in range(8) for y in range(8) for s in range(8) for n in range(16)])
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005
    # 24-bit resolution like ti.random(f32)
    assert np.all(a * 2 ** 24 == np.round(a * 2 ** 24))
    # different seeds / pixels / samples give different streams
    assert f(0, 1, 2, 3, 0) != f(1, 1, 2, 3, 0)
    assert f(0, 1, 2, 3, 0) != f(0, 2, 1, 3, 0)
    assert f(0, 1, 2, 3, 0) != f(0, 1, 2, 4, 0)
    # neighbouring draws are uncorrelated
    c = a.reshape(-1, 16)
    assert abs(np.corrcoef(c[:, 0], c[:, 1])[0, 1]) < 0.1
