"""On-disk formats and the animation driver ON THE HIP PATH (run with -m gpu).

The CPU tests (test_formats_animation.py) check the codecs against known answers; here the decoded environment map
and the per-frame uniform drive the HIP kernels and the result is compared with the oracle bit for bit."""
import os

import numpy as np
import pytest

from cases import case_by_name
from oracle_backend import OracleRenderer
from raytracingpbr_amd import Renderer
from raytracingpbr_amd.animation import render_animation
from raytracingpbr_amd.imageio import imread, imwrite, write_hdr

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_hdr_file_env_drives_the_hip_path(tmp_path):
    """write a Radiance .hdr -> imread (RGBE/RLE decode + stb-style 8-bit conversion, src/ibl.py:15) -> set_env
    (uint8, exposure 1.8, gamma 2.2: Image.process) -> render on HIP and on the oracle."""
    case = case_by_name("tokyo_ibl_env")
    rng = np.random.default_rng(7)
    hdr = (rng.random((64, 128, 3)).astype(np.float32) ** 3) * 6.0          # HDR range, (H, W, 3) top-down
    hdr[10:14, 40:44] = 500.0                                                # a "sun" far above the 8-bit range
    p = str(tmp_path / "env.hdr")
    write_hdr(p, hdr)
    env = imread(p)
    assert env.dtype == np.uint8 and env.shape == (128, 64, 3) and env.max() == 255
    g, o = Renderer(case.scene, case.cfg), OracleRenderer(case.scene, case.cfg)
    for r in (g, o):
        r.set_env(env, 1.8, 2.2)
        r.sample(6)
        r.post_process()
    assert g.counters().sky_lookups == o.counters().sky_lookups > 0
    assert np.array_equal(bits(g.image_buffer), bits(o.image_buffer))
    assert np.array_equal(bits(g.image_pixels), bits(o.image_pixels))
    # and the PNG written from the HIP image round-trips through imread
    out = str(tmp_path / "frame.png")
    imwrite(g.image_pixels, out)
    back = imread(out)
    want = np.clip(np.nan_to_num(g.image_pixels) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    assert back.shape == want.shape and np.array_equal(back, want)


def test_animation_driver_on_the_hip_renderer(tmp_path):
    """bunny_sdf_glass.py:434-451: per frame u_frame -> refresh -> spp x sample -> tone map -> imwrite"""
    case = case_by_name("bunny_chrome_frame30")
    g, o = Renderer(case.scene, case.cfg), OracleRenderer(case.scene, case.cfg)
    case.setup(g)
    case.setup(o)
    frames = [0, 30, 60, 119]
    w = Renderer(case.scene, case.cfg)              # (the driver continues the sample counter across frames and calls)
    case.setup(w)
    paths = render_animation(w, frames, spp=2, out_dir=str(tmp_path))
    assert [os.path.basename(q) for q in paths] == ["frame_%04d.png" % f for f in frames]
    got = render_animation(g, frames, spp=2)
    want = render_animation(o, frames, spp=2)
    for a, b in zip(got, want):
        assert np.array_equal(bits(a), bits(b))
    assert not np.array_equal(got[0], got[1])                     # the bunny moves
    first = imread(paths[0])
    assert first.shape == (case.cfg.width, case.cfg.height, 3)
