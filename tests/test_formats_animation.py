"""§8(f) rows 2-3: on-disk formats either side of the path and the offline animation driver."""
import os

import numpy as np
import pytest

from cases import case_by_name
from oracle_backend import OracleRenderer
from raytracingpbr_amd.animation import render_animation
from raytracingpbr_amd.imageio import hdr_to_ldr_stb, imread, imwrite, read_hdr, write_hdr


def test_rgbe_known_values(tmp_path):
    # hand-built flat scanline: RGBE (128,64,32,129) = (128,64,32) * 2^(129-136) = (1.0, 0.5, 0.25)
    p = tmp_path / "k.hdr"
    body = bytes([128, 64, 32, 129, 0, 0, 0, 0, 255, 255, 255, 128, 1, 2, 3, 136])
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 4\n" + body)
    img = read_hdr(str(p))
    assert img.shape == (1, 4, 3)
    assert np.allclose(img[0, 0], [1.0, 0.5, 0.25])
    assert np.all(img[0, 1] == 0)
    assert np.allclose(img[0, 2], 255 / 256)
    assert np.allclose(img[0, 3], [1, 2, 3])


@pytest.mark.parametrize("rle", [False, True])
def test_hdr_roundtrip_and_stb_ldr(tmp_path, rle):
    rng = np.random.default_rng(0)
    img = (rng.random((20, 64, 3)).astype(np.float32) ** 3) * 8.0
    img[3:9, 10:40] = 2.5                                   # long runs exercise the RLE path
    p = str(tmp_path / "e.hdr")
    write_hdr(p, img, rle=rle)
    back = read_hdr(p)
    assert back.shape == img.shape
    assert np.all(np.abs(back - img) <= img.max(axis=2, keepdims=True) / 128.0 + 1e-6)   # 8-bit mantissa
    ldr = hdr_to_ldr_stb(back)
    assert ldr.dtype == np.uint8 and ldr[4, 20, 0] == 255                               # 2.5 clips
    assert hdr_to_ldr_stb(np.array([[[0.5, 1.0, 0.0]]], np.float32)).tolist() == [[[186, 255, 0]]]   # 255*0.5^(1/2.2)+.5
    f = imread(p)
    assert f.shape == (64, 20, 3) and f.dtype == np.uint8
    assert np.array_equal(f[:, ::-1].swapaxes(0, 1), ldr)    # (W,H,C), origin bottom-left


def test_png_write_read_orientation(tmp_path):
    W, H = 12, 7
    px = np.zeros((W, H, 3), np.float32)
    px[0, 0] = (1, 0, 0)          # bottom-left red
    px[W - 1, H - 1] = (0, 1, 0)  # top-right green
    px[3, 2] = (0.5, 0.5, 0.5)
    p = str(tmp_path / "o.png")
    imwrite(px, p)
    from PIL import Image
    im = np.asarray(Image.open(p))
    assert im.shape == (H, W, 3)
    assert im[H - 1, 0].tolist() == [255, 0, 0] and im[0, W - 1].tolist() == [0, 255, 0]
    assert im[H - 1 - 2, 3].tolist() == [128, 128, 128]                 # 0.5*255+0.5 -> 128
    back = imread(p)
    assert back.shape == (W, H, 3) and back[0, 0].tolist() == [255, 0, 0]


def test_env_file_drives_the_renderer(tmp_path):
    """imread(.hdr) -> set_env(uint8, exposure, gamma): the reference's src/ibl.py:14-23 pipeline."""
    case = case_by_name("tokyo_ibl_env")
    hdr = (np.random.default_rng(1).random((48, 96, 3)).astype(np.float32) * 2.0)
    p = str(tmp_path / "env.hdr")
    write_hdr(p, hdr)
    env = imread(p)
    assert env.shape == (96, 48, 3)
    r = OracleRenderer(case.scene, case.cfg)
    r.set_env(env, 1.8, 2.2)
    r.sample(2)
    assert r.counters().sky_lookups > 0 and np.isfinite(r.image_buffer).all()


def test_animation_driver_matches_manual_loop(tmp_path):
    case = case_by_name("bunny_chrome_frame30")
    r = OracleRenderer(case.scene, case.cfg)
    case.setup(r)
    paths = render_animation(r, [0, 30, 60], spp=1, out_dir=str(tmp_path))
    assert [os.path.basename(p) for p in paths] == ["frame_0000.png", "frame_0030.png", "frame_0060.png"]
    m = OracleRenderer(case.scene, case.cfg)
    case.setup(m)
    frames = render_animation(m, [0, 30, 60], spp=1)
    # frame 30 rendered through the driver == the golden case's config (frame=30) rendered directly,
    # except that the driver continues the sample counter across frames: compare against a manual loop
    k = OracleRenderer(case.scene, case.cfg)
    case.setup(k)
    for i, f in enumerate([0, 30, 60]):
        k.set_config(case.cfg.copy(frame=f)); k.refresh(); k.sample(1); k.post_process()
        assert np.array_equal(np.nan_to_num(k.image_pixels), np.nan_to_num(frames[i]))
    assert not np.array_equal(frames[0], frames[1])            # the bunny actually moves
