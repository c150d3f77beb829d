"""The runnable example scripts (examples/*.py: render -> imwrite on the HIP path), at small sizes (run with -m gpu)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracingpbr_amd.imageio import imread

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *map(str, args)], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_cornell_box_example(tmp_path):
    p = tmp_path / "c.png"
    s = run("cornell_box.py", "--variant", "v3", "--size", 256, 256, "--spp", 64, "--bounces", 4, "--out", p)
    img = imread(str(p))
    assert img.shape == (256, 256, 3) and "Msamples/s" in s
    # the light is on the ceiling (top of the image = high y in the field), red wall left, green wall right
    assert img[128, 221].min() > 200
    assert img[20, 128, 0] > img[20, 128, 1] + 30 and img[235, 128, 1] > img[235, 128, 0] + 30


def test_bunny_glass_example(tmp_path):
    run("bunny_glass.py", "--size", 160, 90, "--spp", 4, "--bounces", 8, "--frames", 0, 2, "--out-dir", tmp_path)
    a, b = imread(str(tmp_path / "frame_0000.png")), imread(str(tmp_path / "frame_0001.png"))
    assert a.shape == (160, 90, 3) and a.std() > 5 and not np.array_equal(a, b)


def test_tokyo_and_src_examples(tmp_path):
    run("tokyo_ibl.py", "--size", 192, 108, "--spp", 8, "--out", tmp_path / "t.png")
    assert imread(str(tmp_path / "t.png")).std() > 5
    s = run("src_viewer.py", "--size", 192, 108, "--frames", 120, "--out", tmp_path / "s.png")
    assert imread(str(tmp_path / "s.png")).shape == (192, 108, 3) and "refreshed while the camera moved" in s


def test_plain_c_host_renders_the_same_bits_as_the_python_host(tmp_path):
    """examples/c_host.c: BASELINE configs[0] (Cornell 256x256, 16 spp, 4 bounces) through the C ABI from a C program; the
    checksum of image_pixels it prints equals the Python host's for the same call sequence, and the PPM is written."""
    import subprocess
    from raytracingpbr_amd import Config, Renderer, _capi, cornell_box
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(_capi.HIP_LIB_PATH)
    exe, ppm = str(tmp_path / "c_host"), str(tmp_path / "out.ppm")
    subprocess.run(["gcc", "-std=c99", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_host.c"),
                    "-L" + lib_dir, "-lrtpbr_hip", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe], check=True)
    out = subprocess.run([exe, "256", "256", "16", ppm], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.split("\n") if l.startswith("backend")][-1]
    assert "hip-gfx950" in line and os.path.getsize(ppm) == len(b"P6\n256 256\n255\n") + 256 * 256 * 3
    r = Renderer(cornell_box("v3"), Config.cornell_v3(256, 256, 0, 4))
    r.render(refreshing=True, spp=16)
    h = 1469598103934665603
    for b in np.ascontiguousarray(r.image_pixels).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f"fnv1a {h:016x}" in line, (line, f"{h:016x}")
    c = r.counters()
    assert f"samples {c.samples} " in line and f"raycasts {c.raycasts} " in line
