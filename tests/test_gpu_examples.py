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
