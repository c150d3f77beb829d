"""K3 for the neural-SDF bunny: the silhouette in the reference's committed result image others/sdf_bunny_glass.jpg
(fixture tests/golden/bunny_glass_jpg_silhouette.npz, made by tools/make_jpg_fixture.py) against the primary-ray hit mask
of this implementation, same camera, 480 x 270.  CPU (oracle); the HIP path is bit-identical to it.

The image is one frame of the script's animation (rotation about the bunny's axis by pi f / 120), so the frame is fitted.
Finding recorded here: the outline agrees to IoU 0.96 at f = 62 only when the animation's vertical bob
(`p += vec3(0, 0, 0.1 sin t)`, bunny_sdf_glass.py:216) is cancelled — with the bob as committed the outline sits 0.10
world units (13 of 270 pixels) lower.  The published image was therefore rendered before that line existed (or with the
object lifted by the same amount); shape, camera and rotation convention agree.  The committed line itself is pinned by
tests/test_oracle_refpin.py::test_bunny_sdf_and_raycast against the reference's own code."""
import math
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle_backend import OracleRenderer
from raytracingpbr_amd import SHAPE, Config, Scene, SDFObject, Transform, bunny
from raytracingpbr_amd.config import SKY
from raytracingpbr_amd.ibl import load_bunny_weights

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 480, 270


def silhouette():
    d = np.load(os.path.join(HERE, "golden", "bunny_glass_jpg_silhouette.npz"))
    shape = tuple(int(v) for v in d["shape"])
    m = np.unpackbits(d["bits"])[:shape[0] * shape[1]].reshape(shape).astype(bool)
    return ndi.binary_erosion(m, iterations=2)          # the 21-px analysis window dilated the outline by ~2 px at this size


def hit_mask(frame, cancel_bob, make=OracleRenderer):
    """pixels whose camera rays hit the bunny: one bounce, black sky -> radiance > 0 exactly where the first raycast hits"""
    sc = bunny(aspect=W / H)
    cfg = Config.bunny_glass(W, H, 0, 1, frame).copy(sky_kind=SKY.BLACK)
    if cancel_bob:
        probe = make(sc, cfg)
        M = np.array(list(probe.get_scene()[0].transform.matrix), np.float64).reshape(3, 3)    # world -> local rotation
        probe.close() if hasattr(probe, "close") else None
        b = 0.1 * math.sin(math.pi * frame / 120.0)
        delta = M.T @ np.array([0.0, 0.0, b])            # local (0, 0, b) in world coordinates
        ob = sc.objects[0]
        pos = tuple(float(ob.transform.position[k] + delta[k]) for k in range(3))
        rot = tuple(float(v) for v in ob.transform.rotation)
        scl = tuple(float(v) for v in ob.transform.scale)
        sc = Scene([SDFObject(SHAPE.BUNNY, Transform(pos, rot, scl), ob.material)], False, sc.camera, "bunny_no_bob")
    r = make(sc, cfg)
    r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())
    r.sample(2)
    m = r.image_buffer[..., :3].sum(axis=2) > 0
    r.close() if hasattr(r, "close") else None
    return m


def iou(a, b):
    return (a & b).sum() / max((a | b).sum(), 1)


def test_bunny_outline_matches_the_published_image():
    ref = silhouette()
    assert 15000 < ref.sum() < 20000
    scores = {f: iou(hit_mask(f, True), ref) for f in range(50, 75, 2)}
    best = max(scores, key=scores.get)
    assert 58 <= best <= 66, scores
    assert scores[best] >= 0.95, scores
    # the fit is sharp in the rotation: +-12 frames (18 degrees) lose more than 0.06 of IoU
    assert scores[best] - max(scores[50], scores[74]) > 0.06, scores
    # with the bob of the committed script the outline is displaced by the bob's amplitude
    with_bob = hit_mask(best, False)
    assert iou(with_bob, ref) < scores[best] - 0.15
    shifts = {dy: iou(np.roll(with_bob, dy, axis=1), ref) for dy in range(8, 19)}
    dy = max(shifts, key=shifts.get)
    expect = 0.1 * math.sin(math.pi * best / 120.0) / (2 * 4 * math.tan(math.radians(15.0))) * H     # bob in pixels at distance 4
    assert abs(dy - expect) <= 1.5 and shifts[dy] >= 0.94, (dy, expect, shifts)
