"""Scene catalogue: the reference's object tables as data.

reference: src/scene.py:11-33 (7-object demo scene, sorted by type),
examples/cornell_box/cornell_box_v3/scene.py:6-27 (Cornell Box, 8 boxes),
examples/scene_demo/tokyo_ibl.py:100-123, examples/bunny/bunny_sdf_glass.py:221-225.
A Scene is a list of SDFObject plus the per-script camera pose and the "x10" flag of
cornell_box_v2/v3 (position and scale multiplied by 10 inside signed_distance,
cornell_box_v3/sdf.py:16-18).
"""
from dataclasses import dataclass, field
from typing import List

from .dataclass import SHAPE, Camera, Material, SDFObject, Transform, vec3


def _v(a, s):
    return tuple(x * s for x in a)


@dataclass
class Scene:
    objects: List[SDFObject]
    scale10: bool = False
    camera: Camera = field(default_factory=Camera)
    name: str = "scene"

    def __len__(self):
        return len(self.objects)


def _cornell_objects(tall_yaw=-253.0):
    grey, W = _v((1, 1, 1), 0.4), vec3(1)
    wall = lambda pos, rot, alb: SDFObject(SHAPE.BOX, Transform(pos, rot, (1, 1, 0.2)), Material(alb, W, 1, 0, 0, 1.530))
    return [
        wall((0, 0, -1), (0, 0, 0), grey),
        wall((0, 1, 0), (90, 0, 0), grey),
        wall((0, -1, 0), (90, 0, 0), grey),
        wall((-1, 0, 0), (0, 90, 0), _v((1, 0, 0), 0.5)),
        wall((1, 0, 0), (0, 90, 0), _v((0, 1, 0), 0.5)),
        SDFObject(SHAPE.BOX, Transform((-0.275, -0.3, -0.2), (0, tall_yaw, 0), (0.25, 0.5, 0.25)), Material(grey, W, 1, 0, 0, 1.530)),
        SDFObject(SHAPE.BOX, Transform((0.275, -0.55, 0.2), (0, -197, 0), (0.25, 0.25, 0.25)), Material(grey, W, 1, 0, 0, 1.530)),
        SDFObject(SHAPE.BOX, Transform((0, 0.809, 0), (90, 0, 0), (0.2, 0.2, 0.01)), Material((1, 1, 1), vec3(100), 1, 0, 0, 1)),
    ]


def cornell_box(variant="v3", aspect=1.0):
    """Cornell Box.  variant in {"v1","v2","v3","shortest"}."""
    if variant in ("v2", "v3"):
        # cornell_box_v3/main.py:13 camera.position(0,0,35); config.py:21-24; GGUI default lookat (0,0,1)
        cam = Camera((0, 0, 35), (0, 0, 1), (0, 1, 0), 35, aspect, 0.01, 4)
        return Scene(_cornell_objects(), True, cam, f"cornell_{variant}")
    if variant == "v1":
        cam = Camera((0, 0, 3), (0, 0, 1), (0, 1, 0), 43.6, aspect, 0.01, 4)   # cornell_box.py:32-34,384
        return Scene(_cornell_objects(), False, cam, "cornell_v1")
    if variant == "shortest":
        cam = Camera((0, 0, 3.5), (0, 0, -1), (0, 1, 0), 35, 1.0, 0.0, 1.0)    # shortest:135,111
        return Scene(_cornell_objects(112.0), False, cam, "cornell_shortest")
    raise ValueError(variant)


def src_scene(aspect=768 / 432, tokyo=False):
    """7-object demo scene; sorted by type like src/scene.py:33."""
    W = vec3(1)
    if not tokyo:
        objs = [
            SDFObject(SHAPE.SPHERE, Transform((0, -100.501, 0), 0, vec3(100)), Material(_v((1, 1, 1), 0.6), W, 1.0, 1.0, 0, 1.100)),
            SDFObject(SHAPE.SPHERE, Transform((0, 0, 0), 0, vec3(0.5)), Material(_v((1, 1, 1), 0.9), (1, 10, 1), 0, 1, 0, 1.000)),
            SDFObject(SHAPE.SPHERE, Transform((1, -0.2, 0), 0, vec3(0.3)), Material(_v((0.2, 0.2, 1), 0.9), W, 0.2, 1, 0, 1.100)),
            SDFObject(SHAPE.SPHERE, Transform((0.0, -0.2, 2), 0, vec3(0.3)), Material(_v((1, 1, 1), 0.9), W, 0, 0, 1, 1.500)),
            SDFObject(SHAPE.CYLINDER, Transform((-1.0, -0.2, 0), 0, vec3(0.3)), Material(_v((1.0, 0.2, 0.2), 0.9), W, 0, 0, 0, 1.460)),
            SDFObject(SHAPE.BOX, Transform((0, 0, 5), 0, (2, 1, 0.2)), Material(_v((1, 1, 0.2), 0.9), W, 0, 1, 0, 0.470)),
            SDFObject(SHAPE.BOX, Transform((0, 0, -2), 0, (2, 1, 0.2)), Material(_v((1, 1, 1), 0.9), W, 0, 1, 0, 2.950)),
        ]
        cam = Camera((0, -0.2, 4), (0, 0, 1), (0, 1, 0), 35, aspect, 0.01, 4)   # src/main.py:17, camera.py:127-129
    else:
        objs = [
            SDFObject(SHAPE.SPHERE, Transform((0, -100.501, 0), 0, vec3(100)), Material(_v((1, 1, 1), 0.6), W, 1, 1, 0, 1.635)),
            SDFObject(SHAPE.SPHERE, Transform((0, 0, 0), 0, vec3(0.5)), Material((1, 1, 1), _v((0.1, 1, 0.1), 10), 1, 0, 0, 1)),
            SDFObject(SHAPE.SPHERE, Transform((1, -0.2, 0), 0, vec3(0.3)), Material((0.2, 0.2, 1), W, 0.2, 1, 0, 1.100)),
            SDFObject(SHAPE.SPHERE, Transform((0.0, -0.2, 2), 0, vec3(0.3)), Material(_v((1, 1, 1), 0.9), W, 0, 0, 1, 1.5)),
            SDFObject(SHAPE.CYLINDER, Transform((-1.0, -0.2, 0), 0, vec3(0.3)), Material((1.0, 0.2, 0.2), W, 0, 0, 0, 1.460)),
            SDFObject(SHAPE.BOX, Transform((0, 0, 5), 0, (2, 1, 0.2)), Material(_v((1, 1, 0.2), 0.9), W, 0, 1, 0, 0.470)),
            SDFObject(SHAPE.BOX, Transform((0, 0, -2), 0, (2, 1, 0.2)), Material(_v((1, 1, 1), 0.9), W, 0, 1, 0, 2.950)),
        ]
        cam = Camera((0, -0.2, 4), (0, 0, 1), (0, 1, 0), 30, aspect, 0.01, 4)   # tokyo_ibl.py:33-35,445
    objs = sorted(objs, key=lambda o: o.type)        # stable, like Python's sorted in the reference
    return Scene(objs, False, cam, "tokyo" if tokyo else "src")


def bunny(aspect=1920 / 1080, chrome=False, v2=False):
    """Neural-SDF bunny; bunny_sdf_glass.py:221-225 / bunny_sdf.py:219-221 /
    bunny_sdf_v2.py (chrome, camera at z = 4: bunny_sdf_v2.py:437)."""
    if chrome:
        mat = Material(_v((1, 1, 1), 0.9), vec3(1), 0, 1, 0, 2.950)
        cam = Camera((0, 0, 4 if v2 else 5), (0, 0, 1), (0, 1, 0), 30, aspect, 0.01, 4)
    else:
        mat = Material(_v((1, 1, 1), 0.9), vec3(1), 0, 0, 1, 1.500)
        cam = Camera((0, 0, 4), (0, 0, 1), (0, 1, 0), 30, aspect, 0.03, 4)
    return Scene([SDFObject(SHAPE.BUNNY, Transform((0, 0, 0), (-90, 0, 0), (1, 1, 1)), mat)], False, cam, "bunny")
