"""POD data model shared with the C ABI (include/rtpbr.h).

Mirrors the reference's ``@ti.dataclass`` types field for field
(reference: src/dataclass.py:5-46): Ray, Material, Transform, SDFObject, Camera.
They are ``ctypes.Structure`` so a list of SDFObject can be handed to
``rtpbr_set_scene`` without conversion.
"""
import ctypes as C
from enum import IntEnum

vec3_t = C.c_float * 3


def vec3(x, y=None, z=None):
    """taichi.math.vec3-like constructor: vec3(1) -> (1,1,1); vec3(a,b,c)."""
    if y is None and z is None:
        if isinstance(x, (tuple, list)) or hasattr(x, "__len__"):
            x, y, z = x
        else:
            y = z = x
    return (float(x), float(y), float(z))


class SHAPE(IntEnum):
    """reference: src/sdf.py:12-18; BUNNY = neural SDF of bunny_sdf_glass.py:149-203."""
    NONE = 0
    SPHERE = 1
    BOX = 2
    CYLINDER = 3
    CONE = 4
    PLANE = 5
    BUNNY = 6


class _Pod(C.Structure):
    def __repr__(self):
        parts = []
        for name, typ in self._fields_:
            v = getattr(self, name)
            if hasattr(v, "__len__"):
                v = tuple(v)
            parts.append(f"{name}={v}")
        return f"{type(self).__name__}({', '.join(parts)})"


class Ray(_Pod):
    """reference: src/dataclass.py:5-10 (40 B)."""
    _fields_ = [("origin", vec3_t), ("direction", vec3_t), ("color", vec3_t), ("depth", C.c_int32)]


class Material(_Pod):
    """reference: src/dataclass.py:13-20 (40 B). emission is multiplicative."""
    _fields_ = [("albedo", vec3_t), ("emission", vec3_t), ("roughness", C.c_float),
                ("metallic", C.c_float), ("transmission", C.c_float), ("ior", C.c_float)]

    def __init__(self, albedo=(1, 1, 1), emission=(1, 1, 1), roughness=1.0, metallic=0.0,
                 transmission=0.0, ior=1.0):
        super().__init__(vec3_t(*vec3(albedo)), vec3_t(*vec3(emission)), roughness, metallic,
                         transmission, ior)


class Transform(_Pod):
    """reference: src/dataclass.py:23-28 (72 B). rotation in Euler degrees; matrix is filled
    by the library at scene upload (src/scene.py:99-109)."""
    _fields_ = [("position", vec3_t), ("rotation", vec3_t), ("scale", vec3_t), ("matrix", C.c_float * 9)]

    def __init__(self, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        super().__init__(vec3_t(*vec3(position)), vec3_t(*vec3(rotation)), vec3_t(*vec3(scale)))


class SDFObject(_Pod):
    """reference: src/dataclass.py:31-35 (116 B)."""
    _fields_ = [("type", C.c_int32), ("transform", Transform), ("material", Material)]

    def __init__(self, type=SHAPE.BOX, transform=None, material=None):
        super().__init__(int(type), transform or Transform(), material or Material())


class Camera(_Pod):
    """reference: src/dataclass.py:38-46 (52 B); vfov in degrees."""
    _fields_ = [("lookfrom", vec3_t), ("lookat", vec3_t), ("vup", vec3_t), ("vfov", C.c_float),
                ("aspect", C.c_float), ("aperture", C.c_float), ("focus", C.c_float)]

    def __init__(self, lookfrom=(0, 0, 0), lookat=(0, 0, 1), vup=(0, 1, 0), vfov=35.0, aspect=1.0,
                 aperture=0.01, focus=4.0):
        super().__init__(vec3_t(*vec3(lookfrom)), vec3_t(*vec3(lookat)), vec3_t(*vec3(vup)), vfov,
                         aspect, aperture, focus)


class Counters(_Pod):
    _fields_ = [("samples", C.c_uint64), ("raycasts", C.c_uint64), ("march_steps", C.c_uint64),
                ("hits", C.c_uint64), ("sky_lookups", C.c_uint64), ("deposits", C.c_uint64)]


assert C.sizeof(Ray) == 40 and C.sizeof(Material) == 40 and C.sizeof(Transform) == 72
assert C.sizeof(SDFObject) == 116 and C.sizeof(Camera) == 52
