"""raytracingpbr_amd — MI355X-native SDF path-tracing sample path.

Host-side mirror of the Scene / Camera / render() surface of HK-SHAO/RayTracingPBR
(reference: src/renderer.py, src/dataclass.py, src/scene.py, examples/*), driving
hand-written gfx950 HIP kernels through the C ABI in include/rtpbr.h.
"""
from .config import Config, FORM, MARCH, PRIMARY, SKY, TONEMAP
from .dataclass import SHAPE, Camera, Counters, Material, Ray, SDFObject, Transform, vec3
from .renderer import Renderer, display_image
from .scene import Scene, bunny, cornell_box, src_scene

__all__ = ["Config", "FORM", "MARCH", "PRIMARY", "SKY", "TONEMAP", "SHAPE", "Camera", "Counters",
           "Material", "Ray", "SDFObject", "Transform", "vec3", "Renderer", "display_image", "Scene",
           "bunny", "cornell_box", "src_scene"]
