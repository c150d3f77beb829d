"""ctypes binding of the C ABI declared in include/rtpbr.h.

The reference crosses from Python into device code through Taichi kernel launches
(src/renderer.py:25-32); here the same calls go through a C-ABI shared library built from
hand-written HIP (raytracingpbr_amd/csrc/).  There is NO fallback: if the HIP library is
missing or fails to load this module raises, it never substitutes a CPU path.

``CApi(path, prefix)`` is generic over the symbol prefix so that tests can drive any
library exporting the same entry points.
"""
import ctypes as C
import os

from .config import Config
from .dataclass import Camera, Counters, SDFObject

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTPBR_HIP_LIB overrides the path (A/B of differently built HIP libraries); it must still be a HIP build
HIP_LIB_PATH = os.environ.get("RTPBR_HIP_LIB") or os.path.join(_HERE, "csrc", "librtpbr_hip.so")

# every symbol include/rtpbr.h declares (checked by tests/test_host_logic.py::test_hip_library_exports_every_declared_symbol)
ENTRY_POINTS = [
    "create", "destroy", "last_error", "backend", "set_config", "set_scene", "get_scene",
    "set_camera", "set_env", "set_tiles", "refresh", "sample", "post_process", "sync",
    "read_buffer", "write_buffer", "host_alloc", "host_free", "buffer_device_ptr", "read_buffer_async", "read_wait",
    "packed_bytes", "pack_tiles", "unpack_tiles",
    "get_counters", "get_counter", "last_sample_ms", "last_primary_ms", "get_stream", "set_option", "set_shape_data",
    "rccl_unique_id", "rccl_init", "rccl_init_all", "gather_tiles", "gather_tiles_all", "rccl_info", "jit_prebuild",
]


class RtpbrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtpbr error {code}: {msg}")
        self.code = code


class CApi:
    def __init__(self, path, prefix="rtpbr_", optional=("test_math", "jit_prebuild", "rccl_unique_id", "rccl_init", "rccl_init_all", "gather_tiles", "gather_tiles_all", "rccl_info")):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found — build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        p = C.c_void_p
        sig = {
            "create": (C.c_int, [C.c_int, C.POINTER(p)]),
            "destroy": (C.c_int, [p]),
            "last_error": (C.c_char_p, []),
            "backend": (C.c_char_p, []),
            "set_config": (C.c_int, [p, C.POINTER(Config)]),
            "set_scene": (C.c_int, [p, C.POINTER(SDFObject), C.c_int, C.c_int]),
            "get_scene": (C.c_int, [p, C.POINTER(SDFObject), C.c_int]),
            "set_camera": (C.c_int, [p, C.POINTER(Camera)]),
            "set_env": (C.c_int, [p, p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
            "set_tiles": (C.c_int, [p, C.c_int, C.c_int, C.c_int, C.c_int]),
            "refresh": (C.c_int, [p]),
            "sample": (C.c_int, [p, C.c_int]),
            "post_process": (C.c_int, [p]),
            "sync": (C.c_int, [p]),
            "read_buffer": (C.c_int, [p, C.c_int, p, C.c_size_t]),
            "write_buffer": (C.c_int, [p, C.c_int, p, C.c_size_t]),
            "host_alloc": (C.c_int, [p, C.c_size_t, C.POINTER(p)]),
            "host_free": (C.c_int, [p, p]),
            "buffer_device_ptr": (C.c_int, [p, C.c_int, C.POINTER(p), C.POINTER(C.c_size_t)]),
            "read_buffer_async": (C.c_int, [p, C.c_int, p, C.c_size_t, C.POINTER(C.c_int)]),
            "read_wait": (C.c_int, [p, C.c_int]),
            "packed_bytes": (C.c_int, [p, C.POINTER(C.c_size_t)]),
            "pack_tiles": (C.c_int, [p, p]),
            "unpack_tiles": (C.c_int, [p, p, C.c_int]),
            "get_counters": (C.c_int, [p, C.POINTER(Counters)]),
            "get_counter": (C.c_int, [p, C.c_char_p, C.POINTER(C.c_uint64)]),
            "last_sample_ms": (C.c_int, [p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
            "last_primary_ms": (C.c_int, [p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
            "get_stream": (C.c_int, [p, C.POINTER(p)]),
            "set_option": (C.c_int, [p, C.c_char_p, C.c_longlong]),
            "set_shape_data": (C.c_int, [p, C.c_int, p, C.c_int]),
            "rccl_unique_id": (C.c_int, [p, C.c_size_t]),
            "rccl_init": (C.c_int, [p, p, C.c_size_t, C.c_int, C.c_int]),
            "rccl_init_all": (C.c_int, [C.POINTER(p), C.c_int]),
            "gather_tiles": (C.c_int, [p]),
            "gather_tiles_all": (C.c_int, [C.POINTER(p), C.c_int]),
            "rccl_info": (C.c_int, [p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "jit_prebuild": (C.c_int, [C.POINTER(SDFObject), C.c_int, C.c_int, C.POINTER(Config), C.POINTER(Camera), C.c_int, C.c_int, C.c_int,
                                       C.c_char_p, C.c_char_p, C.c_size_t]),
            "test_math": (C.c_int, [p, C.c_int, p, p, p, p, C.c_int]),
        }
        self.fn = {}
        for name, (res, args) in sig.items():
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in optional:
                    continue
                raise
            f.restype, f.argtypes = res, args
            self.fn[name] = f

    def has(self, name):
        return name in self.fn

    def backend(self):
        return self.fn["backend"]().decode()

    def call(self, name, *args):
        rc = self.fn[name](*args)
        if rc != 0:
            raise RtpbrError(rc, self.fn["last_error"]().decode(errors="replace"))
        return rc


_hip = None


def hip_api():
    """The product library. Raises if it is not built; never falls back to anything else."""
    global _hip
    if _hip is None:
        _hip = CApi(HIP_LIB_PATH, "rtpbr_")
        if not _hip.backend().startswith("hip"):
            raise RuntimeError(f"{HIP_LIB_PATH} reports backend {_hip.backend()!r}, expected hip-*")
    return _hip
