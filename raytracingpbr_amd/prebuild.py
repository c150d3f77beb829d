"""Ahead-of-time catalog of scene-specialised code objects (no GPU needed).

The reference gets its specialised kernels from Taichi's JIT at first call (`ti.init`, src/config.py:5; `ti.static` unrolling of
the object loop, src/scene.py:44-56); this library compiles them with `hipcc --genco` at run time (csrc/rt_jit.hip) — which needs
a compiler and the kernel sources on the target.  The scenes of BASELINE.json are known at build time, so their instances are
compiled HERE, into ``raytracingpbr_amd/data/jit/`` (a build artefact like the library itself: git-ignored, shipped with the
package), and ``rt_jit.hip`` looks there before it forks a compiler: a target without hipcc still runs the baked kernels.

    python -m raytracingpbr_amd.prebuild [--all] [--list] [names ...]

``__graft_entry__.build()`` calls :func:`prebuild` for the default set (exact kernels of every BASELINE config as bench.py
sets them up, the src/ pipeline at its three sizes, the headline's tile partitions for 2 / 4 / 8 ranks); ``--all`` adds the
tolerance flavour's instances.  A code object's name carries a hash of the kernel sources: after a source change the stale
files are simply never found (and this script replaces them).
"""
import ctypes as C
import os
import sys
import time

from . import workloads
from ._capi import HIP_LIB_PATH
from .config import Config
from .dataclass import Camera, SDFObject
from .tiles import default_tile

CATALOG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "jit")

# (name, workload, (W, H), world, options) — options exactly as bench.py's make_renderer / side_config set them
DEFAULT = [
    ("c2", "c2", (0, 0), 1, "jit=1 jit_bake=2"),                 # the headline: scene, configuration and camera baked
    ("c2_camera_free", "c2", (0, 0), 1, "jit=1 jit_bake=1"),     # ... the camera a launch argument (an interactive host)
    ("c2_unbaked", "c2", (0, 0), 1, "jit=1 jit_bake=0"),         # ... shape types and rotation classes only
    ("c2_tiles2", "c2", (0, 0), 2, "jit=1 jit_bake=2"),          # bench.py --gpus N: the frame tile-partitioned
    ("c2_tiles4", "c2", (0, 0), 4, "jit=1 jit_bake=2"),
    ("c2_tiles8", "c2", (0, 0), 8, "jit=1 jit_bake=2"),
    ("c1", "c1", (0, 0), 1, "jit=1 jit_bake=2"),
    ("c3", "c3", (0, 0), 1, "jit=1 jit_bake=2"),
    ("c3_valu", "c3", (0, 0), 1, "jit=1 jit_bake=2 mlp_mfma=0"),
    ("c4", "c4", (0, 0), 4, "jit=1 jit_bake=2"),
    ("c5", "c5", (0, 0), 8, "jit=1 jit_bake=2"),
    ("src", "src", (0, 0), 1, "jit=1 jit_bake=2"),
    ("src_768", "src", (768, 432), 1, "jit=1 jit_bake=2"),
    ("src_4k", "src", (3840, 2160), 1, "jit=1 jit_bake=2"),
    ("src_768_camera_free", "src", (768, 432), 1, "jit=1 jit_bake=1"),      # the interactive viewer (examples/src_viewer.py)
]
FAST = [
    ("c2_fast", "c2", (0, 0), 1, "jit=1 jit_bake=2 precision=1"),
    ("c3_fast", "c3", (0, 0), 1, "jit=1 jit_bake=2 precision=1"),
    ("c4_fast", "c4", (0, 0), 4, "jit=1 jit_bake=2 precision=1"),
    ("src_fast", "src", (0, 0), 1, "jit=1 jit_bake=2 precision=1"),
]


def _lib():
    lib = C.CDLL(HIP_LIB_PATH)
    lib.rtpbr_last_error.restype = C.c_char_p
    lib.rtpbr_jit_prebuild.argtypes = [C.POINTER(SDFObject), C.c_int, C.c_int, C.POINTER(Config), C.POINTER(Camera), C.c_int, C.c_int, C.c_int,
                                       C.c_char_p, C.c_char_p, C.c_size_t]
    return lib


def prebuild_one(lib, wl_name, dims, world, options):
    """compile (or find) the instance rtpbr_sample() would ask for; returns the code object's path"""
    wl = workloads.get(wl_name, dims[0], dims[1])
    n = len(wl.scene.objects)
    arr = (SDFObject * n)(*wl.scene.objects)
    W, H = wl.cfg.width, wl.cfg.height
    tw, th = default_tile(W, H, world) if world > 1 else (0, 0)
    buf = C.create_string_buffer(1024)
    rc = lib.rtpbr_jit_prebuild(arr, n, 1 if wl.scene.scale10 else 0, C.byref(wl.cfg), C.byref(wl.scene.camera), tw, th, world,
                                options.encode(), buf, 1024)
    if rc:
        raise RuntimeError(f"rtpbr_jit_prebuild({wl_name}, {options}): {lib.rtpbr_last_error().decode(errors='replace')}")
    return buf.value.decode()


def prebuild(entries=None, dest=CATALOG_DIR, verbose=False):
    """Fill the catalog; returns {name: path}.  Cheap when everything is there already (a lookup per entry)."""
    entries = DEFAULT if entries is None else entries
    os.makedirs(dest, mode=0o700, exist_ok=True)
    old = os.environ.get("RTPBR_JIT_CACHE")
    os.environ["RTPBR_JIT_CACHE"] = dest          # the compiler's output directory IS the catalog
    os.environ.setdefault("RTPBR_JIT_CACHE_MAX", "4096")
    out = {}
    try:
        lib = _lib()
        for name, wl_name, dims, world, options in entries:
            t0 = time.perf_counter()
            out[name] = prebuild_one(lib, wl_name, dims, world, options)
            if verbose:
                print(f"  {name:22s} {time.perf_counter() - t0:5.1f} s  {os.path.basename(out[name])}")
    finally:
        if old is None:
            del os.environ["RTPBR_JIT_CACHE"]
        else:
            os.environ["RTPBR_JIT_CACHE"] = old
    # code objects of OTHER source versions are dead weight (their names carry the old source hash): drop them
    keep = {os.path.basename(p) for p in out.values()}
    if entries is DEFAULT or entries == DEFAULT + FAST:
        suffix = {b.rsplit("_", 1)[-1] for b in keep}
        for f in os.listdir(dest):
            if f.endswith(".hsaco") and f.rsplit("_", 1)[-1] not in suffix:
                os.unlink(os.path.join(dest, f))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    table = DEFAULT + (FAST if "--all" in sys.argv else [])
    if "--list" in sys.argv:
        for e in DEFAULT + FAST:
            print(e)
        raise SystemExit(0)
    if args:
        table = [e for e in DEFAULT + FAST if e[0] in args]
    t0 = time.perf_counter()
    res = prebuild(table, verbose=True)
    print(f"{len(res)} code objects in {CATALOG_DIR} ({time.perf_counter() - t0:.1f} s)")
