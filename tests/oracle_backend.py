"""Test helper: drive the CPU oracle (oracle/librt_oracle.so) through the same Python
Renderer class that drives the HIP library.  Only tests/, smoke() and bench.py's
cpu_baseline leg may use this."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracingpbr_amd._capi import CApi
from raytracingpbr_amd.renderer import Renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "librt_oracle.so")
_OPTIONAL = ("host_alloc", "host_free", "buffer_device_ptr", "read_buffer_async", "read_wait", "jit_prebuild", "packed_bytes", "pack_tiles", "unpack_tiles", "last_sample_ms", "last_primary_ms", "get_stream", "set_option",
             "test_math", "rccl_unique_id", "rccl_init", "rccl_init_all", "gather_tiles", "gather_tiles_all", "rccl_info")

_api = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def use_library(path):
    """bench.py's cpu_baseline leg times the -O3 -march=native build of the same source"""
    global _api, ORACLE_LIB
    ORACLE_LIB, _api = path, None


def oracle_api():
    global _api
    if _api is None:
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        _api = CApi(ORACLE_LIB, "rto_", optional=_OPTIONAL)
        assert _api.backend() == "cpu-oracle"
        lib = _api.lib
        lib.rto_set_threads.argtypes = [C.c_void_p, C.c_int]
        lib.rto_set_sample_base.argtypes = [C.c_void_p, C.c_uint32]
    return _api


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU
    box shows 256 logical CPUs to a container limited to 16; 256 OpenMP threads on 16 cores crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


class OracleRenderer(Renderer):
    def __init__(self, scene, config, camera=None, threads=0):
        super().__init__(scene, config, camera, device=0, api=oracle_api())
        self.api.lib.rto_set_threads(self._ctx, threads or min(usable_cores(), 32))

    def set_sample_base(self, base):
        self.api.lib.rto_set_sample_base(self._ctx, base)
