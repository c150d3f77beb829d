"""Build driver: compiles the HIP module for gfx950 in-tree (the .so travels to the GPU box).

  python -m raytracingpbr_amd.build            # build if sources are newer than the .so
  python -m raytracingpbr_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "librtpbr_hip.so")
SOURCES = ["rt_kernels.hip", "rt_capi.hip", "rt_rccl.hip", "rt_jit.hip"]
HEADERS = ["rt_math.hpp", "rt_types.hpp", "rt_device.hpp", "rt_trace.hpp", "rt_persistent.hpp", "rt_split.hpp", "rt_chain.hpp", "rt_ctx.hpp", "rt_jit_tu.hip", os.path.join("..", "..", "include", "rtpbr.h")]
# -ffp-contract=off: only the fmaf written in rt_math.hpp are fused (bit-reproducible results);
# no -ffast-math: f32 divide and sqrt stay correctly rounded.
# -fno-slp-vectorize: the SLP vectoriser pairs independent f32 ops of neighbouring boxes into
# v_pk_mul/v_pk_fma_f32; on gfx950 those issue no faster than two scalar ops and cost register-pair
# moves: measured +6 % on the headline kernel without them (122 -> 113 VGPRs).
# -amdgpu-use-amdgpu-trackers: the scheduler tracks register pressure with the GCN trackers: +1.7 %.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-fPIC", "-shared", "-Wno-unused-value", "-ldl"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra) + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
