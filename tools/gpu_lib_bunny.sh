#!/bin/bash
# tools/gpu_bunny.py with several differently built libraries: bash tools/gpu_lib_bunny.sh <spp> lib1.so lib2.so ...
S=$1; shift
for lib in "$@"; do echo "== $lib"; RTPBR_HIP_LIB=$PWD/raytracingpbr_amd/csrc/$lib timeout 300 python tools/gpu_bunny.py $S 2>&1 | tail -2; done
