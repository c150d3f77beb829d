#!/bin/bash
# run tools/gpu_configs.py <configs...> with several libraries: bash tools/gpu_lib_cfg.sh "C4_..." lib1.so lib2.so
C=$1; shift
for lib in "$@"; do echo "== $lib"; RTPBR_HIP_LIB=$PWD/raytracingpbr_amd/csrc/$lib timeout 300 python tools/gpu_configs.py $C 2>&1 | tail -3; done
