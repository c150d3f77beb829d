#!/bin/bash
# A/B of differently built libraries x option sets:  bash tools/gpu_lib_ab2.sh 'VARIANTS-json' lib1.so lib2.so ...
V=$1; shift
for lib in "$@"; do
  echo "== $lib"
  RTPBR_HIP_LIB=$PWD/raytracingpbr_amd/csrc/$lib VARIANTS="$V" timeout 300 python tools/gpu_ab2.py 2>&1 | tail -4
done
