#!/bin/bash
# A/B differently built HIP libraries (RTPBR_HIP_LIB) on the headline config, interleaved twice.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for lib in "" $(ls $R/gpurun_libs_*.so 2>/dev/null); do
  echo -n "lib=${lib:-default} : "
  RTPBR_HIP_LIB=$lib VARIANTS='[{}]' REPS=2 timeout 120 python $R/tools/gpu_ab2.py 2>&1 | tail -1
done; done
