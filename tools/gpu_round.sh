#!/bin/bash
# One GPU-box visit: parity tests, the bench line, and a rocprofv3 kernel-trace summary.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
echo "== nproc $(nproc)  $(grep -m1 'model name' /proc/cpuinfo)" | tee gpurun_out/host.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
find gpurun_out/prof_$TAG -name '*stats*' | head; 
for f in $(find gpurun_out/prof_$TAG -name '*kernel_stats.csv'); do head -12 $f; done
