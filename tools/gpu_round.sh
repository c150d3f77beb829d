#!/bin/bash
# One GPU-box visit: parity tests, smoke, the bench line, a rocprofv3 kernel-trace summary and
# two PMC passes (FETCH_SIZE / WRITE_SIZE, separately, kernel-trace only — see
# MI355X_MICROARCH.md "HBM" and "rocprofv3 PMC slots").
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "nproc $(nproc); $(grep -m1 'model name' /proc/cpuinfo); cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee $OUT/host_$TAG.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
python bench.py --steps 3 --warmup 1 2>$OUT/bench_$TAG.err | tee $OUT/bench_$TAG.json
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --keep-jit-cache"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- $BENCH > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_bench_$TAG.json 2> $OUT/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_bench_$TAG.json 2> $OUT/pmc_write_$TAG.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $BENCH > $OUT/pmc_sq_bench_$TAG.json 2> $OUT/pmc_sq_$TAG.err
cd $R
python tools/prof_summary.py $TAG 2>&1 | tail -40
