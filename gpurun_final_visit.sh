TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
bash tools/gpu_round6.sh r06final
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/pmc_sq_$TAG $OUT/pmc_fast_FETCH_SIZE_$TAG $OUT/pmc_fast_WRITE_SIZE_$TAG $OUT/prof_cfg_${TAG}c
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --keep-jit-cache"
$BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- $BENCH > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_bench_$TAG.json 2> $OUT/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_bench_$TAG.json 2> $OUT/pmc_write_$TAG.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $BENCH > $OUT/pmc_sq_bench_$TAG.json 2> $OUT/pmc_sq_$TAG.err
cd $R
bash tools/gpu_round3.sh $TAG > $OUT/round3_$TAG.log 2>&1; tail -5 $OUT/round3_$TAG.log
timeout 1200 python tests/soak/fuzz_soak.py 5000 5120 2>&1 | tail -3
ONLY_R6=1 ONLY_SRC=1 timeout 1200 python tests/soak/fuzz_soak.py 5000 5200 2>&1 | tail -3
