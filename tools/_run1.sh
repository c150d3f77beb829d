cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_r03a.err | tail -1 > gpurun_out/bench_r03a.json; tail -3 gpurun_out/bench_r03a.err; cut -c1-300 gpurun_out/bench_r03a.json
for w in c1 c3 c4 c5 src; do timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_r03a_$w.json; cut -c1-250 gpurun_out/bench_r03a_$w.json; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench or error_channel or schedule_independence" 2>&1 | tail -5
LIMIT=300 timeout 1000 python tools/gpu_rccl_init.py 2>&1 | tail -5
