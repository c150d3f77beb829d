cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fake or stand_in or schedule_independence or staging or full_size or rccl" > gpurun_out/pytest_sel.log 2>&1; tail -5 gpurun_out/pytest_sel.log | cut -c1-300
