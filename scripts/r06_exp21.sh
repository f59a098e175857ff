#!/bin/bash
# prior / IMU / wheel sweep with the IMU factors shared out over seven wavefronts: stage clocks, back-end tests, back-end-alone trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp21
bash scripts/r06_exp20.sh | tail -9
timeout 1500 python -m pytest tests/test_backend_gpu.py tests/test_estimator_gpu.py tests/test_stale_memory_gpu.py tests/test_replay_gpu.py -m gpu -q -x > gpurun_out/r06_exp21/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp21/pytest.log
bash scripts/r06_run.sh r06_exp21 backend | cut -c1-200
