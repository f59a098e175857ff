#!/bin/bash
# visual sweep: pair keys by readlane + operand prefetch in the MFMA loops -- stage clocks, back-end tests, back-end-alone trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp15
bash scripts/r06_exp14.sh | tail -5 | cut -c1-700
unset GF_LIB_PATH GF_BA_COST_ONLY
timeout 1200 python -m pytest tests/test_backend_gpu.py tests/test_estimator_gpu.py tests/test_stale_memory_gpu.py -m gpu -q -x > gpurun_out/r06_exp15/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp15/pytest.log
bash scripts/r06_run.sh r06_exp15 backend | cut -c1-200
