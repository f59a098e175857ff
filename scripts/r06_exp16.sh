#!/bin/bash
# visual sweep without the head zeroing of the E^T F rows: back-end tests, back-end-alone trace, write bytes by the counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp16
timeout 1500 python -m pytest tests/test_backend_gpu.py tests/test_estimator_gpu.py tests/test_stale_memory_gpu.py tests/test_replay_gpu.py -m gpu -q -x > gpurun_out/r06_exp16/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp16/pytest.log
bash scripts/r06_run.sh r06_exp16 backend | cut -c1-200
for i in 1 2; do python bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
