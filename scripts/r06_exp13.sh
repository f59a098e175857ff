#!/bin/bash
# detector strips with row-level skipping: tracker parity tests + tracker-alone kernel stats + two default-size bench values
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp13
timeout 1200 python -m pytest tests/test_tracker_gpu.py tests/test_featsweep_gpu.py tests/test_replay_gpu.py tests/test_estimator_gpu.py -m gpu -q -x > gpurun_out/r06_exp13/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp13/pytest.log
bash scripts/r06_run.sh r06_exp13 tracker | cut -c1-200
for i in 1 2; do python bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
