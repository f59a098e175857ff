#!/bin/bash
# corner selection by repeated maximum for the frames that want a handful of corners: tracker parity tests (both forms), tracker-alone trace, bench values
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp22
timeout 1500 python -m pytest tests/test_tracker_gpu.py tests/test_featsweep_gpu.py tests/test_replay_gpu.py tests/test_estimator_gpu.py -m gpu -q -x > gpurun_out/r06_exp22/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp22/pytest.log
GF_SELECT_TOPK=0 timeout 900 python -m pytest tests/test_tracker_gpu.py -m gpu -q -x > gpurun_out/r06_exp22/pytest_sort_only.log 2>&1; echo "pytest (GF_SELECT_TOPK=0) rc $?"; tail -2 gpurun_out/r06_exp22/pytest_sort_only.log
bash scripts/r06_run.sh r06_exp22 tracker | cut -c1-200
for i in 1 2; do python bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
