#!/bin/bash
# (runs on commit 9c042d3, which has the two-solves-in-flight API and the --no-pipeline switch; reverted afterwards: profiles/r06_step_timeline_and_queued_solve.txt)
# two solves in flight: the new test, the back-end / shard suites, and the default-size loop with and without the queued next solve
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp19
timeout 1500 python -m pytest tests/test_shard_product_gpu.py tests/test_backend_gpu.py tests/test_estimator_gpu.py -m gpu -q -x > gpurun_out/r06_exp19/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp19/pytest.log
B="--no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run"
for i in 1 2 3; do
  for f in "" "--no-pipeline"; do python bench.py $B $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],4))"; done
done
