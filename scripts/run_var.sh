cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 250 python -m pytest tests/test_backend_gpu.py tests/test_golden.py tests/test_estimator_gpu.py -m gpu -x -q 2>&1 | tail -1
timeout 200 python bench.py --no-frontend --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step'])"
