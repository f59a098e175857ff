cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python -m pytest tests/test_backend_gpu.py tests/test_golden.py tests/test_estimator_gpu.py -m gpu -x -q 2>&1 | tail -2; done
rm -rf /tmp/p0; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -- python scripts/prof_misc.py > /dev/null 2>&1
python scripts/timeline.py /tmp/p0 12 | tail -10
