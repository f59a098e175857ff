cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_backend_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
rm -rf /tmp/p0; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -- python scripts/prof_misc.py > /dev/null 2>&1
python scripts/prof_misc.py --parse /tmp/p0 | head -8
python scripts/timeline.py /tmp/p0 30 | tail -22
