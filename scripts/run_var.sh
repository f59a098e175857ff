cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_backend_gpu.py -m gpu -x -q 2>&1 | tail -1
rm -rf /tmp/p0; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -- python scripts/prof_misc.py > /dev/null 2>&1
python scripts/timeline.py /tmp/p0 22 | head -6
