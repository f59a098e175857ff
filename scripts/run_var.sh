cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/p0; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -- python scripts/prof_misc.py > /dev/null 2>&1
python scripts/timeline.py /tmp/p0 22 | head -12
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step'])"
