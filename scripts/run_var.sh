cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $1 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step'])"; }
run "X=1"
run "GF_BA_PRIO=1"
run "GF_BA_PRIO=1 GF_TRK_PRIO=0"
run "GF_TRK_PRIO=1"
run "GF_BA_PRIO=0 GF_TRK_PRIO=1"
