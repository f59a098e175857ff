#!/bin/bash
# usage: scripts/r03_run.sh <tag> [tests] [bench1] [bench2] [bench4] [prof1] [prof2] [prof4]   (on the GPU box, from the repo root)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
for what in "$@"; do
  case $what in
    tests)  timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${tag}_pytest.log ;;
    c4test) timeout 900 python -m pytest tests/test_estimator_gpu.py -m gpu -x -q -s -k config4 > gpurun_out/${tag}_c4test.log 2>&1; echo "c4 rc $?"; tail -5 gpurun_out/${tag}_c4test.log ;;
    bench1) timeout 900 python bench.py > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err; echo "bench1 rc $?"; head -c 600 gpurun_out/${tag}_bench_c1.json; echo ;;
    bench2) timeout 900 python bench.py --config 2 > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err; echo "bench2 rc $?"; head -c 600 gpurun_out/${tag}_bench_c2.json; echo ;;
    bench4) timeout 1200 python bench.py --config 4 > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err; echo "bench4 rc $?"; head -c 600 gpurun_out/${tag}_bench_c4.json; echo; tail -3 gpurun_out/${tag}_bench_c4.err ;;
    prof1|prof2|prof4)
      c=${what#prof}
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_${tag}_c$c && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_c$c -- python $R/bench.py --config $c --no-cpu-baseline --no-e2e --no-pcie --steps 40 > $R/gpurun_out/${tag}_bench_c${c}_traced.json 2> $R/gpurun_out/${tag}_prof_c$c.err )
      f=$(find /tmp/prof_${tag}_c$c -name "*kernel_stats.csv" | head -1)
      python - "$f" > gpurun_out/${tag}_c${c}_kernel_stats.csv <<'P'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
print(",".join('"%s"' % c for c in rows[0]))
for r in rows[1:]:
    if r[0].startswith(("gf", "void gf")):
        print(",".join('"%s"' % c for c in r))
P
      echo "prof$c done"; head -8 gpurun_out/${tag}_c${c}_kernel_stats.csv | cut -c1-200 ;;
  esac
done
