#!/bin/bash
# usage: scripts/r06_run.sh <tag> [tests] [bench1] [bench2] [bench4] [driver] [prof1] [backend] [tracker] [pmc] [adjud]   (pmc: with GF_BA_COST_ONLY=0, so that every dispatch of a sweep is a full linearisation)   (on the GPU box, from the repo root)
# Every artefact carries <tag>; the tag names the HEAD it was taken on (profiles/README.md).
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
filter() {  # rows of this repo's kernels out of a rocprofv3 kernel_stats.csv
python - "$1" <<'P'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
print(",".join('"%s"' % c for c in rows[0]))
for r in rows[1:]:
    if r[0].startswith(("gf", "void gf")):
        print(",".join('"%s"' % c for c in r))
P
}
trace() {  # name, bench args...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_${tag}_$name && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$name -- python $R/bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run --steps 40 "$@" > $R/gpurun_out/${tag}_${name}_traced.json 2> $R/gpurun_out/${tag}_${name}.err )
  f=$(find /tmp/prof_${tag}_$name -name "*kernel_stats.csv" | head -1)
  filter "$f" > gpurun_out/${tag}_${name}_kernel_stats.csv
  echo "$name traced"; head -7 gpurun_out/${tag}_${name}_kernel_stats.csv | cut -c1-160
}
for what in "$@"; do
  case $what in
    tests)   timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${tag}_pytest.log ;;
    bench1)  SECONDS=0; timeout 900 python bench.py > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err; echo "bench1 rc $? in $SECONDS s (the whole default run: timed loop, isolated passes, pcie / small-batch / end-to-end samples, cpu baseline)"; head -c 400 gpurun_out/${tag}_bench_c1.json; echo ;;
    bench2)  timeout 600 python bench.py --config 2 --no-e2e --no-cpu-baseline --no-small-batch --no-large-batch --no-other-configs > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err; echo "bench2 rc $?"; head -c 300 gpurun_out/${tag}_bench_c2.json; echo ;;
    bench4)  timeout 900 python bench.py --config 4 --no-cpu-baseline --no-small-batch --no-large-batch --no-other-configs > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err; echo "bench4 rc $?"; head -c 300 gpurun_out/${tag}_bench_c4.json; echo ;;
    prof1)   trace c1 ;;
    backend) trace backend_alone --no-frontend ;;
    tracker) trace tracker_alone --no-backend ;;
    pmc)     GF_BA_COST_ONLY=0 PMC_SKIP_SPLIT=1 bash scripts/pmc_collect.sh $tag; ls gpurun_out/${tag}_pmc_* | wc -l ;;
    driver)  SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_cmd.json 2> gpurun_out/${tag}_bench_driver_cmd_phases.txt; echo "driver command rc $? in $SECONDS s" ;;
    adjud)   python scripts/adjudicate_free_extrinsic_step.py --dump gpurun_out/${tag}_free_ex_step_hip.pkl | tail -1 ;;
  esac
done
