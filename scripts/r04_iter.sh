#!/bin/bash
# usage: scripts/r03_iter.sh <tag> [what...]   quick iteration loop on the GPU box: backend parity tests, ba_step stage clocks, short bench
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for what in "$@"; do
  case $what in
    chol)    ./bin/bench_chol16 ;;
    btest)   timeout 900 python -m pytest tests/test_backend_gpu.py -m gpu -x -q 2>&1 | tail -5 ;;
    ttest)   timeout 900 python -m pytest tests/test_tracker_gpu.py -m gpu -x -q 2>&1 | tail -5 ;;
    etest)   timeout 900 python -m pytest tests/test_estimator_gpu.py -m gpu -x -q 2>&1 | tail -5 ;;
    stamps)  GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so timeout 600 python scripts/prof_step.py 2>&1 | tail -4 ;;
    vstamps) GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so timeout 600 python scripts/prof_viswin.py 2>&1 | tail -6 ;;
    mstamps) GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so timeout 600 python scripts/prof_miscwin.py 2>&1 | tail -6 ;;
    gstamps) GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so timeout 600 python scripts/prof_marg.py 2>&1 | tail -6 ;;
    bench)   timeout 900 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python - gpurun_out/${tag}_bench.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.0f  ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["gpu_ms_isolated"].items()})
P
      ;;
    backend) ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb_$tag && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -- python $R/bench.py --no-cpu-baseline --no-e2e --no-pcie --steps 40 --no-frontend > $R/gpurun_out/${tag}_backend_bench.json 2> $R/gpurun_out/${tag}_backend.err )
      f=$(find /tmp/pb_$tag -name "*kernel_stats.csv" | head -1)
      python - "$f" <<'P' | tee gpurun_out/${tag}_backend_kernel_stats.csv
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
print(",".join(rows[0][:4] + rows[0][5:7]))
for r in rows[1:]:
    if r[0].startswith(("gf", "void gf")):
        print(",".join([r[0][:70]] + r[1:4] + r[5:7]))
P
      ;;
  esac
done
