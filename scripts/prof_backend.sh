#!/bin/bash
# usage: scripts/prof_backend.sh <tag> [bench args]  -- rocprofv3 kernel trace of the back end alone (bench.py --no-frontend)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb_$tag -- python $R/bench.py --no-cpu-baseline --no-frontend "$@" > $R/gpurun_out/${tag}_backend_bench.json 2> $R/gpurun_out/${tag}_backend_prof.err
f=$(find /tmp/profb_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/${tag}_backend_kernel_stats.csv <<'P'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
print(",".join('"%s"' % c for c in rows[0]))
for r in rows[1:]:
    if r[0].startswith(("gf", "void gf")):
        print(",".join('"%s"' % c for c in r))
P
