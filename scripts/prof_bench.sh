#!/bin/bash
# usage: scripts/prof_bench.sh <tag>  -- untraced bench line + rocprofv3 kernel trace of the default bench (run on the GPU box from the repo root)
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_bench_traced.json 2> $R/gpurun_out/${tag}_prof.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/${tag}_kernel_stats.csv <<'P'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
print(",".join('"%s"' % c for c in rows[0]))
for r in rows[1:]:
    if r[0].startswith(("gf", "void gf")):
        print(",".join('"%s"' % c for c in r))
P
