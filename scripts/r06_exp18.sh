#!/bin/bash
# the GPU timeline of a timed step on the final sources (kernel trace -> scripts/step_timeline.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp18
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 40 --warmup 5 --no-e2e --no-small-batch --no-pcie --no-cpu-baseline --no-large-batch --no-other-configs --no-long-run > $R/gpurun_out/r06_exp18/bench_traced.json 2> $R/gpurun_out/r06_exp18/bench_traced.err )
python scripts/step_timeline.py /tmp/tl 40 > gpurun_out/r06_exp18/step_timeline.txt 2>&1
cat gpurun_out/r06_exp18/step_timeline.txt | head -60
