#!/bin/bash
# usage: scripts/r04_variants.sh <tag> "<ENV1=.. ENV2=..|bench args>" ...   each item: environment assignments, a '|', bench.py arguments; prints the isolated kernel times
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
i=0
for item in "$@"; do
  envs=${item%%|*}; args=${item#*|}
  out=gpurun_out/${tag}_v${i}.json
  env $envs timeout 600 python bench.py $args > $out 2> gpurun_out/${tag}_v${i}.err
  echo "== v$i [$envs] [$args] rc $?"
  python - $out <<'P'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("no json", e); sys.exit(0)
keys = ("value", "ms_per_step", "solves_per_s", "tracked_features_per_s")
print({k: round(r[k], 3) for k in keys if k in r})
for k in ("gpu_ms_isolated", "gpu_ms", "roofline", "roofline_step", "roofline_jtj", "roofline_jtj_split", "end_to_end", "pcie_inclusive"):
    if k in r: print(k, json.dumps(r[k])[:700])
P
  i=$((i+1))
done
