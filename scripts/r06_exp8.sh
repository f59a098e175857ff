#!/bin/bash
# round 6, experiment 8: the full GPU suite on the tree with the visual sweep's concurrent phases and the cost-only last linearisation, then the bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp8; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt
K="--steps 60 --warmup 5 --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run"
timeout 300 python bench.py $K > $O/kr_new.json 2> $O/kr_new.err
GF_BA_COST_ONLY=0 timeout 300 python bench.py $K > $O/kr_full_last.json 2> $O/kr_full_last.err
for f in kr_new kr_full_last; do python -c "
import json; r=json.load(open('$O/$f.json')); i=r['gpu_ms_isolated']; print('$f', round(r['value']), round(r['ms_per_step'],3), 'jtj', round(i['jtj_ms'],4), 'step', round(i['step_ms'],4), 'solve', round(i['ba_solve_ms'],3), 'marg', round(i['ba_marginalize_ms'],3))"; done
