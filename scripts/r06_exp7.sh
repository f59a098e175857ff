#!/bin/bash
# round 6, experiment 7: LK with 2 / 4 points per wavefront: bit-exactness (the whole tracker suite under each setting), then the kernel's time
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp7; mkdir -p $O
for P in 2 4; do
  GF_LK_POINTS=$P timeout 1200 python -m pytest tests/test_tracker_gpu.py -x -q -m gpu > $O/test_tracker_p$P.txt 2>&1
  echo "P=$P"; tail -3 $O/test_tracker_p$P.txt
done
K="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run"
for P in 1 2 4; do
  GF_LK_POINTS=$P timeout 300 python bench.py $K > $O/kr_p$P.json 2> $O/kr_p$P.err
  python -c "
import json; r=json.load(open('$O/kr_p$P.json')); print('P=$P', round(r['value']), r['ms_per_step'], 'lk isolated ms', r['roofline']['launch_ms_isolated'], 'frac_isolated', r['roofline']['frac_isolated'], 'in step', r['roofline']['launch_ms'])"
done
