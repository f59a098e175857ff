#!/bin/bash
# round 6, experiment 2: the drop-in path (honest clock) against the number of estimator groups and their worker pools, gate spin off
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp2; mkdir -p $O
for G in 1 2 4 8; do for T in 8 16 32; do
  GF_GROUP_THREADS=$T GF_GROUP_SPIN_US=0 GF_GROUP_TIMING=1 timeout 300 python bench.py --e2e-only --e2e-groups $G > $O/e2e_g${G}_t${T}.json 2> $O/e2e_g${G}_t${T}.err
done; done
GF_GROUP_THREADS=16 GF_GROUP_SPIN_US=50 timeout 300 python bench.py --e2e-only --e2e-groups 4 > $O/e2e_g4_t16_s50.json 2> $O/e2e_g4_t16_s50.err
GF_GROUP_THREADS=4 GF_GROUP_SPIN_US=0 timeout 300 python bench.py --e2e-only --e2e-groups 8 > $O/e2e_g8_t4.json 2> $O/e2e_g8_t4.err
for G in 2 4 8; do
  GF_GROUP_SPIN_US=0 timeout 300 python bench.py --e2e-only --e2e-groups $G --host-threads 8 > $O/e2e_h8_g${G}.json 2> $O/e2e_h8_g${G}.err
done
echo done
