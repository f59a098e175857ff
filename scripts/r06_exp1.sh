#!/bin/bash
# round 6, experiment 1: (a) the kernel-rate step at 256 / 512 / 1024 sequences per GPU for the shipped back end and the small-footprint variants that exist as switches;
# (b) the drop-in path (honest clock) against worker-pool sizes and the gate spin.  Output: gpurun_out/r06_exp1/*.json
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp1; mkdir -p $O
K="--steps 60 --no-large-batch --no-other-configs --no-long-run --warmup 5 --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --distinct 64"
for B in 256 512 1024; do
  timeout 300 python bench.py $K --batch $B > $O/kr_default_$B.json 2> $O/kr_default_$B.err
  GF_BA_STEP_WAVES=4 timeout 300 python bench.py $K --batch $B > $O/kr_w4_$B.json 2> $O/kr_w4_$B.err
  GF_BA_FORCE_GLOBAL=1 timeout 300 python bench.py $K --batch $B > $O/kr_glob_$B.json 2> $O/kr_glob_$B.err
  GF_BA_FORCE_GLOBAL=1 GF_BA_STEP_WAVES=4 timeout 300 python bench.py $K --batch $B > $O/kr_glob_w4_$B.json 2> $O/kr_glob_w4_$B.err
  timeout 300 python bench.py $K --batch $B --no-frontend > $O/kr_be_default_$B.json 2> /dev/null
  GF_BA_FORCE_GLOBAL=1 GF_BA_STEP_WAVES=4 timeout 300 python bench.py $K --batch $B --no-frontend > $O/kr_be_glob_w4_$B.json 2> /dev/null
done
for T in 8 16 32 64; do for S in 0 300; do
  GF_GROUP_THREADS=$T GF_GROUP_SPIN_US=$S GF_GROUP_TIMING=1 timeout 300 python bench.py --e2e-only > $O/e2e_t${T}_s${S}.json 2> $O/e2e_t${T}_s${S}.err
done; done
GF_GROUP_THREADS=32 GF_HOST_THREADS=32 timeout 300 python bench.py --e2e-only > $O/e2e_t32_h32.json 2> $O/e2e_t32_h32.err
GF_GROUP_THREADS=32 GF_HOST_THREADS=4 timeout 300 python bench.py --e2e-only > $O/e2e_t32_h4.json 2> $O/e2e_t32_h4.err
timeout 300 python bench.py --e2e-only --e2e-groups 1 > $O/e2e_one_group.json 2> $O/e2e_one_group.err
echo done
