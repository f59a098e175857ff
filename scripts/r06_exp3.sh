#!/bin/bash
# round 6, experiment 3: the chain form of ba_step -- parity first, then the kernel-rate step at 256 / 512 / 1024 sequences per GPU against the dense form
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp3; mkdir -p $O
timeout 900 python -m pytest tests/test_backend_gpu.py -x -q -m gpu -k "chain_form" > $O/test_chain.txt 2>&1
tail -5 $O/test_chain.txt
GF_BA_CHAIN=1 timeout 1500 python -m pytest tests/test_backend_gpu.py -q -m gpu > $O/test_backend_under_chain.txt 2>&1
tail -8 $O/test_backend_under_chain.txt
K="--steps 60 --warmup 5 --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run --distinct 64"
for B in 256 512 1024; do
  for C in 0 1; do
    GF_BA_CHAIN=$C timeout 300 python bench.py $K --batch $B > $O/kr_chain${C}_$B.json 2> $O/kr_chain${C}_$B.err
    GF_BA_CHAIN=$C timeout 300 python bench.py $K --batch $B --no-frontend > $O/kr_be_chain${C}_$B.json 2> /dev/null
  done
done
echo done
