#!/bin/bash
# round 6, experiment 4: chain form v2 (pipelined fronts): parity, cycle split (profiling build), kernel rate at 256 / 512 / 1024
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp4; mkdir -p $O
timeout 900 python -m pytest tests/test_backend_gpu.py -x -q -m gpu -k "chain_form" > $O/test_chain.txt 2>&1
tail -5 $O/test_chain.txt
GF_BA_CHAIN=1 timeout 1500 python -m pytest tests/test_backend_gpu.py -q -m gpu > $O/test_backend_under_chain.txt 2>&1
tail -4 $O/test_backend_under_chain.txt
python scripts/build_profile.py > $O/build_prof.txt 2>&1
GF_LIB_PATH=$PWD/ground-fusion_amd/lib/libgroundfusion_hip_prof.so GF_BA_CHAIN=1 timeout 300 python scripts/prof_chain.py 256 > $O/prof_chain_256.txt 2>&1
GF_LIB_PATH=$PWD/ground-fusion_amd/lib/libgroundfusion_hip_prof.so GF_BA_CHAIN=1 timeout 300 python scripts/prof_chain.py 512 > $O/prof_chain_512.txt 2>&1
GF_LIB_PATH=$PWD/ground-fusion_amd/lib/libgroundfusion_hip_prof.so timeout 300 python scripts/prof_step.py > $O/prof_dense_256.txt 2>&1
cat $O/prof_chain_256.txt | tail -6
K="--steps 60 --warmup 5 --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run --distinct 64"
for B in 256 512 1024; do
    GF_BA_CHAIN=1 timeout 300 python bench.py $K --batch $B > $O/kr_chain1_$B.json 2> $O/kr_chain1_$B.err
done
echo done
