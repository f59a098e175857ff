#!/bin/bash
# marg_finish: stage clocks of the pivot loop (profiling build) + parity of the marginalisation tests + determinism
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp12
timeout 900 python -m pytest tests/test_backend_gpu.py -m gpu -q -x -k "marg or gnss or prior" > gpurun_out/r06_exp12/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r06_exp12/pytest.log
python scripts/marg_determinism.py > gpurun_out/r06_exp12/determinism.txt 2>&1; tail -1 gpurun_out/r06_exp12/determinism.txt
python scripts/build_profile.py > gpurun_out/r06_exp12/build_profile.log 2>&1
GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so python scripts/prof_marg.py > gpurun_out/r06_exp12/prof_marg.txt 2>&1; tail -2 gpurun_out/r06_exp12/prof_marg.txt
bash scripts/r06_run.sh r06_exp12 backend | grep marg_finish | cut -c1-200
