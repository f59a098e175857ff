#!/bin/bash
# marg_finish with one-wavefront pivot blocks and register-resident substitutions: parity tests, stage clocks, backend-alone kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp11
timeout 1200 python -m pytest tests/test_backend_gpu.py tests/test_estimator_gpu.py tests/test_replay_gpu.py -m gpu -q -x > gpurun_out/r06_exp11/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_exp11/pytest.log
python scripts/marg_determinism.py > gpurun_out/r06_exp11/determinism.txt 2>&1; tail -2 gpurun_out/r06_exp11/determinism.txt
python scripts/build_profile.py > gpurun_out/r06_exp11/build_profile.log 2>&1
GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so python scripts/prof_marg.py > gpurun_out/r06_exp11/prof_marg.txt 2>&1; tail -2 gpurun_out/r06_exp11/prof_marg.txt
bash scripts/r06_run.sh r06_exp11 backend
for i in 1 2; do python bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
