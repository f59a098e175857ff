#!/bin/bash
# stage clocks of a full prior / IMU / wheel sweep (profiling build; GF_BA_COST_ONLY=0 so that the last launch of the solve is a full one)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp20
python scripts/build_profile.py > gpurun_out/r06_exp20/build_profile.log 2>&1
GF_BA_COST_ONLY=0 GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so python scripts/prof_miscwin.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_exp20/miscwin.txt
