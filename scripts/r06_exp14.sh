#!/bin/bash
# stage clocks of the sweeps and the step on the final sources (profiling build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp14
python scripts/build_profile.py > gpurun_out/r06_exp14/build_profile.log 2>&1
export GF_LIB_PATH=$R/ground-fusion_amd/lib/libgroundfusion_hip_prof.so
export GF_BA_COST_ONLY=0
for s in "prof_viswin.py" "prof_viswin.py marg"; do echo "== $s"; python scripts/$s 2>&1 | grep -v "amdgpu.ids" | tail -3; done > gpurun_out/r06_exp14/stages.txt
cat gpurun_out/r06_exp14/stages.txt
