"""profiling build of the library (-DGF_PROFILE_STEP: per-phase clock64 stamps inside ba_step)"""
import subprocess, os
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ground-fusion_amd')
srcs = [os.path.join(HERE, 'csrc', f) for f in sorted(os.listdir(os.path.join(HERE, 'csrc'))) if f.endswith('.hip')]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-DGF_PROFILE_STEP'] + os.environ.get('GF_EXTRA_FLAGS', '').split() + [
       '-Wno-unused-variable', '-o', HERE + '/lib/libgroundfusion_hip_prof.so'] + srcs
subprocess.check_call(cmd)
print('profiling build ok')
